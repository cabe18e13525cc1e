"""Probe: fp32-accurate GEMM from three fp16 products (xh*wh + xl*wh + xh*wl, weights pre-scaled by a power of two) on the
library's f16 GEMM with fp32 output, for the network's layer shapes."""
import torch
torch.manual_seed(0)
M = 204800
def t(fn, n=5):
    for _ in range(2): y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y
def split2(x):
    h = x.to(torch.float16); l = (x - h.float()).to(torch.float16)
    return h, l
for K, N in ((1024, 1024), (5120, 1024)):
    x = torch.relu(torch.randn(M, K, device="cuda")) * 1.3
    w = torch.randn(N, K, device="cuda") * (0.03 if K == 1024 else 0.014)
    ms32, y32 = t(lambda: x @ w.t())
    s = 2.0 ** torch.floor(torch.log2(1024.0 / w.abs().max())).item()
    xh, xl = split2(x); wh, wl = split2(w * s)
    A3 = torch.cat([xh, xl, xh], dim=1).contiguous(); W3 = torch.cat([wh, wh, wl], dim=1).contiguous()
    ms3, y3 = t(lambda: torch.mm(A3, W3.t(), out_dtype=torch.float32))
    y3 = y3 / s
    idx = torch.randint(0, M, (2048,), device="cuda")
    ref = x[idx].double() @ w.double().t()
    e32 = ((y32[idx].double() - ref).abs().max() / ref.abs().max()).item()
    e3 = ((y3[idx].double() - ref).abs().max() / ref.abs().max()).item()
    print("K=%d N=%d: fp32 GEMM %.3f ms (%.0f TF, err %.2e) | f16x3 %.3f ms (%.0f TF on pipe, %.0f TF fp32-equivalent, err %.2e, scale 2^%d)"
          % (K, N, ms32, 2e-9*M*K*N/ms32, e32, ms3, 2e-9*M*3*K*N/ms3, 2e-9*M*K*N/ms3, e3, int(torch.log2(torch.tensor(s)).item())))
