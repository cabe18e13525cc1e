"""Probe: fp32-accurate GEMM from six bf16 products on the library's bf16 GEMM (K-concatenated planes, fp32 output)."""
import torch
M, K, N = 204800, 1024, 1024
torch.manual_seed(0)
def t(fn, n=5):
    for _ in range(2): y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y
def split3(x):
    h = x.to(torch.bfloat16); r = x - h.float(); m = r.to(torch.bfloat16); l = (r - m.float()).to(torch.bfloat16)
    return h, m, l
x = torch.relu(torch.randn(M, K, device="cuda")) * 1.3
w = torch.randn(N, K, device="cuda") * 0.03
ms32, y32 = t(lambda: x @ w.t())
print("fp32 GEMM %.3f ms (%.0f TF)" % (ms32, 2e-9 * M * K * N / ms32))
xh, xm, xl = split3(x); wh, wm, wl = split3(w)
A6 = torch.cat([xh, xh, xm, xh, xm, xl], dim=1).contiguous()      # [M, 6K]
W6 = torch.cat([wh, wm, wh, wl, wm, wh], dim=1).contiguous()      # [N, 6K]
try:
    ms6, y6 = t(lambda: torch.mm(A6, W6.t(), out_dtype=torch.float32))
    print("bf16x6 K-concat, fp32 out: %.3f ms (%.0f TF on the pipe, %.0f TF fp32-equivalent)" % (ms6, 2e-9 * M * 6 * K * N / ms6, 2e-9 * M * K * N / ms6))
except Exception as e:
    print("out_dtype path failed:", repr(e)[:200]); y6 = None
ms6b, _ = t(lambda: A6 @ W6.t())
print("same GEMM, bf16 out: %.3f ms (%.0f TF)" % (ms6b, 2e-9 * M * 6 * K * N / ms6b))
idx = torch.randint(0, M, (4096,), device="cuda")
ref = x[idx].double() @ w.double().t()
print("max rel err vs fp64: fp32 GEMM %.3g" % ((y32[idx].double() - ref).abs().max() / ref.abs().max()).item())
if y6 is not None:
    print("max rel err vs fp64: bf16x6     %.3g" % ((y6[idx].double() - ref).abs().max() / ref.abs().max()).item())
ms_split, _ = t(lambda: torch.cat(list(split3(x)), dim=1))
print("split (torch, unfused) %.3f ms" % ms_split)
