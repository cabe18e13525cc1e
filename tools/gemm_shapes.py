"""Library GEMM rate vs padding of the ResNet's dims (hidden 1000 -> 1024, 5000 -> 5120, one-hot 324 -> 384)."""
import sys, torch
dev = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for dt in (torch.bfloat16, torch.float32):
    for M in (60000, 61440, 204800):
        for (k, n) in ((324, 5000), (384, 5120), (320, 5120), (5000, 1000), (5120, 1024), (1000, 1000), (1024, 1024)):
            A = torch.randn(M, k, device=dev).to(dt); W = torch.randn(k, n, device=dev).to(dt); Wt = W.t().contiguous()
            bias = torch.randn(n, device=dev).to(dt)
            ms = t(lambda: A @ W)
            ms2 = t(lambda: torch.nn.functional.linear(A, Wt))
            ms3 = t(lambda: torch._addmm_activation(bias, A, W))
            print("%s M=%d K=%d N=%d: A@W %.3f ms %.0f TF | linear(W^T) %.3f ms %.0f TF | addmm_relu %.3f ms %.0f TF" % (
                str(dt)[6:], M, k, n, ms, 2e-9*M*k*n/ms, ms2, 2e-9*M*k*n/ms2, ms3, 2e-9*M*k*n/ms3), flush=True)
