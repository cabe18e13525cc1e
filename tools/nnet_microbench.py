"""Where does the heuristic forward's time go?  Times the cube3 ResNet (BN folded) on [M,324] one-hot rows:
  eager      Linear / relu / add as separate kernels (what ResnetModel.trunk does under autocast)
  fused      torch._addmm_activation (bias+ReLU epilogue) + addmm with the skip as C + one in-place bias/ReLU pass
  gemm_only  the 19 matmuls alone (floor for any epilogue fusion)
Run on the GPU box:  python tools/nnet_microbench.py [M]"""
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
dev = "cuda"
torch.manual_seed(0)
dims = [(324, 5000), (5000, 1000)] + [(1000, 1000)] * 8 + [(1000, 1)]
FLOP = 2 * M * sum(a * b for a, b in dims)


def mk(dt):
    Ws = [(torch.randn(a, b, device=dev) * (1.0 / a ** 0.5)).to(dt) for a, b in dims]  # [in,out]
    bs = [(torch.randn(b, device=dev) * 0.1).to(dt) for a, b in dims]
    return Ws, bs


def eager(x, Ws, bs):
    x = torch.relu(torch.addmm(bs[0], x, Ws[0]))
    x = torch.relu(torch.addmm(bs[1], x, Ws[1]))
    for b in range(4):
        skip = x
        x = torch.relu(torch.addmm(bs[2 + 2 * b], x, Ws[2 + 2 * b]))
        x = torch.addmm(bs[3 + 2 * b], x, Ws[3 + 2 * b])
        x = torch.relu(x + skip)
    return torch.addmm(bs[10], x, Ws[10])


def fused(x, Ws, bs):
    x = torch._addmm_activation(bs[0], x, Ws[0])
    x = torch._addmm_activation(bs[1], x, Ws[1])
    for b in range(4):
        h = torch._addmm_activation(bs[2 + 2 * b], x, Ws[2 + 2 * b])
        x = torch.addmm(x, h, Ws[3 + 2 * b])  # skip rides in as C
        x.add_(bs[3 + 2 * b]).relu_()
    return torch.addmm(bs[10], x, Ws[10])


def gemm_only(x, Ws, bs):
    x = x @ Ws[0]
    x = x @ Ws[1]
    for b in range(4):
        x = x @ Ws[2 + 2 * b]
        x = x @ Ws[3 + 2 * b]
    return x @ Ws[10]


def t(fn, *a, n=5):
    for _ in range(2):
        y = fn(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = fn(*a)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y


for dt in (torch.bfloat16, torch.float16, torch.float32):
    Ws, bs = mk(dt)
    idx = torch.randint(0, 6, (M, 54), device=dev)
    x = torch.nn.functional.one_hot(idx, 6).reshape(M, 324).to(dt)
    ref = None
    for name, fn in (("eager", eager), ("fused", fused), ("gemm_only", gemm_only)):
        ms, y = t(fn, x, Ws, bs)
        d = ""
        if name == "eager":
            ref = y.float()
        elif name == "fused":
            d = " maxdiff_vs_eager=%.3g" % (y.float() - ref).abs().max().item()
        print("%-8s %-10s M=%d %.3f ms  %.1f TFLOP/s  %.3g states/s%s" % (str(dt).split(".")[1], name, M, ms, FLOP / ms / 1e9,
                                                                        M / ms * 1e3, d), flush=True)
    # per-layer GEMM rates
    for (a, b) in ((324, 5000), (5000, 1000), (1000, 1000)):
        A = torch.randn(M, a, device=dev).to(dt)
        W = torch.randn(a, b, device=dev).to(dt)
        ms, _ = t(torch.matmul, A, W, n=10)
        print("   gemm %dx%dx%d %s: %.3f ms %.1f TFLOP/s" % (M, b, a, str(dt).split(".")[1], ms, 2 * M * a * b / ms / 1e9), flush=True)
