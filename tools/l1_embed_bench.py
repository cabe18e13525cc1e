"""Layer 1 of the heuristic network, two ways, per environment geometry at the network's real width (5120 units): the one-hot MFMA
kernel (dca_l1_onehot_gemm: fp32 = three bf16 weight planes, bf16 = one) against the embedding sum on the vector pipes
(dca_l1_embed).  Prints ms per launch (median of 7 after 2 warm-ups) for the two output forms the network uses."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcubea_amd import _lib
from deepcubea_amd.utils.pytorch_models import l1_weight_tiles

M = int(sys.argv[1]) if len(sys.argv) > 1 else 409600
N = 5120


def timed(fn, reps=7):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


for name, D, depth in (("cube3", 54, 6), ("puzzle15", 16, 16), ("puzzle24", 25, 25), ("puzzle35", 36, 36), ("puzzle48", 49, 49), ("lightsout7", 49, 6)):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(N, D * depth, generator=g) * 0.1
    b = torch.randn(N, generator=g).cuda()
    if D == depth:
        x = torch.stack([torch.randperm(D, generator=g) for _ in range(4096)]).to(torch.uint8).repeat(M // 4096, 1).cuda()
    else:
        x = torch.randint(0, depth, (M, D), generator=g).to(torch.uint8).cuda()
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    wt = w.t().contiguous().cuda()
    row = [name]
    if _lib.l1_supported(D, depth):
        t3 = l1_weight_tiles(w, 3, _lib.l1_kpad(D, depth)).cuda()
        t1 = l1_weight_tiles(w.bfloat16().float(), 1, _lib.l1_kpad(D, depth)).cuda()
        row.append("mfma planes %.3f" % timed(lambda: _lib.l1_onehot_gemm(x, depth, t3, 3, b, True, torch.float32, split="planes", overflow=ovf)))
        row.append("mfma bf16 %.3f" % timed(lambda: _lib.l1_onehot_gemm(x, depth, t1, 1, b, True, torch.bfloat16)))
    row.append("embed planes %.3f" % timed(lambda: _lib.l1_embed(x, depth, wt, b, True, split="planes", overflow=ovf)))
    row.append("embed bf16 %.3f" % timed(lambda: _lib.l1_embed(x, depth, wt, b, True, torch.bfloat16)))
    row.append("embed f32 %.3f" % timed(lambda: _lib.l1_embed(x, depth, wt, b, True)))
    lds = M * N * D * 4 / 1e9
    print("  ".join(row), " | ms per %d x %d; LDS gather bytes %.1f GB" % (M, N, lds), flush=True)
