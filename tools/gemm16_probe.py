#!/usr/bin/env python3
"""What bounds the bf16 dense layer?  Times dca_gemm16 variants 2 (general tail) and 3 (lean tail, the default) and the
library GEMM on (a) random operands, (b) ALL-ZERO operands (same instructions, same bytes, no
switching activity in the matrix pipe: if the chip is power-limited the zero run clocks higher) and (c) a sweep over K
(tail-dominated ... K-loop-dominated).   python tools/gemm16_probe.py [rows]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["2", "3"])]
dt = torch.bfloat16


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def interleaved(cands, rounds=4):
    ms = {name: [] for name, _ in cands}
    for r in range(rounds):
        for name, fn in cands:
            t = timed(fn)
            if r > 0:
                ms[name].append(t)
    return {name: round(sorted(v)[len(v) // 2], 4) for name, v in ms.items()}


def hip(v, x, w, b):
    def run():
        _lib.gemm16_variant(v)
        return _lib.gemm16(x, w, b, None, True)
    return run


n = 1024
for k in [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1024", "5120"])]:
    for fill in ("random", "zeros"):
        g = torch.Generator().manual_seed(k)
        if fill == "random":
            x = (torch.randn(m, k, generator=g) * 0.5).to(dt).cuda()
            w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dt).cuda()
        else:
            x = torch.zeros(m, k, dtype=dt, device="cuda")
            w = torch.zeros(n, k, dtype=dt, device="cuda")
        b32 = torch.randn(n, generator=g).cuda()
        bdt = b32.to(dt)
        cands = [("hip_v%d" % v, hip(v, x, w, b32)) for v in variants]
        cands.append(("library", lambda: torch._addmm_activation(bdt, x, w.t())))
        res = interleaved(cands)
        flops = 2.0 * m * n * k
        print(json.dumps({"k": k, "operands": fill, "ms": res, "tflops": {a: round(flops / t / 1e9, 1) for a, t in res.items()}}))
        del x, w
        torch.cuda.empty_cache()
_lib.gemm16_variant(3)
