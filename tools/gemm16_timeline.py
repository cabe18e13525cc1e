#!/usr/bin/env python3
"""Per-tile timeline of the persistent bf16 layer kernel (dca_gemm16 variant 5): wave 0 of every workgroup stamps the 100 MHz
wall clock at tile start, after the K loop, after the tail's last store is issued and after it is acknowledged
(dca_gemm16_debug knob 1).   python tools/gemm16_timeline.py [rows] [skew_us,...]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
skews = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]
dt = torch.bfloat16
L = _lib.lib()
L.dca_gemm16_debug.argtypes = [C.c_int, C.c_longlong]
n = 1024


def pct(a):
    return [round(float(np.percentile(a, q)), 2) for q in (10, 50, 90)]


for k in (1024, 5120):
    g = torch.Generator().manual_seed(k)
    x = (torch.randn(m, k, generator=g) * 0.5).to(dt).cuda()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dt).cuda()
    b32 = torch.randn(n, generator=g).cuda()
    skip = torch.randn(m, n, generator=g).to(dt).cuda()
    for with_skip in (False, True):
        for skew in skews:
            grid = 256
            per = ((m + 255) // 256 * 4 + grid - 1) // grid + 1
            prof = torch.zeros(per * grid * 4, dtype=torch.int64, device="cuda")
            _lib.gemm16_variant(5)
            L.dca_gemm16_debug(2, skew)
            run = (lambda: _lib.gemm16(x, w, None, skip, True)) if with_skip else (lambda: _lib.gemm16(x, w, b32, None, True))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            L.dca_gemm16_debug(1, prof.data_ptr())
            run()
            torch.cuda.synchronize()
            L.dca_gemm16_debug(1, 0)
            L.dca_gemm16_debug(2, 0)
            p = prof.cpu().numpy().reshape(per, grid, 4).astype(np.float64) / 100.0  # us
            ok = p[:, :, 3] > 0
            t0 = p[:, :, 0][ok].min()
            kl = (p[:, :, 1] - p[:, :, 0])[ok]
            tail = (p[:, :, 2] - p[:, :, 1])[ok]
            drain = (p[:, :, 3] - p[:, :, 2])[ok]
            # tails in flight at the same instant, sampled every 0.5 us
            ts = np.arange(t0, p[:, :, 3][ok].max(), 0.5)
            a, b = p[:, :, 1][ok], p[:, :, 3][ok]
            conc = np.array([np.sum((a <= t) & (b > t)) for t in ts])
            print(json.dumps({"k": k, "skip": with_skip, "skew_us": skew, "ms": round(ms, 4), "tiles": int(ok.sum()),
                              "kloop_us_p10_50_90": pct(kl), "tail_issue_us_p10_50_90": pct(tail),
                              "store_drain_us_p10_50_90": pct(drain), "span_us": round(float(p[:, :, 3][ok].max() - t0), 1),
                              "tails_in_flight_p50_90_max": [float(np.percentile(conc, 50)), float(np.percentile(conc, 90)), int(conc.max())]}))
    del x, w, skip
    torch.cuda.empty_cache()
_lib.gemm16_variant(2)
