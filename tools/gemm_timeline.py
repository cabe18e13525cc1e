#!/usr/bin/env python3
"""Where a tile's time goes inside dca_f16x3_gemm (variant 3): per workgroup the device wall clock at entry, when the first
operands have landed, at the end of the K loop and when the tile's stores are acknowledged (dca_f16x3_gemm_timeline).
Prints per (k, outputs) the median / p90 of: launch-to-first-operands, K loop, tail; the CU occupancy timeline (how many
tiles per CU slot, idle gaps between consecutive tiles on one CU) and how synchronised the tails of different CUs are.
    python tools/gemm_timeline.py [rows]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402
from deepcubea_amd.utils.pytorch_models import _pow2_scale, _split_f16  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
for n, k in ((1024, 1024), (1024, 5120)):
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(m, k, generator=g).cuda()
    w = torch.randn(n, k, generator=g) / k ** 0.5
    sc = _pow2_scale(w)
    wh, wl = _split_f16(w, sc)
    inv = (1.0 / sc).cuda()
    b = torch.randn(n, generator=g).cuda()
    skip = torch.randn(m, n, generator=g).cuda()
    planes = _lib.split_planes(x)
    whc, wlc = wh.cuda().contiguous(), wl.cuda().contiguous()
    blocks = ((m + 255) // 256 + 7) // 8 * 8 * ((n + 255) // 256)
    for name, sk, want_x in (("planes_only", None, False), ("skip_planes_x", skip, True)):
        _lib.f16x3_gemm_variant(3)
        for _ in range(3):
            _lib.f16x3_gemm(planes, whc, wlc, inv, 1.0, b, sk, True, True, want_x)
        stamps = torch.zeros((blocks, 8), dtype=torch.int64, device="cuda")
        _lib.check(_lib.lib().dca_f16x3_gemm_timeline(C.c_void_p(stamps.data_ptr())), "timeline")
        torch.cuda.synchronize()
        _lib.f16x3_gemm(planes, whc, wlc, inv, 1.0, b, sk, True, True, want_x)
        torch.cuda.synchronize()
        _lib.check(_lib.lib().dca_f16x3_gemm_timeline(C.c_void_p(0)), "timeline")
        s = stamps.cpu().numpy()
        s = s[s[:, 0] != 0]
        t0 = s[:, 0].min()
        us = (s[:, :4] - t0) / 100.0  # 100 MHz -> microseconds
        fill, loop, tail = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
        issue = (s[:, 5] - t0) / 100.0 - us[:, 2]   # K loop end -> wave 0 has issued its last store
        drain = us[:, 3] - (s[:, 5] - t0) / 100.0    # ... -> its stores are acknowledged
        ghz = s[:, 6] / np.maximum(loop, 1e-3) / 1000.0  # shader cycles per microsecond of K loop -> GHz
        # matrix-pipe cycles the K loop needs per SIMD: 2 waves x 48 MFMAs of 32x32x16 (8 passes x 4 cycles) per 32-deep K-step
        need = (k // 32) * 2 * 48 * 32
        util = need / np.maximum(s[:, 6], 1)
        hw = s[:, 4]
        cu_key = (hw >> 32) * 100000 + ((hw & 0xFFFFFFFF) >> 8 & 0xF) * 1000 + ((hw & 0xFFFFFFFF) >> 12 & 0x3) * 100 + ((hw & 0xFFFFFFFF) >> 13 & 0x7) * 0  # xcc, cu id, sh id (HW_ID layout varies: the key only has to separate CUs)
        cu_key = (hw >> 32) * 4096 + ((hw & 0xFFFFFFFF) >> 8 & 0xFFF)
        gaps, per_cu = [], []
        for key in np.unique(cu_key):
            rows = us[cu_key == key]
            rows = rows[np.argsort(rows[:, 0])]
            per_cu.append(len(rows))
            gaps += list(rows[1:, 0] - rows[:-1, 3])
        # how many tiles are in their tail at the same instant (sampled every microsecond over the launch)
        end = us[:, 3].max()
        grid = np.arange(0.0, end, 1.0)
        in_tail = ((us[:, 2][None, :] <= grid[:, None]) & (grid[:, None] < us[:, 3][None, :])).sum(axis=1)
        in_loop = ((us[:, 1][None, :] <= grid[:, None]) & (grid[:, None] < us[:, 2][None, :])).sum(axis=1)
        q = lambda a: [round(float(np.percentile(a, p)), 2) for p in (10, 50, 90)]  # noqa: E731
        print(json.dumps({"m": m, "n": n, "k": k, "tail_form": name, "tiles": int(len(s)), "launch_us": round(float(end), 1),
                          "fill_us_p10_50_90": q(fill), "kloop_us_p10_50_90": q(loop), "tail_us_p10_50_90": q(tail),
                          "kloop_clock_ghz_p10_50_90": q(ghz), "kloop_mfma_cycle_utilisation_p10_50_90": q(util),
                          "tail_issue_us_p10_50_90": q(issue), "tail_store_drain_us_p10_50_90": q(drain),
                          "cu_slots": int(len(per_cu)), "tiles_per_cu_slot_min_max": [int(min(per_cu)), int(max(per_cu))],
                          "gap_between_tiles_on_a_cu_us_p10_50_90": q(np.array(gaps)) if gaps else None,
                          "tiles_in_tail_at_once_p10_50_90_max": q(in_tail) + [int(in_tail.max())],
                          "tiles_in_kloop_at_once_p10_50_90": q(in_loop)}))
    del x, planes, skip
    torch.cuda.empty_cache()
