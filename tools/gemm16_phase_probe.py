#!/usr/bin/env python3
"""Where a K-tile of the 8-phase dca_gemm16 schedule spends its cycles: s_memtime stamps of K-tiles 6..9 in workgroup 0 (one
wave of each wave row), from the diagnostic build (dca_debug_gemm16_profile), on a full-chip problem.  Shader cycles.
    python tools/gemm16_phase_probe.py [rows] [k]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
n = 1024
g = torch.Generator().manual_seed(5)
x = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).cuda()
w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).cuda()
b = torch.randn(n, generator=g).cuda()
_lib.gemm16_variant(2)
for _ in range(3):
    _lib.gemm16(x, w, b, None, True)
st = torch.zeros(128, dtype=torch.int64, device="cuda")
_lib.gemm16_phase_stamps(st)
for rep in range(3):
    st.zero_()
    _lib.gemm16(x, w, b, None, True)
    torch.cuda.synchronize()
    s = st.cpu().view(2, 4, 4, 4)  # [wave row][tile][phase][point]
    print("launch %d" % rep)
    for g_ in range(2):
        t0 = int(s[g_, 0, 0, 0])
        print("  wave row %d: K-tile times %s" % (g_, [int(s[g_, i + 1, 0, 0] - s[g_, i, 0, 0]) for i in range(3)]))
        for i in range(3):
            row = []
            for ph in range(4):
                e, r, b1, mm = (int(s[g_, i, ph, q]) for q in range(4))
                nxt = int(s[g_, i, ph + 1, 0]) if ph < 3 else int(s[g_, i + 1, 0, 0])
                row.append("P%d issue+vmcnt %4d | lgkm+bar %4d | mfma %4d | bar %4d" % (ph + 1, r - e, b1 - r, mm - b1, nxt - mm))
            print("    tile %d (entry @%d): " % (i, int(s[g_, i, 0, 0]) - t0) + "\n" + "\n".join("      " + r for r in row))
_lib.gemm16_phase_stamps(None)
