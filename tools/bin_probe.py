#!/usr/bin/env python3
"""Per-iteration selection geometry of a long cube3 search (batch 20 000): FRONT size, threshold bin index, entries handed to
k_rank, largest bin among them — with the fine binning of the plain iterations on and off.   python tools/bin_probe.py [iters]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402
from deepcubea_amd.search_methods.engine import BwasEngine  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
g = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
root = np.ascontiguousarray(g["cube3_test_states"][0])
for fine_off in (0, 1):
    _lib.check(_lib.lib().dca_debug_tune(8, fine_off), "tune")
    eng = BwasEngine("cube3", 0.8, 20000, max_nodes=iters * 240000 + (1 << 20))
    eng.reset(root)
    eng.root_commit(_lib.heuristic_builtin(2, torch.from_numpy(root[None].copy()).cuda()))
    rows = []
    for i in range(iters):
        eng.run_builtin(2, 1)
        d = eng.debug()
        rows.append((i, int(d["front_n"]), int(d["back_n"]), int(d["bstar"]), int(d["n_ord"]), int(d["max_bin"])))
    print("fine binning", "OFF" if fine_off else "ON")
    for r in rows[40:]:
        print("  it %3d front %8d back %9d bstar %4d n_ord %6d max_bin %5d%s" % (r + (" (rebase)" if r[0] % 8 == 0 else "",)))
    eng.close()
