#!/usr/bin/env python3
"""Which configurations make the OPEN tiers spill AND refill on their own?  Runs the engine (no oracle) and prints, per
configuration, how often the tier threshold fell (spill) and rose (refill from BACK) and the giant-bin count.

    python tools/tier_probe.py [iters]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402
from deepcubea_amd.search_methods.engine import BwasEngine  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
CASES = [("cube3", 0, 0.8, 20000, 2), ("cube3", 0, 0.8, 20000, 1), ("cube3", 0, 0.2, 20000, 2), ("cube3", 0, 1.0, 20000, 2),
         ("cube3", 3, 0.6, 10000, 1), ("puzzle48", 3, 0.6, 20000, 1), ("puzzle24", 5, 0.6, 10000, 4), ("puzzle35", 2, 0.8, 20000, 1)]
for env, idx, w, B, hid in CASES:
    A = 12 if env == "cube3" else 4
    root = np.ascontiguousarray(g[env + "_test_states"][idx])
    eng = BwasEngine(env, w, B, max_nodes=iters * B * A + (1 << 20))
    eng.reset(root)
    eng.root_commit(_lib.heuristic_builtin(hid, torch.from_numpy(root[None].copy()).cuda()))
    thr, back, done_at = [], [], None
    for i in range(iters):
        eng.run_builtin(hid, 1)
        d = eng.debug()
        thr.append(d["T"])
        back.append(d["back_n"])
        if eng.status()["done"]:
            done_at = i
            break
    thr, back = np.array(thr), np.array(back)
    fin = np.isfinite(thr)
    df = np.diff(thr[fin])
    drops = int((np.diff(back) < 0).sum())
    st = eng.status()
    print("%-9s state %d w %.1f B %5d heur %d: iters %d done_at %s |OPEN| %d BACK max %d  T fell %d rose %d  BACK shrank %d times  "
          "first rise at %s  giant bins %d" % (env, idx, w, B, hid, len(thr), done_at, st["open_size"], back.max(),
                                               int((df < 0).sum()), int((df > 0).sum()), drops,
                                               (np.nonzero(df > 0)[0][:3] + int(np.nonzero(fin)[0][0]) + 1).tolist() if (df > 0).any() else None,
                                               int(eng.debug()["giant_bins_seen"])), flush=True)
    eng.close()
    del eng
    torch.cuda.empty_cache()
