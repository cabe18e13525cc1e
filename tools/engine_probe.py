#!/usr/bin/env python3
"""Per-iteration view of the engine at the bench geometry: pop diagnostics (bins handed to k_rank, largest bin, FRONT
size) next to the device-side launch spans of the same iteration.  python tools/engine_probe.py [env] [B] [warm] [n] [knob=value ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402
from deepcubea_amd.search_methods.engine import BwasEngine  # noqa: E402

env = sys.argv[1] if len(sys.argv) > 1 else "cube3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 60
n = int(sys.argv[4]) if len(sys.argv) > 4 else 24
late = []
for kv in sys.argv[5:]:  # knob=value pairs for dca_debug_tune (diagnostics); @knob=value: only after the warm-up
    k, v = kv.lstrip("@").split("=")
    if kv.startswith("@"):
        late.append((int(k), int(v)))
    else:
        _lib.check(_lib.lib().dca_debug_tune(int(k), int(v)), "dca_debug_tune")
A = 12 if env == "cube3" else 4
g = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
root = np.ascontiguousarray(g[env + "_test_states"][0])
eng = BwasEngine(env, 0.8, B, max_nodes=(warm + n + 40) * B * A + (1 << 16))
eng.reset(root)
eng.root_commit(_lib.heuristic_builtin(2, torch.from_numpy(root[None].copy()).cuda()))
eng.run_builtin(2, warm, use_graph=True)
for k, v in late:
    _lib.check(_lib.lib().dca_debug_tune(k, v), "dca_debug_tune")
for it in range(n):
    prof = eng.profile_builtin(2, 1, use_graph=True)
    d = eng.debug()
    row = {"it": warm + it, "front_n": int(d["front_n"]), "back_n": int(d["back_n"]), "bstar": int(d["bstar"]),
           "n_ord": int(d["n_ord"]), "max_bin": int(d["max_bin"]), "max_sub": int(d["max_sub"]),
           "span_us": {k: round(v * 1e3, 1) for k, v in prof["span_ms"].items()},
           "gap_us": {k: round(v * 1e3, 1) for k, v in prof["gap_ms"].items()}}
    print(json.dumps(row))
eng.close()
