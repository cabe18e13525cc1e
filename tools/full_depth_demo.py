#!/usr/bin/env python3
"""Full-depth searches on REAL test scrambles through the `--language hip` CLI: the shipped
data/<env>/test states (kept as fixtures in tests/golden/golden.npz) solved to completion with the built-in admissible
Manhattan heuristic, nodes/s over the WHOLE search (OPEN growth, refills and ties included), and the reference's
compare_solutions report against the optimal lengths shipped with the test set — the nearest available stand-in for the
results/ length-parity target while the trained weights are absent from the mount.

    python tools/full_depth_demo.py puzzle15 100 0.8 10000 py  [max_nodes]
"""
import json
import os
import pickle
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd.environments.n_puzzle import NPuzzleState  # noqa: E402
from deepcubea_amd.search_methods import astar  # noqa: E402
from deepcubea_amd.utils import compare_solutions as cs  # noqa: E402
from deepcubea_amd.utils import data_utils  # noqa: E402

env = sys.argv[1] if len(sys.argv) > 1 else "puzzle15"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w = sys.argv[3] if len(sys.argv) > 3 else "0.8"
B = sys.argv[4] if len(sys.argv) > 4 else "10000"
sem = sys.argv[5] if len(sys.argv) > 5 else "py"
max_nodes = sys.argv[6] if len(sys.argv) > 6 else "auto"
g = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
states = g[env + "_test_states"][:n]
opt = g[env + "_test_opt_len"][:n] if env + "_test_opt_len" in g.files else None
tmp = tempfile.mkdtemp()
spath = os.path.join(tmp, "data_0.pkl")
pickle.dump({"states": [NPuzzleState(s.copy()) for s in states]}, open(spath, "wb"))
rdir = os.path.join(tmp, "res")
t0 = time.time()
astar.main(["--states", spath, "--model_dir", "builtin:manhattan", "--env", env, "--weight", w, "--batch_size", B,
            "--results_dir", rdir, "--language", "hip", "--semantics", sem, "--max_nodes", max_nodes, "--debug"])
wall = time.time() - t0
res = data_utils.load_pickle(os.path.join(rdir, "results.pkl"))
lens = np.array([len(s) for s in res["solutions"]])
nodes = np.array(res["num_nodes_generated"], np.float64)
times = np.array(res["times"], np.float64)
A = 4
summary = {"env": env, "states": int(n), "weight": float(w), "batch_size": int(B), "semantics": sem,
           "heuristic": "built-in Manhattan distance (admissible, consistent), evaluated inside the expansion launch",
           "total_nodes_generated": float(nodes.sum()), "total_search_seconds": float(times.sum()), "wall_seconds": wall,
           "nodes_generated_per_s_whole_search": float(nodes.sum() / times.sum()),
           "nodes_expanded_per_s_whole_search": float(nodes.sum() / A / times.sum()),
           "max_nodes_one_state": float(nodes.max()), "mean_len": float(lens.mean())}
print("\nSUMMARY " + json.dumps(summary))
if opt is not None:
    ref = {"lens": opt.astype(np.int64), "times": np.ones(n), "num_nodes_generated": np.ones(n)}
    mine = {"lens": lens, "times": times, "num_nodes_generated": nodes}
    print(cs.format_report(cs.compare(ref, mine)))
    print("solution lengths vs the shipped optimal lengths: %d/%d optimal, mean excess %.3f moves, none shorter than optimal: %s"
          % (int((lens == opt).sum()), n, float((lens - opt).mean()), bool((lens >= opt).all())))
