#!/usr/bin/env python3
"""cube3 searches of the PUBLISHED SIZE through the `--language hip` CLI (VERDICT r02 missing #1: results/cube3 reaches 6.1e7
nodes per state, and no cube3 search of that size had run on the engine — the trained weights that make the shipped 18-26-move
test scrambles tractable are not in the mount).  Stand-in: scrambles of `depth` random moves from the goal, solved by
uniform-cost BWAS (built-in zero heuristic, weight 1, batch 10 000 = the train.sh batch): every f-level is ONE tie group of up
to tens of millions of entries ordered by push count (astar.py:64-67) — the grid-wide tie refinement on cube3's 54-byte rows
and 12 moves — and the answer is checkable: the solution is valid and no longer than the scramble.

    python tools/cube3_depth_demo.py [n_states] [depth] [batch]
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
from deepcubea_amd.environments.cube3 import Cube3State  # noqa: E402
from deepcubea_amd.search_methods import astar  # noqa: E402
from deepcubea_amd.utils import data_utils, env_utils  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 7
B = sys.argv[3] if len(sys.argv) > 3 else "10000"
env = env_utils.get_environment("cube3")
rng = np.random.default_rng(2026)
states = []
for _ in range(n):
    st = env.generate_goal_states(1)[0]
    last = -1
    for _ in range(depth):
        a = int(rng.integers(0, 12))
        while a // 2 == last // 2 and last >= 0:  # never turn the same face twice in a row: the scramble stays `depth` deep-ish
            a = int(rng.integers(0, 12))
        st = env.next_state([st], a)[0][0]
        last = a
    states.append(Cube3State(st.colors.copy()))
tmp = tempfile.mkdtemp()
spath = os.path.join(tmp, "data_0.pkl")
pickle.dump({"states": states}, open(spath, "wb"))
rdir = os.path.join(tmp, "res")
t0 = time.time()
astar.main(["--states", spath, "--model_dir", "builtin:zero", "--env", "cube3", "--weight", "1.0", "--batch_size", B,
            "--results_dir", rdir, "--language", "hip", "--semantics", "py", "--max_nodes", "auto", "--debug"])
wall = time.time() - t0
res = data_utils.load_pickle(os.path.join(rdir, "results.pkl"))
lens = np.array([len(s) for s in res["solutions"]])
nodes = np.array(res["num_nodes_generated"], np.float64)
times = np.array(res["times"], np.float64)
assert (lens <= depth).all(), lens
print("\nSUMMARY " + json.dumps({
    "env": "cube3", "states": n, "scramble_depth": depth, "batch_size": int(B), "weight": 1.0,
    "heuristic": "built-in zero (uniform-cost search: every f-level one tie group)", "solution_lengths": lens.tolist(),
    "nodes_generated": nodes.tolist(), "seconds": [round(float(t), 3) for t in times],
    "total_nodes_generated": float(nodes.sum()), "total_search_seconds": float(times.sum()), "wall_seconds": wall,
    "nodes_generated_per_s_whole_search": float(nodes.sum() / times.sum()),
    "nodes_expanded_per_s_whole_search": float(nodes.sum() / 12 / times.sum()), "max_nodes_one_state": float(nodes.max())}))
