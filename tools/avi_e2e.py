#!/usr/bin/env python3
"""update -> train -> search, end to end on one MI355X, with the repo's own heuristic: the reference's
checkpoints are not in the mount, so the parity half of the metric ("solution lengths matching results/<env>/") has never had a
network-driven datum.  This runs `ctg_approx/avi.py` (GBFS updates on the device, `train_nnet`, target hand-over) on puzzle15
(or cube3) for a fixed wall-time budget, then `search_methods/astar.py --language hip` with the reference's own search settings
(train.sh:21 / train.sh:9) on the first N shipped `data/<env>/test` states with the network that came out, and prints the reference's compare_solutions
report against (a) the optimal lengths shipped with the test set and (b) the published per-state results of the
reference's fully trained network (results/puzzle15/output.txt, kept as a fixture).

    python tools/avi_e2e.py [train_seconds] [n_states] [states_per_update] [save_dir] [epochs_per_update] [env]

env = puzzle15 (default; train.sh:18,21: loss threshold 0.1, back_max 500, search weight 0.8, batch 20 000) or cube3
(train.sh:4,9: loss threshold 0.06, back_max 30, search weight 0.6, batch 10 000 — the north star's own configuration; the
reference trained it for 1.2 M iterations, saved_models/cube3/output.txt).

Environment switches: DCA_E2E_MAX_NODES (node ids per search), DCA_E2E_CHUNK (states per CLI call, 20), DCA_E2E_DEADLINE (seconds
since the start after which no further search call is made), DCA_E2E_EXPORT=path (+ DCA_E2E_EXPORT_FP32) / DCA_E2E_IMPORT=path
(carry the trained network out of / into a GPU-box visit: `train_seconds` 0 + an import = search only), DCA_E2E_WEIGHT,
DCA_E2E_STATES=i,j,... (explicit test-set indices).

Seeds are fixed (torch / numpy / random / the device state generator), so a rerun on the same build repeats the schedule up to
the order of floating-point atomics."""
import json
import os
import pickle
import random
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd.ctg_approx import avi  # noqa: E402
from deepcubea_amd.environments.cube3 import Cube3State  # noqa: E402
from deepcubea_amd.environments.n_puzzle import NPuzzleState  # noqa: E402
from deepcubea_amd.search_methods import astar  # noqa: E402
from deepcubea_amd.utils import compare_solutions as cs  # noqa: E402
from deepcubea_amd.utils import data_utils  # noqa: E402

env = sys.argv[6] if len(sys.argv) > 6 else "puzzle15"
LOSS_THRESH, BACK_MAX, WEIGHT, SEARCH_B, StateCls = {"puzzle15": ("0.1", "500", "0.8", "20000", NPuzzleState),
                                                     "cube3": ("0.06", "30", "0.6", "10000", Cube3State)}[env]
WEIGHT = os.environ.get("DCA_E2E_WEIGHT", WEIGHT)  # (a search-only rerun of an exported network at another path-cost weight)
train_s = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
spu = int(sys.argv[3]) if len(sys.argv) > 3 else 3_000_000
save = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else tempfile.mkdtemp()
epochs = int(sys.argv[5]) if len(sys.argv) > 5 else 1
torch.manual_seed(0)
np.random.seed(0)
random.seed(0)
B = 10000  # train.sh:18
t0 = time.time()
# the reference's own line (train.sh:18: 50 M states per update, 5000 steps of 10 000) scaled to the budget: `spu` states and
# spu / 10 000 steps per update — value iteration moves the cost-to-go frontier about one move per update, so the number of
# updates is what the budget has to buy (the reference ran ~200)
IMPORT, EXPORT = os.environ.get("DCA_E2E_IMPORT"), os.environ.get("DCA_E2E_EXPORT")
if IMPORT:  # a network exported by an earlier run (fp16 copy of the state dict): search only, or train on from it
    cur = os.path.join(save, env, "current")
    os.makedirs(cur, exist_ok=True)
    blob = torch.load(IMPORT, map_location="cpu")
    torch.save({k: (v.float() if v.is_floating_point() else v) for k, v in blob["state_dict"].items()}, os.path.join(cur, "model_state_dict.pt"))
    pickle.dump(int(blob["train_itr"]), open(os.path.join(cur, "train_itr.pkl"), "wb"), protocol=-1)
    pickle.dump(int(blob["update_num"]), open(os.path.join(cur, "update_num.pkl"), "wb"), protocol=-1)
    print("imported %s: %d iterations, %d target updates" % (IMPORT, blob["train_itr"], blob["update_num"]))
    if train_s > 0:  # training on: the imported network is also the target (the reference's manual `cp current/* target/`, train.sh:5)
        import shutil
        shutil.copytree(cur, os.path.join(save, env, "target"), dirs_exist_ok=True)
if train_s > 0:
    avi.main(["--env", env, "--states_per_update", str(spu), "--batch_size", str(B), "--nnet_name", env, "--max_itrs", "100000000",
              "--loss_thresh", LOSS_THRESH, "--back_max", BACK_MAX, "--num_test", "1000", "--save_dir", save, "--max_seconds",
              str(train_s), "--update_nnet_batch_size", "100000", "--epochs_per_update", str(epochs), "--seed", "0", "--debug"])
if EXPORT:
    cur = os.path.join(save, env, "current")
    sd = torch.load(os.path.join(cur, "model_state_dict.pt"), map_location="cpu")
    keep32 = os.environ.get("DCA_E2E_EXPORT_FP32") is not None  # (59 MB for the cube3 network; fp16: 29 MB, weights rounded)
    torch.save({"state_dict": {k: (v.half() if v.is_floating_point() and not keep32 else v) for k, v in sd.items()},
                "train_itr": pickle.load(open(os.path.join(cur, "train_itr.pkl"), "rb")),
                "update_num": pickle.load(open(os.path.join(cur, "update_num.pkl"), "rb"))}, EXPORT)
train_wall = time.time() - t0
itr = pickle.load(open(os.path.join(save, env, "current", "train_itr.pkl"), "rb"))
upd = pickle.load(open(os.path.join(save, env, "current", "update_num.pkl"), "rb"))
print("\nTRAINED %s" % json.dumps({"seconds": round(train_wall, 1), "train_iterations": int(itr), "target_updates": int(upd),
                                  "states_per_update": spu, "epochs_per_update": epochs,
                                  "adam_steps_per_update": epochs * -(-spu // B), "batch_size": B}))

g = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
sel = np.arange(n)
if os.environ.get("DCA_E2E_STATES"):  # explicit test-set indices instead of the first n (e.g. a state an earlier run left unsolved)
    sel = np.array([int(v) for v in os.environ["DCA_E2E_STATES"].split(",")])
    n = len(sel)
    print("test-set states %s" % sel.tolist())
states = g[env + "_test_states"][sel]
opt = g[env + "_test_opt_len"][sel].astype(np.int64)
pub = {"lens": g["published_%s_len" % env][sel].astype(np.int64), "times": g["published_%s_time" % env][sel].astype(np.float64),
       "num_nodes_generated": g["published_%s_nodes" % env][sel].astype(np.float64)}
tmp = tempfile.mkdtemp()
spath = os.path.join(tmp, "data_0.pkl")
pickle.dump({"states": [StateCls(s.copy()) for s in states]}, open(spath, "wb"))
rdir = os.path.join(tmp, "res")
t1 = time.time()
MAXN = os.environ.get("DCA_E2E_MAX_NODES", "300000000")  # ids per search: a heuristic too weak for a state fails it instead of running on


def search(path, out):
    astar.main(["--states", path, "--model_dir", os.path.join(save, env, "current"), "--env", env, "--weight", WEIGHT, "--batch_size",
                SEARCH_B, "--results_dir", out, "--language", "hip", "--nnet_batch_size", "10000", "--max_nodes", MAXN, "--debug"])
    return data_utils.load_pickle(os.path.join(out, "results.pkl"))


# chunks of CHUNK states per CLI call (searched side by side as engine instances); a chunk with a state the network cannot solve
# inside MAXN nodes is redone state by state and keeps what solves.  DCA_E2E_DEADLINE (seconds since the tool started) stops
# the search between calls: what was searched by then is what is reported.
CHUNK = int(os.environ.get("DCA_E2E_CHUNK", "20"))
DEADLINE = float(os.environ.get("DCA_E2E_DEADLINE", "0"))
res = {"solutions": [], "times": [], "num_nodes_generated": []}
solved_idx, tried = [], 0


def out_of_time():
    return DEADLINE > 0 and time.time() - t0 > DEADLINE


def dump(idx, path):
    pickle.dump({"states": [StateCls(states[i].copy()) for i in idx]}, open(path, "wb"))


for c0 in range(0, n, CHUNK):
    if out_of_time():
        print("deadline reached after %d of %d states" % (tried, n))
        break
    idx = list(range(c0, min(c0 + CHUNK, n)))
    cp_path = os.path.join(tmp, "chunk_%d.pkl" % c0)
    dump(idx, cp_path)
    try:
        r = search(cp_path, os.path.join(tmp, "res_chunk_%d" % c0))
        solved_idx += idx
        tried += len(idx)
        for k in res:
            res[k] += list(r[k])
        continue
    except Exception as e:  # noqa: BLE001
        print("chunk %d-%d stopped (%s): state by state" % (idx[0], idx[-1], str(e)[:160]))
    for i in idx:
        if out_of_time():
            break
        sp_i = os.path.join(tmp, "one_%d.pkl" % i)
        dump([i], sp_i)
        tried += 1
        try:
            r = search(sp_i, os.path.join(tmp, "res_%d" % i))
        except Exception as e2:  # noqa: BLE001
            print("state %d: not solved within %s nodes (%s)" % (i, MAXN, str(e2)[:120]))
            continue
        solved_idx.append(i)
        for k in res:
            res[k] += list(r[k])
print("solved %d of the %d states tried (of %d): %s" % (len(solved_idx), tried, n, solved_idx))
if not solved_idx:
    print("NO shipped state solved inside the node budget: the GBFS test lines above (Back Steps / %Solved) say how deep the network solves")
    sys.exit(0)
opt = opt[solved_idx]
pub = {k: v[solved_idx] for k, v in pub.items()}
n = len(solved_idx)
search_wall = time.time() - t1
mine = {"lens": np.array([len(s) for s in res["solutions"]]), "times": np.array(res["times"], np.float64),
        "num_nodes_generated": np.array(res["num_nodes_generated"], np.float64)}
print("\nSEARCH %s" % json.dumps({"states": n, "weight": float(WEIGHT), "batch_size": int(SEARCH_B), "wall_seconds": round(search_wall, 2),
                                 "mean_len": float(mine["lens"].mean()), "mean_nodes": float(mine["num_nodes_generated"].mean()),
                                 "mean_seconds_per_state": float(mine["times"].mean())}))
print("\n==== vs the OPTIMAL lengths shipped with data/%s/test (soln1 = optimal, soln2 = this run)" % env)
print(cs.format_report(cs.compare({"lens": opt, "times": np.ones(n), "num_nodes_generated": np.ones(n)}, mine)))
print("optimal on %d of %d states, mean excess %.3f moves, none shorter than optimal: %s"
      % (int((mine["lens"] == opt).sum()), n, float((mine["lens"] - opt).mean()), bool((mine["lens"] >= opt).all())))
print("\n==== vs the PUBLISHED results of the reference's fully trained network (results/%s/output.txt; soln1 = published)" % env)
print(cs.format_report(cs.compare(pub, mine)))
print("published: optimal on %d of %d, mean excess %.3f moves" % (int((pub["lens"] == opt).sum()), n, float((pub["lens"] - opt).mean())))
