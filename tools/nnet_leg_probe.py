#!/usr/bin/env python3
"""Where does a network-in-the-loop iteration spend its time?  Splits bench.py's end_to_end_nnet step (cube3, batch 20 000,
dedup-first stepping + FastResnet) into the engine half and the network half.   python tools/nnet_leg_probe.py [fp32|bf16]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402
from deepcubea_amd.search_methods.engine import BwasEngine  # noqa: E402
from deepcubea_amd.utils import env_utils, nnet_utils  # noqa: E402
from deepcubea_amd.utils.pytorch_models import FastResnet  # noqa: E402
from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights  # noqa: E402

dtn = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[dtn]
env = env_utils.get_environment("cube3")
model = env.get_nnet_model()
load_synthetic_weights(model, 2024)
fast = FastResnet(model, dt).cuda()
hfn = nnet_utils.get_heuristic_fn_dev(fast, clip_zero=False, batch_size=245760)
B = 20000
g = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
root = np.ascontiguousarray(g["cube3_test_states"][0])
eng = BwasEngine("cube3", 0.8, B, max_nodes=40 * B * 12, packed=True)
eng.reset(root)
eng.root_commit(hfn(eng.root_nnet_in()))
for _ in range(64):
    eng.run_builtin(_lib.HEUR_HASHU01, 1)
    if eng.status()["open_size"] >= 3 * B:
        break


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


t_pop, t_net, t_com, rows_all = [], [], [], []
for it in range(12):
    t0 = sync()
    nn, oh, _, rows = eng.pop_expand_packed()
    t1 = sync()
    n = min((rows + 1023) // 1024 * 1024, eng.packed_capacity)
    h = hfn(nn[:n])
    t2 = sync()
    eng.commit_packed(h.float().contiguous())
    t3 = sync()
    t_pop.append(t1 - t0), t_net.append(t2 - t1), t_com.append(t3 - t2), rows_all.append(rows)
    d = eng.debug()
    print("it %2d rows %6d pop+expand+dedup+pack %.3f ms  network %.3f ms  commit %.3f ms  | n_ord %d max_bin %d giant_seen %d front %d"
          % (it, rows, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, d["n_ord"], d["max_bin"], d["giant_bins_seen"], d["front_n"]), flush=True)
print("fallbacks", fast.split_fallbacks, "h range", float(h[:rows].min()), float(h[:rows].max()))
x = nn[:196608].clone()
for _ in range(3):
    hfn(x)
t0 = sync()
for _ in range(5):
    hfn(x)
print("network alone on 196608 rows: %.3f ms" % ((sync() - t0) / 5 * 1e3))
