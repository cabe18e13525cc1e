"""Worker of tests/test_shared_gpu_hip.py: one PROCESS that searches a list of shipped puzzle15 test states to completion
with the built-in Manhattan heuristic (integer costs: every f-level is one tie group — the grid-wide refinement path of
k_sel_collect and its grid barriers) and writes what it found as JSON.  Run as a script; not collected by pytest."""
import json
import os
import sys
import time

import numpy as np


def main():
    out_path, batch, start_file = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    idxs = [int(x) for x in sys.argv[4:]]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch  # noqa: F401
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods.engine import BwasEngine
    g = np.load(os.path.join(root, "tests", "golden", "golden.npz"))
    states = g["puzzle15_test_states"]
    eng = BwasEngine("puzzle15", 0.8, batch, max_nodes=1 << 23)
    info0 = eng.info()
    # start together with the other process (both have their HIP context and engine by now)
    open(start_file + ".%d" % os.getpid(), "w").close()
    t0 = time.time()
    while time.time() - t0 < 120:
        if len([f for f in os.listdir(os.path.dirname(start_file)) if f.startswith(os.path.basename(start_file) + ".")]) >= 2:
            break
        time.sleep(0.01)
    res = []
    for i in idxs:
        t1 = time.time()
        r = eng.solve_builtin(np.ascontiguousarray(states[i]), _lib.HEUR_MANHATTAN, chunk=16, use_graph=True)
        res.append({"idx": i, "solved": bool(r["solved"]), "failed": int(r["failed"]), "nodes": int(r["nodes_generated"]),
                    "iterations": int(r["iterations"]), "moves": r["moves"], "seconds": time.time() - t1,
                    "giant_bins_seen": int(eng.debug()["giant_bins_seen"])})
    json.dump({"results": res, "info_before": info0, "info_after": eng.info()}, open(out_path, "w"))
    eng.close()


if __name__ == "__main__":
    main()
