"""GPU: real test-set instances solved end to end.  With an admissible, consistent heuristic (Manhattan distance,
DCA_HEUR_MANHATTAN), weight 1 and the reference C++ core's deferred-termination rule
(parallel_weighted_astar.cpp:205-208) BWAS returns OPTIMAL solutions, so the engine's solution lengths must equal the
optimal lengths shipped with data/puzzle15/test (kept as a fixture).  Integer costs make
every f-level one giant tie group: this also hammers the exact tie handling of the pop."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env,count,batch", [("puzzle15", 12, 10000)])  # (24-puzzle needs >1e9 nodes with Manhattan)
def test_optimal_lengths_on_shipped_test_states(golden, env, count, batch):
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods.engine import BwasEngine
    from oracle import c_oracle as co
    states = golden[env + "_test_states"]
    opt = golden[env + "_test_opt_len"]
    order = np.argsort(opt, kind="stable")[:count]  # the shortest instances of the set
    eng = BwasEngine(env, 1.0, batch, max_nodes=1 << 26, semantics=_lib.SEM_CPP)
    for i in order:
        root = states[i]
        h0 = co.heur_builtin(4, root[None])[0]
        assert h0 <= opt[i]                                  # admissible on this instance
        res = eng.solve_builtin(root, _lib.HEUR_MANHATTAN, max_iters=20000, chunk=64, use_graph=True)
        assert res["solved"], (env, i, res)
        assert len(res["moves"]) == opt[i], (env, i, len(res["moves"]), opt[i], res["nodes_generated"])
        s = root[None].copy()
        for a in res["moves"]:
            s = co.next_state(env, s, a)
        assert co.is_solved(env, s)[0]
    eng.close()
