"""GPU: the exact candidate-refinement fallback of the pop (threshold bin larger than the ordering buffers).
Uniform-cost search (zero heuristic, weight 1) makes every node of a depth level tie on cost, so at depth 5
OPEN holds ~2e5 entries with ONE key and the engine must pick the batch by push order alone."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_massive_ties_take_the_exact_fallback_path():
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods.engine import BwasEngine
    from oracle import c_oracle as co
    s = np.arange(54, dtype=np.uint8)[None]
    for a in [3, 8, 1, 10, 6, 4]:
        s = co.next_state("cube3", s, a)
    root = s[0]
    B = 50  # ordering buffers hold 2*50 + 131072 entries < the ~2e5-entry tie group
    ref = co.astar("cube3", root, 1.0, B, co.SEM_PY, heur_builtin_id=3, max_iters=6000, trace_cap=6000)
    eng = BwasEngine("cube3", 1.0, B, max_nodes=1 << 22)
    eng.reset(root)
    import torch
    eng.root_commit(torch.zeros(1, device="cuda"))
    big = 0
    for it in range(ref["iterations"]):
        eng.run_builtin(_lib.HEUR_ZERO, 1)
        st = eng.status()
        assert not st["failed"]
        assert (st["open_size"], st["closed_size"], st["nodes_generated"]) == tuple(ref["trace"][it]), it
        dbg = eng.debug()
        big = max(big, int(dbg["cand_n"]))
    assert big > 2 * B + 131072, "the tie group never outgrew the ordering buffers (%d)" % big
    eng.close()
