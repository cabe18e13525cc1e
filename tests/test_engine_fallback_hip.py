"""GPU: the exact refinement paths of the pop for bins too large for the LDS sort (massive cost ties).
Uniform-cost search (zero heuristic, weight 1) makes every node of a depth level tie on cost, so at depth 5
OPEN holds ~2e5 entries with ONE key and the engine must pick — and order — the batch by push order alone:
  * batch 50: the threshold bin's workgroup cuts 2e5 candidates down to 50 by radix refinement on the (key,id)
    composite, then orders them in LDS;
  * batch 9000 (> the 8192-entry LDS sort): the chosen entries are ordered through arithmetic sub-bins of their own
    exact composite range in global memory."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,scr", [(50, [3, 8, 1, 10, 6, 4]), (9000, [3, 8, 1, 10, 6, 4])])
def test_massive_ties_take_the_exact_refinement_path(B, scr):
    import torch
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods.engine import BwasEngine
    from oracle import c_oracle as co
    s = np.arange(54, dtype=np.uint8)[None]
    for a in scr:
        s = co.next_state("cube3", s, a)
    root = s[0]
    ref = co.astar("cube3", root, 1.0, B, co.SEM_PY, heur_builtin_id=3, max_iters=6000, trace_cap=6000)
    eng = BwasEngine("cube3", 1.0, B, max_nodes=1 << 23)
    eng.reset(root)
    eng.root_commit(torch.zeros(1, device="cuda"))
    big = 0
    for it in range(ref["iterations"]):
        eng.run_builtin(_lib.HEUR_ZERO, 1)
        st = eng.status()
        assert not st["failed"]
        assert (st["open_size"], st["closed_size"], st["nodes_generated"]) == tuple(ref["trace"][it]), it
        dbg = eng.debug()
        big = max(big, int(dbg["max_bin"]))
    res = eng._result()
    assert res["nodes_generated"] == ref["nodes_generated"] and bool(res["solved"]) == bool(ref["solved"])
    if ref["solved"]:  # batch 9000 reaches the goal (63 iterations); batch 50 is cut off at 6000 iterations like the oracle
        assert res["moves"] == ref["moves"]
    assert big > 131072, "the tie group never outgrew the LDS sort (%d)" % big
    assert int(eng.debug()["giant_bins_seen"]) > 0
    # the same search enqueued blind, as hipGraph replays
    r2 = eng.solve_builtin(root, _lib.HEUR_ZERO, max_iters=ref["iterations"], chunk=9, use_graph=True)
    assert r2["moves"] == ref["moves"] and r2["nodes_generated"] == ref["nodes_generated"]
    eng.close()
