"""Parity where the engine's hard paths run NATURALLY: no shrunken tiers, no forced ties — the
configured geometry, followed by the C++ oracle (oracle/dca_oracle.cpp restating astar.py:50-90,180-203,256-333 and
cpp/parallel_weighted_astar.cpp:169-330) iteration by iteration.

 (a) cube3, batch 20 000, weight 0.8 (configs[2]), both search semantics, 160 iterations: FRONT crosses its spill
     mark (96 batches), spills to BACK, refills from it, and the every-8th-iteration rebase runs 18+ times.
 (b) whole searches on shipped puzzle15 test states (1.5-2.8 M nodes), batch 10 000, the built-in Manhattan heuristic
     (integer costs: every f-level is one tie group, ordered by push count like astar.py:64-67): nodes generated,
     iterations and the move list equal the oracle's.
 (c) the AVI update at ONE RANK'S SHARE of configs[4]: 50 M puzzle48 states over 8 GPUs = 6.25 M states.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from deepcubea_amd import _lib
    _lib.require_gpu()
    return _lib


@pytest.fixture(scope="module")
def co():
    from oracle import c_oracle
    return c_oracle


# ------------------------------------------------------------------------------------------------ (a)
@pytest.mark.parametrize("sem", ["py", "cpp"])
def test_cube3_batch_20000_through_spill_refill_rebase(L, co, golden, sem):
    from deepcubea_amd.search_methods.engine import BwasEngine
    # KNUTH3: few exact float32 cost ties (what the cpp core's heap order is sensitive to); with it FRONT drains to its
    # refill mark at iteration ~136 (tools/tier_probe.py), so 160 iterations see the spill AND a refill from BACK
    B, w, hid, iters = 20000, 0.8, L.HEUR_KNUTH3, 160
    root = np.ascontiguousarray(golden["cube3_test_states"][0])
    semv, osem = (L.SEM_PY, co.SEM_PY) if sem == "py" else (L.SEM_CPP, co.SEM_CPP)
    ref = co.astar("cube3", root, w, B, osem, heur_builtin_id=hid, max_iters=iters, trace_cap=iters, stop_on_goal=False)
    assert ref["iterations"] == iters
    eng = BwasEngine("cube3", w, B, max_nodes=iters * B * 12 + (1 << 20), semantics=semv)
    eng.reset(root)
    if semv == L.SEM_PY:
        eng.root_commit(L.heuristic_builtin(hid, torch.from_numpy(root[None].copy()).cuda()))
    tr, back, thr = [], [], []
    for i in range(iters):
        # eager and graph-replayed iterations alternate in blocks (the same search either way)
        eng.run_builtin(hid, 1, use_graph=(i // 16) % 2 == 1)
        st = eng.status()
        assert not st["failed"], (i, st, eng.debug())
        tr.append((st["open_size"], st["closed_size"], st["nodes_generated"]))
        d = eng.debug()
        back.append(d["back_n"])
        thr.append(d["T"])
    tr, back, thr = np.array(tr, np.int64), np.array(back), np.array(thr)
    # the hard paths really ran: entries were spilled to BACK, and BACK gave entries back (the tier threshold rose again)
    assert back.max() > 0, "no spill happened"
    assert tr[:, 0].max() > 96 * B, "OPEN never outgrew the FRONT tier"
    finite = np.isfinite(thr)
    assert finite.any()
    rises = int((np.diff(thr[finite]) > 0).sum())
    print("sem=%s  |OPEN| end %d  BACK max %d  tier threshold rose %d times, fell %d times"
          % (sem, tr[-1, 0], back.max(), rises, int((np.diff(thr[finite]) < 0).sum())))
    assert rises >= 1, "no refill from BACK happened"
    assert len(tr) // 8 >= 18  # rebase iterations (every 8th) behind the ramp
    assert np.array_equal(tr[:, 2], ref["trace"][:, 2])  # nodes generated per iteration: exact in both semantics
    if sem == "py":
        assert np.array_equal(tr, ref["trace"])  # |OPEN|, |CLOSED| after every one of the 160 iterations
    else:
        # cpp: equal float32 costs pop in libstdc++'s heap order in the reference, in push order here
        rel = np.abs(tr[:, :2] - ref["trace"][:, :2]) / np.maximum(ref["trace"][:, :2], 1)
        assert rel.max() < 1e-2, rel.max()
    eng.close()


# ------------------------------------------------------------------------------------------------ (b)
@pytest.mark.parametrize("idx", [2, 7, 17])
def test_whole_puzzle15_searches_node_for_node(L, co, golden, idx):
    from deepcubea_amd.search_methods.engine import BwasEngine
    B, w, hid = 10000, 0.8, L.HEUR_MANHATTAN
    root = np.ascontiguousarray(golden["puzzle15_test_states"][idx])
    ref = co.astar("puzzle15", root, w, B, co.SEM_PY, heur_builtin_id=hid, trace_cap=4096)
    assert ref["solved"] and 1_000_000 < ref["nodes_generated"] < 3_000_000
    eng = BwasEngine("puzzle15", w, B, max_nodes=ref["nodes_generated"] + 8 * B * 4 + 64)
    for graph in (True, False):
        res = eng.solve_builtin(root, hid, chunk=16, use_graph=graph)
        assert res["solved"], res
        assert res["nodes_generated"] == ref["nodes_generated"] and res["iterations"] == ref["iterations"]
        assert res["nodes_expanded"] == ref["nodes_expanded"]
        assert res["moves"] == ref["moves"] and res["path_cost"] == ref["path_cost"]
    # and the per-iteration |OPEN| / |CLOSED| of the same search
    eng.reset(root)
    eng.root_commit(L.heuristic_builtin(hid, torch.from_numpy(root[None].copy()).cuda()))
    tr = []
    for _ in range(ref["iterations"]):
        eng.run_builtin(hid, 1)
        st = eng.status()
        tr.append((st["open_size"], st["closed_size"], st["nodes_generated"]))
    assert st["done"] and np.array_equal(np.array(tr, np.int64), ref["trace"])
    assert len(ref["moves"]) >= int(golden["puzzle15_test_opt_len"][idx])  # never shorter than the shipped optimum
    eng.close()


# ------------------------------------------------------------------------------------------------ (c)
@torch.no_grad()
def test_avi_update_one_ranks_share_of_50M_puzzle48_states(L, co):
    """configs[4]: states_per_update 50 M over 8 GPUs -> 6.25 M states on this rank (index0 = rank 3's offset, so the
    shard's RNG streams are the ones the 8-rank job would use).  Same size-independent properties as the 2^20 test."""
    from deepcubea_amd.updaters.updater import Updater
    from deepcubea_amd.utils import env_utils, nnet_utils
    from deepcubea_amd.utils.pytorch_models import FastResnet
    from deepcubea_amd.utils.pytorch_models import ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    env = env_utils.get_environment("puzzle48")
    net = ResnetModel(49, 49, 5000, 1000, 4, 1, True)  # n_puzzle.py:94-98
    load_synthetic_weights(net, 2026)
    fast = FastResnet(net.eval()).cuda()
    hfn = nnet_utils.get_heuristic_fn_dev(fast, clip_zero=True, batch_size=1 << 18)
    oh = None if fast.uses_l1_kernel else fast.onehot_dtype
    total, world, rank = 50_000_000, 8, 3
    n = total // world
    upd = Updater(env, total, 1000, hfn, 1, update_batch_size=1 << 18, seed=77, onehot_dtype=oh)
    upd.local_n, upd.index0, upd.rank = n, rank * n, rank
    sn, ctg, sv = upd.update_dev()
    assert sn.shape == (n, 49) and ctg.shape == (n, 1) and sv.shape == (n,)
    ctg = ctg[:, 0]
    goal = torch.tensor(np.concatenate((np.arange(1, 49), [0])).astype(np.uint8), device="cuda")
    is_goal = (sn == goal).all(dim=1)
    assert torch.equal(is_goal, sv.bool()) and int(is_goal.sum()) >= 1
    assert float(ctg[is_goal].abs().max()) == 0.0
    assert float(ctg[~is_goal].min()) >= 1.0 and bool(torch.isfinite(ctg).all())
    for lo in range(0, n, 1 << 20):  # every row a permutation of 0..48
        blk = sn[lo:lo + (1 << 20)]
        assert bool((torch.sort(blk.long(), dim=1).values == torch.arange(49, device="cuda")).all())
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(6))[:4096]
    st = sn[idx.cuda()].cpu().numpy()
    ch, _, _ = co.expand("puzzle48", st)
    hc = hfn(torch.from_numpy(ch.reshape(-1, 49)).cuda()).view(-1, 4)
    want = 1.0 + hc.min(dim=1).values
    want[torch.from_numpy((st == goal.cpu().numpy()).all(1)).cuda()] = 0.0
    assert float((want - ctg[idx.cuda()]).abs().max()) < 1e-4
    # the first 2^16 states of this shard regenerate bit-identically on their own (what a 763-rank job would hold)
    part = Updater(env, total, 1000, hfn, 1, update_batch_size=1 << 16, seed=77, onehot_dtype=oh)
    part.local_n, part.index0, part.rank = 1 << 16, rank * n, rank
    s2, c2, v2 = part.update_dev()
    assert torch.equal(s2, sn[:1 << 16]) and torch.equal(v2, sv[:1 << 16])
    assert float((c2[:, 0] - ctg[:1 << 16]).abs().max()) < 1e-4
