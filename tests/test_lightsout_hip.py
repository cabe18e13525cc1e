"""GPU: LightsOut (lightsout7) on the device path — env kernels, both BWAS semantics on the engine, the Environment mirror,
the scrambler and the AVI update step — against fixtures recorded from the reference's environments/lights_out.py and
search_methods/astar.py (tests/golden/lightsout.npz) and against the oracle (itself pinned to those fixtures and to the
reference's cpp/environments.cpp in tests/test_lightsout_cpu.py).  Integer / byte work: everything is compared exactly."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from deepcubea_amd import _lib
    _lib.require_gpu()
    return _lib


@pytest.fixture(scope="module")
def co():
    from oracle import c_oracle
    return c_oracle


@pytest.fixture(scope="module")
def lo():
    return np.load(os.path.join(ROOT, "tests", "golden", "lightsout.npz"))


def test_env_kernels_vs_reference_fixture_and_oracle(L, co, lo):
    from oracle import np_oracle as no
    e, d, D, A, depth = L.env_ids("lightsout7")
    assert (D, A, depth) == (49, 49, 6)
    S = torch.from_numpy(lo["states"]).cuda()
    for a in range(49):
        got = L.next_state(e, d, S, a)
        assert np.array_equal(got.cpu().numpy(), lo["next_state_all_actions"][a]), a
        assert torch.equal(L.next_state(e, d, got, a, prev=True), S)  # prev_state == next_state
    rng = np.random.default_rng(3)
    for n in (1, 7, 64, 65, 1000, 4099):  # ragged sizes around the 64-parent tile
        X = rng.integers(0, 2, size=(n, 49)).astype(np.uint8)
        if n > 2:
            X[0] = 0
            X[1] = co.next_state("lightsout7", X[:1], 24)[0]
        out = L.expand_fused(e, d, torch.from_numpy(X).cuda(), nnet_in=True, onehot_dtype=torch.float32)
        ch, sv, hs = co.expand("lightsout7", X)
        assert np.array_equal(out["children"].cpu().numpy(), ch)
        assert np.array_equal(out["solved"].cpu().numpy().astype(bool), sv)
        assert np.array_equal(out["hash"].cpu().numpy().view(np.uint64), hs)
        assert np.array_equal(out["nnet_in"].cpu().numpy(), ch.reshape(-1, 49))
        if n <= 65:  # one-hot rows, f32 bit pattern (pytorch_models.py:49-52: index pos*6 + value)
            want = no.onehot(ch.reshape(-1, 49), 6)
            assert np.array_equal(out["onehot"].cpu().numpy().view(np.uint32), want.astype(np.float32).view(np.uint32))
        assert np.array_equal(L.is_solved(e, d, torch.from_numpy(X).cuda()).cpu().numpy().astype(bool),
                              co.is_solved("lightsout7", X))
    assert np.array_equal(L.expand_fused(e, d, torch.from_numpy(lo["states"][:8].copy()).cuda())["children"].cpu().numpy(),
                          lo["expand_children_8"])
    empty = L.expand_fused(e, d, torch.zeros((0, 49), dtype=torch.uint8, device="cuda"))
    assert empty["children"].shape == (0, 49, 49)


def test_environment_api_mirror(L, lo):
    from deepcubea_amd.environments.lights_out import LOState
    from deepcubea_amd.utils import env_utils
    env = env_utils.get_environment("lightsout7")
    states = [LOState(s.copy()) for s in lo["states"][:8]]
    for a in (0, 24, 48):
        ns, tc = env.next_state(states, a)
        assert tc == [1.0] * 8 and all(np.array_equal(x.tiles, lo["next_state_all_actions"][a][i]) for i, x in enumerate(ns))
        ps = env.prev_state(ns, a)
        assert all(p == s for p, s in zip(ps, states))
    exp, tcs = env.expand(states)
    assert len(exp) == 8 and len(exp[0]) == 49 and all(np.all(t == 1.0) and t.shape == (49,) for t in tcs)
    assert np.array_equal(np.stack([np.stack([c.tiles for c in row]) for row in exp]), lo["expand_children_8"])
    probe = [LOState(s.copy()) for s in lo["is_solved_probe_states"]]
    assert np.array_equal(env.is_solved(probe), lo["is_solved_probe"])
    assert np.array_equal(env.state_to_nnet_input(states)[0], lo["states"][:8])
    with pytest.raises(IndexError):
        env.next_state(states, 49)
    st, nb = env.generate_states(200, (0, 30), seed=4)
    assert len(st) == 200 and min(nb) >= 0 and max(nb) <= 30 and all(set(np.unique(s.tiles)) <= {0, 1} for s in st)
    assert all(not s.tiles.any() for s, k in zip(st, nb) if k == 0)


def test_engine_python_semantics_reproduces_reference_astar_traces(L, lo):
    from deepcubea_amd.search_methods.engine import BwasEngine
    from tests.test_engine_hip import run_traced
    for key in [str(k) for k in lo["astar_py_cases"]]:
        w, B, hid = lo[key + "_cfg"]
        pc, nn = lo[key + "_result"]
        eng = BwasEngine("lightsout7", float(w), int(B), max_nodes=max(1 << 16, int(nn) + 4 * int(B) * 49 + 64))
        res = run_traced(L, eng, lo[key + "_root"], int(hid))
        assert res["solved"] and res["moves"] == lo[key + "_moves"].tolist(), key
        assert res["path_cost"] == pc and res["nodes_generated"] == int(nn), key
        assert np.array_equal(res["trace"], lo[key + "_trace"]), key
        r2 = eng.solve_builtin(lo[key + "_root"], int(hid), chunk=9, use_graph=True)
        assert r2["moves"] == lo[key + "_moves"].tolist() and r2["nodes_generated"] == int(nn)
        eng.close()


@pytest.mark.parametrize("sem", ["py", "cpp"])
def test_engine_vs_oracle_at_train_sh_geometry(L, co, sem):
    """train.sh:68: lightsout7, weight 0.2, batch 1000 (49 000 children per iteration)."""
    from deepcubea_amd.search_methods.engine import BwasEngine
    from tests.test_engine_hip import run_traced
    s = np.zeros((1, 49), np.uint8)
    for a in (3, 17, 40, 22, 9, 31, 45, 12):
        s = co.next_state("lightsout7", s, a)
    root = s[0]
    osem, semv = (co.SEM_PY, L.SEM_PY) if sem == "py" else (co.SEM_CPP, L.SEM_CPP)
    ref = co.astar("lightsout7", root, 0.2, 1000, osem, heur_builtin_id=1, trace_cap=4096, max_iters=60)
    eng = BwasEngine("lightsout7", 0.2, 1000, max_nodes=ref["nodes_generated"] + 8 * 49000 + 64, semantics=semv)
    res = run_traced(L, eng, root, 1, max_iters=60)
    assert res["nodes_generated"] == ref["nodes_generated"] and res["iterations"] == ref["iterations"]
    assert np.array_equal(res["trace"][:, 2], ref["trace"][:, 2])
    if sem == "py":
        assert np.array_equal(res["trace"], ref["trace"])
    if ref["solved"]:
        assert res["solved"] and len(res["moves"]) == len(ref["moves"])
        t = root[None].copy()
        for a in res["moves"]:
            t = co.next_state("lightsout7", t, a)
        assert co.is_solved("lightsout7", t)[0]
    eng.close()


def test_scrambler_walks_are_reproducible_and_replayable(L, co):
    e, d, D, A, depth = L.env_ids("lightsout7")
    st, nb, mv = L.generate_states(e, d, 3000, 0, 50, 123, index0=10, want_moves=True)  # train.sh:65 back_max 50
    st, nb, mv = st.cpu().numpy(), nb.cpu().numpy(), mv.cpu().numpy()
    assert nb.min() == 0 and nb.max() == 50 and set(np.unique(st)) <= {0, 1} and (mv[mv >= 0] < 49).all()
    for i in range(0, 3000, 97):  # replaying the recorded presses from the goal reaches the state
        s = np.zeros((1, 49), np.uint8)
        for a in mv[i, :nb[i]]:
            s = co.next_state("lightsout7", s, int(a))
        assert np.array_equal(s[0], st[i])
    again, nb2, _ = L.generate_states(e, d, 1000, 0, 50, 123, index0=1010)
    assert np.array_equal(again.cpu().numpy(), st[1000:2000]) and np.array_equal(nb2.cpu().numpy(), nb[1000:2000])


def test_update_step_targets(L, co):
    """AVI update (GBFS, 1 step) on lightsout7 with a built-in heuristic: ctg = min over the 49 children of 1 + h, 0 at the
    goal — recomputed through the oracle's expansion."""
    from oracle import np_oracle as no
    e, d, D, A, depth = L.env_ids("lightsout7")
    st, _, _ = L.generate_states(e, d, 2000, 0, 50, 9)
    X = st.cpu().numpy()
    out = L.expand_fused(e, d, st, solved=False, hashes=False)
    h = L.heuristic_builtin(0, out["children"].view(-1, D))
    ctg, am = L.bellman_backup(h, L.is_solved(e, d, st), A)
    ch, _, _ = co.expand("lightsout7", X)
    hh = no.heur_builtin(0, ch.reshape(-1, 49)).reshape(-1, 49)
    want = (1.0 + hh.min(1)).astype(np.float32)
    want[~X.any(1)] = 0.0
    assert np.array_equal(ctg.cpu().numpy(), want) and np.array_equal(am.cpu().numpy(), np.argmin(hh, 1))
