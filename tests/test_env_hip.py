"""GPU parity tests of the HIP environment kernels (through the C ABI) against the CPU oracle and the
reference-generated golden fixtures.  Integer/byte work: every comparison is bit-exact."""
import hashlib

import numpy as np
import pytest
import torch

from tests.conftest import synth_states

pytestmark = pytest.mark.gpu

ENVS = ["cube3", "puzzle15", "puzzle24", "puzzle35", "puzzle48"]


@pytest.fixture(scope="module")
def L():
    from deepcubea_amd import _lib
    _lib.require_gpu()
    _lib.lib()
    return _lib


@pytest.fixture(scope="module")
def co():
    from oracle import c_oracle
    return c_oracle


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def u64(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("env", ENVS)
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 4099])
def test_expand_fused_vs_oracle(L, co, env, n):
    from oracle import np_oracle as no
    e, d, D, A, depth = L.env_ids(env)
    S = synth_states(n, D, seed=n)
    if n >= 64:  # plant goal neighbours so is_solved fires
        goal = np.arange(54, dtype=np.uint8) if env == "cube3" else np.concatenate((np.arange(1, D), [0])).astype(np.uint8)
        for a in range(A):
            S[3 + 5 * a] = co.next_state(env, goal[None], a)[0]
        S[1] = goal
    ch, sv, hs = co.expand(env, S)
    for oh_dt in (None, torch.float32, torch.float16, torch.bfloat16):
        out = L.expand_fused(e, d, dev(S), children=True, nnet_in=True, onehot_dtype=oh_dt, solved=True, hashes=True)
        torch.cuda.synchronize()
        assert np.array_equal(out["children"].cpu().numpy(), ch)
        assert np.array_equal(out["solved"].cpu().numpy().astype(bool), sv)
        assert sv.sum() >= (A if n >= 64 else 0)
        assert np.array_equal(u64(out["hash"]), hs)
        nn_in = co.nnet_input(env, ch.reshape(-1, D))
        assert np.array_equal(out["nnet_in"].cpu().numpy().reshape(-1, D), nn_in)
        if oh_dt is not None:
            want = no.onehot(nn_in, depth)
            got = out["onehot"].float().cpu().numpy()
            assert got.shape == want.shape and np.array_equal(got, want)
            if oh_dt == torch.float32:  # bit pattern, not just value
                assert np.array_equal(out["onehot"].cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("env", ENVS)
def test_expand_partial_outputs_and_empty(L, co, env):
    e, d, D, A, depth = L.env_ids(env)
    S = synth_states(130, D, seed=7)
    ch, sv, hs = co.expand(env, S)
    out = L.expand_fused(e, d, dev(S), children=False, nnet_in=False, onehot_dtype=None, solved=False, hashes=True)
    assert set(out.keys()) == {"hash"} and np.array_equal(u64(out["hash"]), hs)
    out = L.expand_fused(e, d, dev(S), children=True, solved=False, hashes=False)
    assert np.array_equal(out["children"].cpu().numpy(), ch)
    empty = L.expand_fused(e, d, torch.zeros((0, D), dtype=torch.uint8, device="cuda"))
    assert empty["children"].shape == (0, A, D)


@pytest.mark.parametrize("env", ENVS)
def test_next_prev_state(L, co, env):
    e, d, D, A, depth = L.env_ids(env)
    for n in (1, 65, 1000):
        S = synth_states(n, D, seed=3 * n)
        dS = dev(S)
        for a in range(A):
            nx = L.next_state(e, d, dS, a)
            assert np.array_equal(nx.cpu().numpy(), co.next_state(env, S, a))
            pv = L.next_state(e, d, dS, a, prev=True)
            assert np.array_equal(pv.cpu().numpy(), co.next_state(env, S, a ^ 1))
    with pytest.raises(L.DcaError):
        L.next_state(e, d, dS, A)  # action out of range -> DCA_E_BADARG


def test_golden_fixtures(L, golden):
    e, d, D, A, depth = L.env_ids("cube3")
    S = golden["cube3_synth64_in"]
    out = L.expand_fused(e, d, dev(S), nnet_in=True, onehot_dtype=torch.float32)
    assert np.array_equal(out["children"].cpu().numpy(), golden["cube3_synth64_children"])
    assert np.array_equal(out["nnet_in"].cpu().numpy(), golden["cube3_synth64_nnet_in"])
    assert np.array_equal(out["solved"].cpu().numpy().astype(bool), golden["cube3_synth64_is_solved"])
    for a in range(12):
        assert np.array_equal(L.next_state(e, d, dev(S), a).cpu().numpy(), golden["cube3_synth64_next_state"][a])
        assert np.array_equal(L.next_state(e, d, dev(S), a, prev=True).cpu().numpy(),
                              golden["cube3_synth64_prev_state"][a])
    S1000 = synth_states(1000, 54, 0)
    ch = L.expand_fused(e, d, dev(S1000))["children"].cpu().numpy()
    assert sha(ch) == str(golden["cube3_synth1000_children_sha256"])
    goal = np.arange(54, dtype=np.uint8)[None]
    g = L.expand_fused(e, d, dev(goal))
    assert np.array_equal(g["children"].cpu().numpy()[0], golden["cube3_goal_children"])
    assert L.is_solved(e, d, dev(goal)).cpu().numpy()[0] == 1
    for name, n in (("puzzle15", 4), ("puzzle24", 5), ("puzzle35", 6), ("puzzle48", 7)):
        e, d, D, A, depth = L.env_ids(name)
        P = golden[name + "_synth64_in"]
        out = L.expand_fused(e, d, dev(P))
        assert np.array_equal(out["children"].cpu().numpy(), golden[name + "_synth64_children"])
        assert np.array_equal(out["solved"].cpu().numpy().astype(bool), golden[name + "_synth64_is_solved"])
        for a in range(4):
            assert np.array_equal(L.next_state(e, d, dev(P), a, prev=True).cpu().numpy(),
                                  golden[name + "_synth64_prev"][:, a])
    P1000 = synth_states(1000, 16, 0)
    e, d, D, A, depth = L.env_ids("puzzle15")
    nxt = torch.stack([L.next_state(e, d, dev(P1000), a) for a in range(4)], 1).cpu().numpy()
    assert sha(nxt) == str(golden["puzzle15_synth1000_next4_sha256"])  # BASELINE configs[0]


def test_unaligned_pointers(L, co):
    e, d, D, A, depth = L.env_ids("cube3")
    S = synth_states(333, 54, seed=9)
    buf = torch.zeros(333 * 54 + 64, dtype=torch.uint8, device="cuda")
    for off in (1, 2, 7, 8):
        view = buf[off:off + 333 * 54].view(333, 54)
        view.copy_(dev(S))
        ch, sv, hs = co.expand("cube3", S)
        obuf = torch.zeros(333 * 648 + 64, dtype=torch.uint8, device="cuda")
        och = obuf[off:off + 333 * 648].view(333, 12, 54)
        ohb = torch.zeros(333 * 12 * 324 + 16, dtype=torch.float32, device="cuda")
        oh = ohb[1:1 + 333 * 12 * 324].view(333 * 12, 324)  # 4-byte aligned only
        out = L.expand_fused(e, d, view, out={"children": och, "onehot": oh})
        assert np.array_equal(out["children"].cpu().numpy(), ch)
        assert np.array_equal(u64(out["hash"]), hs)
        from oracle import np_oracle as no
        assert np.array_equal(oh.cpu().numpy(), no.onehot(co.nnet_input("cube3", ch.reshape(-1, 54)), 6))
        assert int(obuf[:off].sum()) == 0 and int(obuf[off + 333 * 648:].sum()) == 0  # no stray writes


def test_standalone_ops(L, co, golden):
    from oracle import np_oracle as no
    for env in ENVS:
        e, d, D, A, depth = L.env_ids(env)
        S = synth_states(777, D, seed=21)
        dS = dev(S)
        assert np.array_equal(u64(L.hash64(dS)), co.hash64(S))
        assert np.array_equal(L.is_solved(e, d, dS).cpu().numpy().astype(bool), co.is_solved(env, S))
        nn_in = L.nnet_input(e, d, dS)
        assert np.array_equal(nn_in.cpu().numpy(), co.nnet_input(env, S))
        for dt in (torch.float32, torch.float16, torch.bfloat16):
            oh = L.onehot(nn_in, depth, dt)
            assert np.array_equal(oh.float().cpu().numpy(), no.onehot(co.nnet_input(env, S), depth))
        for hid in range(4):
            assert np.array_equal(L.heuristic_builtin(hid, dS).cpu().numpy(), co.heur_builtin(hid, S)), (env, hid)
    S = golden["cube3_synth64_in"]
    assert np.array_equal(L.heuristic_builtin(0, dev(S)).cpu().numpy(), golden["heur_mod97_cube3_synth64"])
    assert np.array_equal(L.heuristic_builtin(1, dev(S)).cpu().numpy(), golden["heur_knuth3_cube3_synth64"])


def test_config2_full_size_properties(L, co):
    """BASELINE configs[1]: 1M synthetic cube3 states, all 12 moves, fused one-hot.  Full bit-compare of
    children/hash/solved against the C oracle; one-hot checked on device by size-independent properties."""
    e, d, D, A, depth = L.env_ids("cube3")
    n = 1_000_000
    S = synth_states(n, 54, 0)
    dS = dev(S)
    out = L.expand_fused(e, d, dS, nnet_in=True, onehot_dtype=torch.float32)
    torch.cuda.synchronize()
    ch, sv, hs = co.expand("cube3", S)
    assert np.array_equal(out["children"].cpu().numpy(), ch)
    assert np.array_equal(u64(out["hash"]), hs)
    assert np.array_equal(out["solved"].cpu().numpy().astype(bool), sv)
    assert np.array_equal(out["nnet_in"].cpu().numpy(), ch.reshape(-1, 54) // 9)
    oh = out["onehot"]
    assert oh.shape == (n * 12, 324)
    step = 1_000_000
    for s in range(0, n * 12, step):
        blk = oh[s:s + step].view(-1, 54, 6)
        assert bool((blk.sum(-1) == 1).all())  # exactly one hot per sticker
        assert bool((blk.argmax(-1).to(torch.uint8) == out["nnet_in"][s:s + step]).all())
        assert bool(((blk == 0) | (blk == 1)).all())
    # move o inverse = identity, on device, for every move
    for a in range(12):
        back = L.next_state(e, d, out["children"][:, a].contiguous(), a ^ 1)
        assert bool((back == dS).all())


def test_environment_api_mirror(L, golden):
    from deepcubea_amd.utils import env_utils, search_utils
    env = env_utils.get_environment("cube3")
    S = golden["cube3_synth64_in"]
    states = env.np_to_states(S)
    nxt, tc = env.next_state(states, 5)
    assert tc == [1.0] * 64 and np.array_equal(np.stack([s.colors for s in nxt]), golden["cube3_synth64_next_state"][5])
    prv = env.prev_state(states, 5)
    assert np.array_equal(np.stack([s.colors for s in prv]), golden["cube3_synth64_prev_state"][5])
    exp, tcs = env.expand(states)
    assert np.array_equal(np.stack([np.stack([c.colors for c in row]) for row in exp]), golden["cube3_synth64_children"])
    assert all(np.all(t == 1.0) and t.shape == (12,) for t in tcs)
    flat = [c for row in exp for c in row]
    assert np.array_equal(env.is_solved(flat), golden["cube3_synth64_is_solved"])
    assert np.array_equal(env.state_to_nnet_input(flat)[0], golden["cube3_synth64_nnet_in"])
    assert env.is_solved(env.generate_goal_states(3)).all()
    assert len({hash(s) for s in states}) == 64 and states[0] == env.np_to_states(S[:1])[0]
    # shipped optimal solutions replay to the goal through the mirror (astar.py:443 check)
    for i in range(5):
        st = env.np_to_states(golden["cube3_test_states"][i:i + 1])[0]
        mv = [int(m) for m in golden["cube3_test_opt_moves"][i] if m >= 0]
        assert search_utils.is_valid_soln(st, mv, env)
    penv = env_utils.get_environment("puzzle15")
    P = golden["puzzle15_synth64_in"]
    pst = penv.np_to_states(P)
    pexp, _ = penv.expand(pst)
    assert np.array_equal(np.stack([np.stack([c.tiles for c in row]) for row in pexp]), golden["puzzle15_synth64_children"])
    st, ks = env.generate_states(200, (0, 8))
    assert len(st) == 200 and max(ks) <= 8 and all(k >= 0 for k in ks)
    zero = [s for s, k in zip(st, ks) if k == 0]
    assert all(env.is_solved(zero)) if zero else True
    with pytest.raises(ValueError):
        env_utils.get_environment("sokoban")
