"""GPU parity tests of the AVI update step against the reference-generated fixtures and the
NumPy oracle; the state generator is checked by size-independent properties."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from deepcubea_amd import _lib
    _lib.require_gpu()
    return _lib


def _hfn(L, hid):
    def f(x_nnet_or_states, is_onehot=False):
        raise AssertionError("built-in heuristics work on raw states")
    return f


def _upd(L, env_name, roots, steps, hid, onehot_dtype=None):
    """gbfs_update_dev with a built-in heuristic evaluated on the CHILD STATES (the built-ins are defined on raw
    sticker/tile bytes, not on the network input), random-child draws stubbed to child 0 like the fixtures."""
    from deepcubea_amd.updaters import updater as up
    from deepcubea_amd.utils import env_utils
    env = env_utils.get_environment(env_name)
    A = env.get_num_moves()
    orig = up.bellman_dev

    def bellman_builtin(env_, states, hfn, oh=None, clip_zero=True):
        out = env_.expand_dev(states, children=True, solved=False, hashes=False)
        h = L.heuristic_builtin(hid, out["children"].view(-1, states.shape[1]))
        ctg, am = L.bellman_backup(h, env_.is_solved_dev(states), A, clip_zero=True)
        return ctg, am, out["children"]
    up.bellman_dev = bellman_builtin
    try:
        su, ctg, sv = up.gbfs_update_dev(torch.from_numpy(roots).cuda(), env, steps, None, 0.0,
                                         rand_child=lambda k, a: torch.zeros(k, dtype=torch.long, device="cuda"))
    finally:
        up.bellman_dev = orig
    return su.cpu().numpy(), ctg.cpu().numpy(), sv.cpu().numpy()


def test_update_step_vs_reference_fixtures(L, golden):
    for steps in (1, 3):
        su, ctg, sv = _upd(L, "cube3", golden["avi_cube3_roots"], steps, 1)
        assert np.array_equal(su, golden["avi_cube3_steps%d_states" % steps])
        # targets are float32 on the device (the trainer casts to float32 anyway, nnet_utils.py:32,78)
        assert np.array_equal(ctg, golden["avi_cube3_steps%d_ctg" % steps].astype(np.float32))
        assert np.array_equal(sv, golden["avi_cube3_steps%d_solved" % steps])
    su, ctg, sv = _upd(L, "puzzle15", golden["avi_puzzle15_roots"], 2, 1)
    assert np.array_equal(su, golden["avi_puzzle15_steps2_states"])
    assert np.array_equal(ctg, golden["avi_puzzle15_steps2_ctg"].astype(np.float32))
    assert np.array_equal(sv, golden["avi_puzzle15_steps2_solved"])


def test_bellman_backup_vs_oracle(L, golden):
    from oracle import np_oracle as no
    from tests.conftest import synth_states
    for env, n in (("cube3", 5000), ("puzzle24", 3000), ("puzzle48", 1000)):
        e, d, D, A, depth = L.env_ids(env)
        S = synth_states(n, D, 17)
        S[0] = np.arange(54, dtype=np.uint8) if env == "cube3" else np.concatenate((np.arange(1, D), [0]))
        bk, nxt, ch = no.bellman(env, S, lambda s: no.heur_builtin(0, s))
        out = L.expand_fused(e, d, torch.from_numpy(S).cuda(), solved=False, hashes=False)
        h = L.heuristic_builtin(0, out["children"].view(-1, D))
        ctg, am = L.bellman_backup(h, L.is_solved(e, d, torch.from_numpy(S).cuda()), A)
        assert np.array_equal(ctg.cpu().numpy(), bk.astype(np.float32)) and ctg[0] == 0
        assert np.array_equal(am.cpu().numpy(), np.argmin(nxt, 1))
    assert np.array_equal(
        no.bellman("cube3", golden["cube3_synth64_in"], lambda s: no.heur_builtin(0, s))[0],
        golden["avi_cube3_bellman_synth64_mod97"])


@pytest.mark.parametrize("env", ["cube3", "puzzle15", "puzzle48"])
def test_generate_states_properties(L, env):
    from oracle import c_oracle as co
    e, d, D, A, depth = L.env_ids(env)
    n, lo, hi = 200_000, 0, 30
    st, nb, mv = L.generate_states(e, d, n, lo, hi, seed=5, want_moves=True)
    st2, nb2, _ = L.generate_states(e, d, n, lo, hi, seed=5)
    assert torch.equal(st, st2) and torch.equal(nb, nb2)                      # reproducible
    st3, _, _ = L.generate_states(e, d, n // 2, lo, hi, seed=5, index0=n // 2)
    assert torch.equal(st3, st[n // 2:])                                      # shard-consistent (index0)
    assert not torch.equal(L.generate_states(e, d, n, lo, hi, seed=6)[0], st)
    nbn = nb.cpu().numpy()
    assert nbn.min() == lo and nbn.max() == hi
    cnt = np.bincount(nbn, minlength=hi + 1)
    assert np.all(np.abs(cnt - n / (hi + 1)) < 6 * np.sqrt(n / (hi + 1)))    # k ~ U{lo..hi}
    mvn = mv.cpu().numpy()
    used = mvn[mvn >= 0]
    mc = np.bincount(used, minlength=A)
    assert np.all(np.abs(mc - used.size / A) < 6 * np.sqrt(used.size / A))    # uniform moves
    assert ((mvn >= 0).sum(1) == nbn).all()
    # the walk itself, re-done with the CPU oracle from the recorded moves (a reverse move a = next_state(a^1);
    # ineligible puzzle moves are no-ops that still count, exactly like the reference's generate_states)
    m = 4000
    goal = np.arange(54, dtype=np.uint8) if env == "cube3" else np.concatenate((np.arange(1, D), [0])).astype(np.uint8)
    cur = np.tile(goal, (m, 1))
    for t in range(hi):
        for a in range(A):
            sel = mvn[:m, t] == a
            if sel.any():
                cur[sel] = co.next_state(env, cur[sel], a ^ 1)
    assert np.array_equal(cur, st[:m].cpu().numpy())
    if env == "cube3":  # cube moves are never no-ops: undoing the recorded moves in reverse order solves the state
        back = st.clone()
        for t in range(hi - 1, -1, -1):
            for a in range(A):
                idx = torch.nonzero(mv[:, t] == a).flatten()
                if idx.numel():
                    back[idx] = L.next_state(e, d, back[idx], a)
        assert bool(L.is_solved(e, d, back).all())
    assert bool(L.is_solved(e, d, st[nb == 0]).all())
    # ragged / empty
    assert L.generate_states(e, d, 0, 0, 5, 1)[0].shape == (0, D)
    assert L.generate_states(e, d, 3, 7, 7, 1)[1].tolist() == [7, 7, 7]


def test_updater_end_to_end_with_network(L):
    """Updater.update() with the ResNet heuristic: shapes / dtypes / target semantics of updater.py:116-123."""
    from deepcubea_amd.updaters.updater import Updater, bellman_dev
    from deepcubea_amd.utils import env_utils, nnet_utils
    from deepcubea_amd.utils.pytorch_models import ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    env = env_utils.get_environment("cube3")
    m = ResnetModel(54, 6, 256, 128, 2, 1, True)
    load_synthetic_weights(m, 5)
    m = m.cuda().eval()
    hfn = nnet_utils.get_heuristic_fn_dev(m, clip_zero=False, batch_size=50_000)
    upd = Updater(env, 30_000, 30, hfn, num_steps=1, update_batch_size=8192, seed=3)
    states_nnet, out, solved = upd.update()
    assert states_nnet[0].shape == (30_000, 54) and states_nnet[0].dtype == np.uint8 and states_nnet[0].max() <= 5
    assert out.shape == (30_000, 1) and out.dtype == np.float32 and solved.shape == (30_000,)
    assert (out[solved] == 0).all() and (out[~solved] >= 1.0).all()   # 1 + max(h,0), solved states -> 0
    assert 0 < solved.mean() < 0.2                                   # k=0 walks (1/31 of them) are solved
    # one-hot fused path gives the same targets as the index path
    st = L.generate_states(env._env_id, env._dim, 4096, 0, 30, 9)[0]
    a = bellman_dev(env, st, hfn)[0]
    b = bellman_dev(env, st, hfn, onehot_dtype=torch.float32)[0]
    assert torch.equal(a, b)
    assert Updater(env, 10, 5, hfn, 1, update_method="ASTAR").method == "ASTAR"  # (tests/test_astar_update_hip.py runs it)
    with pytest.raises(ValueError):
        Updater(env, 10, 5, hfn, 1, update_method="BFS")  # updater.py:72-73
