"""Cube4 environment kernels (the DCA_ENV_CUBE4 instantiation of the tile code, csrc/dca_env.hip; cpp/environments.cpp:263-370
of the reference) against the oracle on the GPU: every move and its inverse, the fused expansion (children, is_solved,
hash) on ragged sizes, is_solved on same-colour arrangements."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _states(n, seed):
    rng = np.random.default_rng(seed)
    return rng.permuted(np.tile(np.arange(96, dtype=np.uint8), (n, 1)), axis=1)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4099])
def test_cube4_moves_and_expansion_match_the_oracle(n):
    from deepcubea_amd import _lib as L
    from oracle import c_oracle as co
    L.require_gpu()
    s = _states(n, n)
    s[0] = np.arange(96)  # the goal among them: its children are one move from solved
    d = torch.from_numpy(s).cuda()
    for a in range(24):
        nxt = L.next_state(L.ENV_CUBE4, 0, d, a)
        assert np.array_equal(nxt.cpu().numpy(), co.next_state("cube4", s, a)), a
        assert torch.equal(L.next_state(L.ENV_CUBE4, 0, nxt, a, prev=True), d)
    out = L.expand_fused(L.ENV_CUBE4, 0, d)
    ch, sv, hs = co.expand("cube4", s)
    assert np.array_equal(out["children"].cpu().numpy(), ch)
    assert np.array_equal(out["solved"].cpu().numpy().astype(bool), sv)
    assert np.array_equal(out["hash"].cpu().numpy().view(np.uint64), hs)
    assert np.array_equal(L.hash64(d).cpu().numpy().view(np.uint64), co.hash64(s))


def test_cube4_is_solved_counts_any_same_colour_arrangement():
    from deepcubea_amd import _lib as L
    from oracle import c_oracle as co
    goal = np.arange(96, dtype=np.uint8)
    rng = np.random.default_rng(3)
    same = goal.copy()
    for f in range(6):
        same[f * 16:(f + 1) * 16] = f * 16 + rng.permutation(16)
    swapped = np.concatenate([goal[32:48], goal[16:32], goal[0:16], goal[48:]])
    off = goal.copy()
    off[[0, 95]] = off[[95, 0]]
    cases = np.stack([goal, same, swapped, off, _states(1, 9)[0]])
    got = L.is_solved(L.ENV_CUBE4, 0, torch.from_numpy(cases).cuda()).cpu().numpy().astype(bool)
    assert np.array_equal(got, co.is_solved("cube4", cases)) and got.tolist() == [True, True, True, False, False]
    # through the fused expansion: the children of a one-move scramble contain exactly the undoing move as solved
    one = co.next_state("cube4", goal[None], 5)
    out = L.expand_fused(L.ENV_CUBE4, 0, torch.from_numpy(one).cuda(), children=False, hashes=False)
    sv = out["solved"].cpu().numpy().astype(bool)
    assert sv.sum() == 1 and sv[5 ^ 1]
