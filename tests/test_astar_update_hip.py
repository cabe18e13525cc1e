"""ASTAR updates on the device (updaters/updater.py:36-54 of the reference; what `--update_method astar` of its lightsout7
training line runs, train.sh:65) against fixtures recorded from the reference's own `astar_update`
(tests/golden/make_golden_astar_update.py -> astar_update.npz: the start states, the per-instance weights numpy drew, the
triple it returned).  One batch-1 weighted A* per training state on the multi-instance engine, every popped node backed up
with 1 + min over its children of max(h, 0) (0 for a solved node): states in the reference's instance-major / pop order,
targets to float32 rounding, is_solved exactly."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(HERE, "golden", "astar_update.npz"))


def _hfn(L):
    # the recorded heuristic: DCA_HEUR_KNUTH3 of the NETWORK-INPUT rows, clipped at zero (>= 0 by construction)
    def h(x, is_onehot=False):
        assert not is_onehot
        return L.heuristic_builtin(L.HEUR_KNUTH3, x.contiguous())
    return h


@pytest.mark.parametrize("env_name,key", [("cube3", "cube3_steps1"), ("cube3", "cube3_steps4"), ("cube3", "cube3_steps12"),
                                          ("puzzle15", "puzzle15_steps6"), ("lightsout7", "lightsout7_steps5")])
@torch.no_grad()
def test_astar_update_matches_the_reference(fx, env_name, key):
    from deepcubea_amd import _lib as L
    from deepcubea_amd.updaters.updater import astar_update_dev
    from deepcubea_amd.utils import env_utils
    L.require_gpu()
    env = env_utils.get_environment(env_name)
    steps = int(key.split("steps")[1])
    roots = torch.from_numpy(fx[env_name + "_roots"]).cuda()
    w = fx[key + "_weights"]
    # one group = one `astar_update` call of the reference (all instances of the fixture: n <= 64)
    su, ctg, solved = astar_update_dev(roots, env, steps, _hfn(L), weights=w, instances_per_launch=64)
    assert su.shape == fx[key + "_states"].shape, (su.shape, fx[key + "_states"].shape)
    assert np.array_equal(su.cpu().numpy(), fx[key + "_states"])
    assert np.array_equal(solved.cpu().numpy(), fx[key + "_solved"])
    assert np.max(np.abs(ctg.double().cpu().numpy() - fx[key + "_ctg"])) < 1e-6
    # groups of 7 (does not divide the instance count: parked instances in the last group) = 7-state calls of their own
    s7, c7, v7 = astar_update_dev(roots, env, steps, _hfn(L), weights=w, instances_per_launch=7)
    parts = [astar_update_dev(roots[g:g + 7].contiguous(), env, steps, _hfn(L), weights=w[g:g + 7], instances_per_launch=64)
             for g in range(0, roots.shape[0], 7)]
    assert torch.equal(s7, torch.cat([p[0] for p in parts])) and torch.equal(c7, torch.cat([p[1] for p in parts]))
    assert torch.equal(v7, torch.cat([p[2] for p in parts]))


@torch.no_grad()
def test_updater_astar_method_feeds_the_training_triple():
    """`Updater(..., "ASTAR")` (updater.py:84-165): the device-resident triple and the host triple of the reference's
    `update()`: targets of solved pops are 0, every other target >= 1, one row per popped node."""
    from deepcubea_amd import _lib as L
    from deepcubea_amd.updaters.updater import Updater
    from deepcubea_amd.utils import env_utils
    env = env_utils.get_environment("lightsout7")
    upd = Updater(env, 300, 6, _hfn(L), 4, "ASTAR", update_batch_size=128, seed=5)
    sn, ctg, sv = upd.update_dev()
    assert sn.dtype == torch.uint8 and sn.shape[1] == 49 and ctg.shape == (sn.shape[0], 1) and sv.shape == (300,)
    assert 300 <= sn.shape[0] <= 4 * 300
    is_goal = (sn == 0).all(dim=1)  # lights_out.py:55-63: the goal is all lights off
    assert float(ctg[is_goal].abs().max()) == 0.0 if bool(is_goal.any()) else True
    assert float(ctg[~is_goal].min()) >= 1.0
    assert int(sv.sum()) >= 1  # back_max 6: some walks are short enough to be solved within 4 pops
    lst, out, solved = upd.update()
    assert isinstance(lst, list) and lst[0].shape == tuple(sn.shape) and out.shape == tuple(ctg.shape) and solved.dtype == bool
    with pytest.raises(ValueError):
        Updater(env, 10, 3, _hfn(L), 1, "BFS")
