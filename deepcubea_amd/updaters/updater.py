"""Approximate-value-iteration UPDATE STEP on the device (BASELINE configs[4]).

Mirror of the data-generation half of the reference's training loop — `ctg_approx/avi.py:129-159` (do_update),
`updaters/updater.py:11-33,57-165` (gbfs_update, update_runner, Updater), `search_methods/gbfs.py:43-120`,
`utils/search_utils.py:16-32` (bellman) — with the 30 CPU worker processes and the heuristic queues replaced by
one replica per GPU: states are generated, expanded, evaluated and backed up without leaving HBM.

    updater = Updater(env, num_states, back_max, heuristic_fn_dev, num_steps, "GBFS", eps_max=0.0)
    states_nnet, ctg, is_solved = updater.update()      # same triple as updater.py:116-123

The training step that consumes these targets is `utils/nnet_utils.train_nnet`; `ctg_approx/avi.py` ties both into
the reference's loop.  ASTAR updates (updater.py:36-54; `--update_method astar`, what the reference's lightsout7 training
line uses, train.sh:65) run on the multi-instance BWAS engine: `astar_update_dev`.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib
from ..search_methods import sharding


def bellman_dev(env, states: torch.Tensor, heuristic_fn_dev: Callable, onehot_dtype=None, clip_zero: bool = True):
    """search_utils.bellman (search_utils.py:16-32) on device.
    -> (ctg_backup f32 [n], argmin i32 [n], children u8 [n,A,D]).
    clip_zero: max(h, 0) before the backup — what the update's heuristic servers do (avi.py:213 clip_zero=True);
    gbfs_test runs on the unclipped current network (avi.py:251 / nnet_utils.get_heuristic_fn default)."""
    A = env.get_num_moves()
    out = env.expand_dev(states, children=True, nnet_in=(onehot_dtype is None), onehot_dtype=onehot_dtype, solved=False,
                         hashes=False)
    h = heuristic_fn_dev(out["onehot"], True) if onehot_dtype is not None else heuristic_fn_dev(out["nnet_in"])
    ctg, am = _lib.bellman_backup(h, env.is_solved_dev(states), A, clip_zero=clip_zero)
    return ctg, am, out["children"]


def gbfs_update_dev(states: torch.Tensor, env, num_steps: int, heuristic_fn_dev: Callable, eps_max: float = 0.0,
                    generator: Optional[torch.Generator] = None, onehot_dtype=None,
                    rand_child: Optional[Callable[[int, int], torch.Tensor]] = None, return_steps: bool = False,
                    clip_zero: bool = True):
    """updater.py:11-33 gbfs_update with GBFS.step (gbfs.py:43-120) for all instances at once.
    -> (states_update u8 [T,D], cost_to_go f32 [T], is_solved bool [n]) in the reference's instance-major order.
    `rand_child(k, A)` may override the random-child draw (tests stub it, like np.random.choice)."""
    n, D = states.shape
    A = env.get_num_moves()
    dev = states.device
    cur = states.clone()
    solved = torch.zeros(n, dtype=torch.bool, device=dev)
    steps = torch.zeros(n, dtype=torch.int32, device=dev)  # Instance.num_steps (gbfs.py:26-28)
    eps = torch.rand(n, device=dev, generator=generator) * eps_max  # updater.py:12
    traj_states: List[torch.Tensor] = []
    traj_ctg: List[torch.Tensor] = []
    traj_inst: List[torch.Tensor] = []
    hist: List[Tuple[torch.Tensor, torch.Tensor]] = []  # (instance idx, state rows) per recorded step, for seen-checks
    arange = torch.arange(n, device=dev)
    for _ in range(num_steps):
        uns = arange[~solved]
        if uns.numel():  # _record_solved (gbfs.py:67-84)
            sv = env.is_solved_dev(cur[uns]).bool()
            idx = uns[sv]
            if idx.numel():
                traj_states.append(cur[idx].clone())
                traj_ctg.append(torch.zeros(idx.numel(), dtype=torch.float32, device=dev))
                traj_inst.append(idx)
                solved[idx] = True
        uns = arange[~solved]
        if uns.numel() == 0:
            continue
        st = cur[uns].contiguous()  # _move (gbfs.py:86-120)
        ctg, am, children = bellman_dev(env, st, heuristic_fn_dev, onehot_dtype, clip_zero)
        traj_states.append(st)
        traj_ctg.append(ctg)
        traj_inst.append(uns)
        hist.append((uns, st))
        k = uns.numel()
        rows = torch.arange(k, device=dev)
        nxt = children[rows, am.long()]
        # seen_states (gbfs.py:110-111): every state this instance already put on its trajectory
        seen = torch.zeros(k, dtype=torch.bool, device=dev)
        pos = torch.full((n,), -1, dtype=torch.long, device=dev)
        pos[uns] = rows
        for h_idx, h_st in hist:
            p = pos[h_idx]
            ok = p >= 0
            if ok.any():
                eq = (h_st[ok] == nxt[p[ok]]).all(dim=1)
                seen[p[ok]] |= eq
        rnd = torch.rand(k, device=dev, generator=generator) < eps[uns]
        pick_rand = rnd | seen
        if pick_rand.any():
            kk = int(pick_rand.sum())
            ridx = rand_child(kk, A) if rand_child is not None else torch.randint(0, A, (kk,), device=dev,
                                                                                   generator=generator)
            nxt[pick_rand] = children[rows[pick_rand], ridx.to(dev).long()]
        cur[uns] = nxt
        steps[uns] += 1
    if return_steps:
        return solved, steps
    if not traj_states:
        return (torch.zeros((0, D), dtype=torch.uint8, device=dev), torch.zeros(0, dtype=torch.float32, device=dev),
                solved)
    su = torch.cat(traj_states)
    cg = torch.cat(traj_ctg)
    inst = torch.cat(traj_inst)
    order = torch.sort(inst, stable=True).indices  # instance-major, steps in order (misc_utils.flatten(trajs))
    return su[order], cg[order], solved


def astar_update_dev(states: torch.Tensor, env, num_steps: int, heuristic_fn_dev: Callable,
                     weights: Optional[np.ndarray] = None, generator: Optional[np.random.Generator] = None,
                     instances_per_launch: int = 64, engine=None, onehot_dtype=None, engines_in_rotation: Optional[int] = None):
    """updater.py:36-54 astar_update on the device: one batch-1 weighted A* per training state (`AStar(states, env,
    heuristic_fn, weights)` with weights ~ U[0, 1), one per instance), `num_steps` steps (`astar.step(heuristic_fn, 1)`:
    every unsolved instance pops its cheapest node, expands it, the network scores ALL children, the CLOSED check drops
    duplicates, the rest is pushed), then `Node.compute_bellman` on every popped node: 0 for a solved node, else
    min over its children of (transition cost 1 + the child's heuristic, clipped at 0 like the update's heuristic servers,
    avi.py:213).  The searches are the engine's (libdca_hip.so, PY semantics: float64 cost = w * g + h, (cost, push count)
    order, sequential CLOSED rule), up to `instances_per_launch` of them sharing every launch (grid.y = instance).
    Each group of `instances_per_launch` states is one `astar_update` call of the reference (its `update_runner` cuts the
    states into `update_batch_size` pieces the same way, updater.py:62-71), including how that call pairs weights with
    instances: `AStar.step` zips `self.weights` with the instances that have not found a goal yet (astar.py:262-263,
    279-281), so once an instance has finished, the j-th REMAINING instance searches with weights[j] — reproduced here step
    by step (the fixtures recorded from the reference depend on it), ON THE DEVICE: rank among the live instances by a
    prefix sum of the done flags, `dca_engine_set_weights_dev`.
    Nothing in the stepping loop talks to the host (one device synchronisation per step and instance made a
    500 000-state update launch- and sync-bound): groups restart from device rows (`reset_many`), `engines_in_rotation`
    engines are stepped side by side and their children scored by ONE network call per step (R x K x moves rows instead of
    K x moves: 64 x 12 rows do not fill a GEMM), popped nodes are collected unfiltered and compacted once at the end.
    -> (states_update u8 [T, D], cost_to_go f32 [T], is_solved bool [n]) in the reference's order: instance-major, each
    instance's popped nodes in pop order (misc_utils.flatten(astar.get_popped_nodes())).
    onehot_dtype: the closure wants one-hot rows (a network whose first layer has no uint8 kernel, e.g. lightsout7's): the
    expansion launch writes them (pytorch_models.py:49-52) and they are handed over with is_onehot=True."""
    from ..search_methods.engine import BwasEngine
    n, D = states.shape
    A = env.get_num_moves()
    dev = states.device
    if weights is None:
        weights = (generator or np.random.default_rng()).random(n)  # updater.py:37 np.random.rand(len(states))
    weights = np.asarray(weights, np.float64)
    assert weights.shape == (n,)
    K = max(1, min(int(instances_per_launch), 64, n))
    if engine is not None:
        engines = [engine]
    else:
        # enough engines side by side that one network call sees ~65 000 rows (never more than the states need)
        R = engines_in_rotation if engines_in_rotation else max(1, min(16, (1 << 16) // max(K * A, 1), -(-n // K)))
        # ids: the root, then <= A + 15 per step (a batch's ids start on a multiple of 16)
        engines = [BwasEngine(env.env_name, 0.0, 1, max_nodes=64 + num_steps * (A + 16) + A + 16, num_instances=K,
                              onehot_dtype=onehot_dtype) for _ in range(R)]
    assert all(e.batch_size == 1 and e.num_instances >= K for e in engines)
    Kc = engines[0].num_instances
    R = len(engines)
    w_all = torch.from_numpy(weights).to(dev)
    states = states.contiguous()
    out_states: List[torch.Tensor] = []
    out_ctg: List[torch.Tensor] = []
    out_inst: List[torch.Tensor] = []
    out_live: List[torch.Tensor] = []
    found = torch.zeros(n, dtype=torch.bool, device=dev)
    zeros_tail = torch.zeros(Kc, dtype=torch.float64, device=dev)
    for s0 in range(0, n, R * Kc):
        # the engines of this round and their slices of the states: (engine, first state, instances in use)
        groups = [(engines[r], s0 + r * Kc, min(Kc, n - (s0 + r * Kc))) for r in range(R) if s0 + r * Kc < n]
        tot = sum(k for _, _, k in groups)
        for eng, lo, k in groups:
            eng.reset_many(states[lo:lo + k])
        # root nodes: heuristic of all roots in one call (astar.py:246-249 add_heuristic_and_cost(root_nodes, ...))
        root_nn = _lib.nnet_input(env._env_id, env._dim, states[s0:s0 + tot])
        h_root = heuristic_fn_dev(root_nn).to(torch.float32).view(-1)
        for eng, lo, k in groups:
            eng.root_commit_many(h_root[lo - s0:lo - s0 + k])
        done = torch.zeros(tot, dtype=torch.bool, device=dev)  # instances that have popped a goal: no longer stepped (astar.py:262-263)
        for step in range(num_steps):
            nns, ohs = [], []
            for eng, lo, k in groups:
                # zip(self.weights, remaining instances): positional (astar.py:279-281) — the j-th live instance gets weights[j]
                alive = ~done[lo - s0:lo - s0 + k]
                rank = torch.cumsum(alive.to(torch.int64), 0) - 1
                w_step = torch.where(alive, w_all[lo:lo + k][rank.clamp_min(0)], zeros_tail[:k])
                eng.set_weights_dev(torch.cat([w_step, zeros_tail[:Kc - k]]) if k < Kc else w_step)
                nn, oh = eng.pop_expand()
                nns.append(nn)
                ohs.append(oh)
            if ohs[0] is not None:
                h_all = heuristic_fn_dev(ohs[0] if len(ohs) == 1 else torch.cat(ohs), True)
            else:
                h_all = heuristic_fn_dev(nns[0] if len(nns) == 1 else torch.cat(nns))
            h_all = h_all.to(torch.float32).contiguous().view(len(groups), Kc * A)
            for gi, (eng, lo, k) in enumerate(groups):
                h = h_all[gi].contiguous()
                popped, flags = eng.last_popped()  # [Kc, D], [Kc]
                fk = flags[:k]
                hk = torch.clamp_min(h.view(Kc, A)[:k], 0.0)
                backup = 1.0 + hk.min(dim=1).values           # tc + node_c.heuristic, min over the children (astar.py:43-44)
                backup = torch.where(fk == 2, torch.zeros_like(backup), backup)  # a solved node backs up to 0
                out_states.append(popped[:k])
                out_ctg.append(backup)
                out_inst.append(torch.arange(lo, lo + k, device=dev))
                out_live.append(fk != 0)
                solved_now = fk == 2
                found[lo:lo + k] |= solved_now
                done[lo - s0:lo - s0 + k] |= solved_now
                eng.commit(h)
            if step % 8 == 7 and step + 1 < num_steps and bool(done.all()):  # (one look at the host every 8 steps)
                break
    if engine is None:
        for e in engines:
            e.close()
    if not out_states:
        return (torch.zeros((0, D), dtype=torch.uint8, device=dev), torch.zeros(0, dtype=torch.float32, device=dev), found)
    live = torch.cat(out_live)
    su, cg, inst = torch.cat(out_states)[live], torch.cat(out_ctg)[live], torch.cat(out_inst)[live]
    order = torch.sort(inst, stable=True).indices  # instance-major, pops in step order
    return su[order], cg[order], found


class Updater:
    """Drop-in for updaters/updater.py:84-165: same constructor meaning and `update()` triple, but
    `heur_fn_i_q / heur_fn_o_qs` become a device heuristic closure and the worker processes become ranks:
    every rank generates and backs up its share (split_evenly) of `num_states` on its own GPU."""

    def __init__(self, env, num_states: int, back_max: int, heuristic_fn_dev: Callable, num_steps: int,
                 update_method: str = "GBFS", update_batch_size: int = 1_000_000, eps_max: float = 0.0, seed: int = 0,
                 onehot_dtype=None):
        if update_method.upper() not in ("GBFS", "ASTAR"):
            raise ValueError("Unknown update method %s" % update_method)  # updater.py:73
        self.method = update_method.upper()
        self.env, self.num_states, self.back_max = env, int(num_states), int(back_max)
        self.hfn, self.num_steps, self.eps_max = heuristic_fn_dev, int(num_steps), float(eps_max)
        self.batch, self.seed, self.onehot_dtype = int(update_batch_size), int(seed), onehot_dtype
        self.world, self.rank = sharding.world_info()
        per = [self.num_states // self.world + (1 if r < self.num_states % self.world else 0) for r in range(self.world)]
        self.local_n = per[self.rank]  # misc_utils.split_evenly (misc_utils.py:29-36)
        self.index0 = sum(per[:self.rank])

    def update_dev(self):
        """This rank's shard, on device: (states_nnet u8 [T,D], ctg f32 [T,1], is_solved bool [local_n])."""
        env = self.env
        gen = torch.Generator(device="cuda")
        gen.manual_seed(self.seed * 1000003 + self.rank)
        sn, cg, sv = [], [], []
        start = 0
        while start < self.local_n:
            m = min(self.batch, self.local_n - start)
            states, _, _ = _lib.generate_states(env._env_id, env._dim, m, 0, self.back_max, self.seed,
                                                self.index0 + start)  # updater.py:66 env.generate_states(n,(0,back_max))
            if self.method == "ASTAR":
                wts = np.random.default_rng([self.seed, self.rank, start]).random(m)  # updater.py:37 np.random.rand
                su, ctg, solved = astar_update_dev(states, env, self.num_steps, self.hfn, wts, onehot_dtype=self.onehot_dtype)
            else:
                su, ctg, solved = gbfs_update_dev(states, env, self.num_steps, self.hfn, self.eps_max, gen,
                                                  self.onehot_dtype)
            sn.append(_lib.nnet_input(env._env_id, env._dim, su))  # updater.py:75 state_to_nnet_input
            cg.append(ctg)
            sv.append(solved)
            start += m
        return torch.cat(sn), torch.cat(cg)[:, None], torch.cat(sv)

    def update(self):
        """updater.py:116-123: (states_update_nnet: List[np.ndarray], output_update [T,1], is_solved) for THIS rank's
        shard (a trainer on the same rank consumes it; no collective is needed for the update itself)."""
        sn, out, sv = self.update_dev()
        return [sn.cpu().numpy()], out.cpu().numpy(), sv.cpu().numpy()


def gbfs_test_dev(num_states: int, back_max: int, env, heuristic_fn_dev: Callable, max_solve_steps: Optional[int] = None,
                  seed: int = 0) -> List[Tuple[int, float, float, float]]:
    """search_methods/gbfs.py:126-183 gbfs_test on the device: `num_states` states spread over 30 scramble depths
    0..back_max, greedy best-first search for `max_solve_steps` steps, one line of statistics per depth in the
    reference's format.  Returns [(back_step, %solved, avgSolveSteps, ctg mean)] for callers / tests."""
    back_steps = list(np.linspace(0, back_max, 30, dtype=int))
    per = [num_states // len(back_steps) + (1 if i < num_states % len(back_steps) else 0) for i in range(len(back_steps))]
    chunks, depth = [], []
    index0 = 0
    for bs, n_i in zip(back_steps, per):
        if n_i > 0:
            st, _, _ = _lib.generate_states(env._env_id, env._dim, n_i, int(bs), int(bs), seed, index0)
            chunks.append(st)
            depth.append(torch.full((n_i,), int(bs), dtype=torch.int64))
            index0 += n_i
    states = torch.cat(chunks)
    state_back_steps = torch.cat(depth).numpy()
    if max_solve_steps is None:
        max_solve_steps = max(int(state_back_steps.max()), 1)
    print("Solving %i states with GBFS with %i steps" % (states.shape[0], max_solve_steps))
    # the reference's gbfs_test greedy-walks on the UNCLIPPED current network (gbfs.py:126-183, avi.py:251)
    solved_d, steps_d = gbfs_update_dev(states, env, max_solve_steps, heuristic_fn_dev, 0.0, return_steps=True,
                                        clip_zero=False)
    is_solved_all = solved_d.cpu().numpy()
    num_steps_all = steps_d.cpu().numpy()
    state_ctg_all = heuristic_fn_dev(_lib.nnet_input(env._env_id, env._dim, states)).float().cpu().numpy()
    rows = []
    for back_step_test in np.unique(state_back_steps):
        idx = np.where(state_back_steps == back_step_test)[0]
        is_solved, num_steps, ctg = is_solved_all[idx], num_steps_all[idx], state_ctg_all[idx]
        per_solved = 100 * float(is_solved.sum()) / float(len(is_solved))
        avg_solve_steps = float(np.mean(num_steps[is_solved])) if per_solved > 0.0 else 0.0
        print("Back Steps: %i, %%Solved: %.2f, avgSolveSteps: %.2f, CTG Mean(Std/Min/Max): %.2f("
              "%.2f/%.2f/%.2f)" % (back_step_test, per_solved, avg_solve_steps, float(np.mean(ctg)), float(np.std(ctg)),
                                   np.min(ctg), np.max(ctg)))
        rows.append((int(back_step_test), per_solved, avg_solve_steps, float(np.mean(ctg))))
    return rows
