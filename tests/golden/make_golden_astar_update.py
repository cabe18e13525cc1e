#!/usr/bin/env python3
"""Record tests/golden/astar_update.npz by IMPORTING the reference (build container only; /root/reference never travels):

    python tests/golden/make_golden_astar_update.py

Every array is produced by the reference's own `updaters/updater.py:astar_update` (AStar with one search instance per
training state, batch size 1, a random weight per instance, `num_steps` steps; then `Node.compute_bellman` on every popped
node): the start states, the weights numpy drew (np.random.rand under a fixed seed — recorded so the device side does not
depend on numpy's generator), and the triple it returns (states_update in the reference's instance-major / pop order,
cost_to_go_update, is_solved).  The heuristic is a deterministic function of the NETWORK-INPUT rows
(`env.state_to_nnet_input`), clipped at zero like the update's heuristic servers (avi.py:213), so the device test can hand
the same function to the engine as its heuristic closure.  Data only — no reference source is stored.
"""
import os
import sys

import numpy as np

np.float = float  # noqa: the reference targets numpy 1.22
np.int = int  # noqa

REF = "/root/reference"
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))

from environments.cube3 import Cube3, Cube3State  # noqa: E402
from environments.lights_out import LightsOut, LOState  # noqa: E402
from environments.n_puzzle import NPuzzle, NPuzzleState  # noqa: E402
from updaters.updater import astar_update  # noqa: E402


def heur_knuth3(s: np.ndarray) -> np.ndarray:
    """DCA_HEUR_KNUTH3 of include/dca.h on uint8 rows: f32(((sum_i s_i (7 i + 3)) * 2654435761 mod 2^32) / 2^32 * 3)."""
    idx = (7 * np.arange(s.shape[1], dtype=np.uint64) + 3)
    sm = (s.astype(np.uint64) * idx).sum(axis=1)
    x = (sm * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    return (x.astype(np.float64) / 4294967296.0 * 3.0).astype(np.float32)


def case(env, mk_state, get_arr, roots, steps, seed):
    def hfn(states, is_nnet_format=False):
        nn = env.state_to_nnet_input(states)[0] if not is_nnet_format else states[0]
        return np.maximum(heur_knuth3(np.asarray(nn, np.uint8)).astype(np.float64), 0.0)

    np.random.seed(seed)
    weights = np.random.rand(len(roots))  # what astar_update will draw (updater.py:37)
    np.random.seed(seed)
    su, ctg, solved = astar_update([mk_state(r.copy()) for r in roots], env, steps, hfn)
    return {"weights": weights.astype(np.float64), "states": np.stack([get_arr(s) for s in su]).astype(np.uint8),
            "ctg": np.asarray(ctg, np.float64), "solved": np.asarray(solved, bool)}


def main():
    out = {}
    rng = np.random.default_rng(4242)
    c3 = Cube3()
    roots = []
    for _ in range(40):
        st = Cube3State(c3.goal_colors.copy())
        for a in rng.integers(0, 12, size=int(rng.integers(0, 5))):
            st = c3.next_state([st], int(a))[0][0]
        roots.append(st.colors.astype(np.uint8))
    roots = np.stack(roots)
    out["cube3_roots"] = roots
    for steps in (1, 4, 12):
        r = case(c3, lambda a: Cube3State(a), lambda s: s.colors, roots, steps, 100 + steps)
        for k, v in r.items():
            out["cube3_steps%d_%s" % (steps, k)] = v
        print("cube3 steps", steps, r["states"].shape, r["ctg"][:4], int(r["solved"].sum()))
    p15 = NPuzzle(4)
    proots = []
    for _ in range(32):
        st = NPuzzleState(p15.goal_tiles.copy())
        for a in rng.integers(0, 4, size=int(rng.integers(0, 7))):
            st = p15.next_state([st], int(a))[0][0]
        proots.append(st.tiles.astype(np.uint8))
    proots = np.stack(proots)
    out["puzzle15_roots"] = proots
    r = case(p15, lambda a: NPuzzleState(a), lambda s: s.tiles, proots, 6, 7)
    for k, v in r.items():
        out["puzzle15_steps6_%s" % k] = v
    print("puzzle15 steps 6", r["states"].shape, int(r["solved"].sum()))
    lo = LightsOut(7)
    lroots = []
    for _ in range(24):
        st = lo.generate_goal_states(1)[0]
        for a in rng.integers(0, 49, size=int(rng.integers(0, 4))):
            st = lo.next_state([st], int(a))[0][0]
        lroots.append(np.asarray(st.tiles, np.uint8))
    lroots = np.stack(lroots)
    out["lightsout7_roots"] = lroots
    r = case(lo, lambda a: LOState(a), lambda s: np.asarray(s.tiles, np.uint8), lroots, 5, 9)
    for k, v in r.items():
        out["lightsout7_steps5_%s" % k] = v
    print("lightsout7 steps 5", r["states"].shape, int(r["solved"].sum()))
    np.savez_compressed(os.path.join(OUT, "astar_update.npz"), **out)
    print("wrote astar_update.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
