#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run in the build container only (needs /root/reference, which never travels to
the GPU box):

    python tests/golden/make_golden.py

What is written is DATA (inputs + the reference's outputs), never reference
source.  Every array below is produced by calling the reference's own
functions:

  environments/cube3.py      Cube3._move_np / expand / is_solved / state_to_nnet_input
  environments/n_puzzle.py   NPuzzle.next_state / expand / is_solved
  utils/pytorch_models.py    ResnetModel.forward
  search_methods/astar.py    AStar (python BWAS) with a deterministic heuristic_fn
  data/*/test/data_0.pkl     the shipped test scrambles + optimal solutions
  results/*/output.txt       the published per-state A* outcomes

The numpy alias shim (np.float / np.int) is needed because the reference
targets numpy 1.22 (requirements.txt:1-3).
"""
import hashlib
import os
import pickle
import re
import sys

import numpy as np

np.float = float  # noqa: reference uses the removed aliases
np.int = int  # noqa

REF = "/root/reference"
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402
from environments.cube3 import Cube3, Cube3State  # noqa: E402
from environments.n_puzzle import NPuzzle, NPuzzleState  # noqa: E402
from search_methods.astar import AStar, get_path  # noqa: E402
from utils.pytorch_models import ResnetModel  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def synth_states(n: int, d: int, seed: int = 0) -> np.ndarray:
    """SURVEY §8d recipe: uniform random permutations of arange(d)."""
    rng = np.random.default_rng(seed)
    return rng.permuted(np.tile(np.arange(d, dtype=np.uint8), (n, 1)), axis=1)


# ----------------------------------------------------------------------------
# deterministic heuristics shared by fixtures, oracle and the HIP library
# (include/dca.h: DCA_HEUR_MOD97 / DCA_HEUR_KNUTH3)
# ----------------------------------------------------------------------------
def heur_mod97(s: np.ndarray) -> np.ndarray:
    k = 7 * np.arange(s.shape[1], dtype=np.int64) + 3
    m = (s.astype(np.int64) * k).sum(1) % 97
    return (m.astype(np.float32) / np.float32(50.0)).astype(np.float32)


def heur_knuth3(s: np.ndarray) -> np.ndarray:
    k = 7 * np.arange(s.shape[1], dtype=np.uint64) + np.uint64(3)
    x = ((s.astype(np.uint64) * k).sum(1) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    return (x.astype(np.float64) / 4294967296.0 * 3.0).astype(np.float32)


def main():
    golden = {}

    # ------------------------------------------------------------------ tables
    c3 = Cube3()
    perm = np.tile(np.arange(54, dtype=np.uint8), (12, 1))
    new_idx = np.zeros((12, 24), np.uint8)
    old_idx = np.zeros((12, 24), np.uint8)
    for a, m in enumerate(c3.moves):
        new_idx[a] = c3.rotate_idxs_new[m]
        old_idx[a] = c3.rotate_idxs_old[m]
        perm[a, c3.rotate_idxs_new[m]] = c3.rotate_idxs_old[m]
    golden["cube3_perm"] = perm  # next[i] = cur[perm[a][i]]
    golden["cube3_rotate_idxs_new"] = new_idx
    golden["cube3_rotate_idxs_old"] = old_idx
    assert sha(perm) == "d090eb61b95bda1d675b1cd60ae1c355c9eee2caaeed1ffb72dabbaeb5e55742", sha(perm)
    for n in (4, 5, 6, 7):
        golden["npuzzle_swap_%d" % n] = NPuzzle(n).swap_zero_idxs.astype(np.uint8)

    # ------------------------------------------------- cube3 kernel vectors
    S = synth_states(1000, 54, 0)
    states = [Cube3State(x.copy()) for x in S]
    exp, tc = c3.expand(states)
    children = np.stack([np.stack([c.colors for c in row]) for row in exp])  # [1000,12,54]
    assert sha(S).startswith("39257674257ac2b9"), sha(S)
    assert sha(children).startswith("2486529cdf6cd993"), sha(children)
    golden["cube3_synth1000_in_sha256"] = np.array(sha(S))
    golden["cube3_synth1000_children_sha256"] = np.array(sha(children))
    golden["cube3_synth64_in"] = S[:64]
    golden["cube3_synth64_children"] = children[:64]
    golden["cube3_synth64_tc"] = np.stack(tc[:64])
    flat = [c for row in exp[:64] for c in row]
    golden["cube3_synth64_nnet_in"] = c3.state_to_nnet_input(flat)[0]
    golden["cube3_synth64_is_solved"] = c3.is_solved(flat)
    # single-action next_state / prev_state
    ns = np.stack([np.stack([s.colors for s in c3.next_state(states[:64], a)[0]]) for a in range(12)])
    ps = np.stack([np.stack([s.colors for s in c3.prev_state(states[:64], a)]) for a in range(12)])
    golden["cube3_synth64_next_state"] = ns  # [12,64,54]
    golden["cube3_synth64_prev_state"] = ps
    # solved detection + children of the goal
    goal = c3.generate_goal_states(1)
    gexp, _ = c3.expand(goal)
    golden["cube3_goal_children"] = np.stack([c.colors for c in gexp[0]])
    golden["cube3_goal_children_is_solved"] = c3.is_solved(gexp[0])
    golden["cube3_goal_is_solved"] = c3.is_solved(goal)

    # ------------------------------------------------- puzzle kernel vectors
    for n in (4, 5, 6, 7):
        env = NPuzzle(n)
        D = n * n
        P = synth_states(1000 if n == 4 else 128, D, 0)
        pst = [NPuzzleState(x.copy().astype(env.dtype)) for x in P]
        nxt = np.stack([np.stack([s.tiles for s in env.next_state(pst, a)[0]]) for a in range(4)], 1)
        nxt = nxt.astype(np.uint8)  # [N,4,D]
        pexp, _ = env.expand(pst)
        pch = np.stack([np.stack([c.tiles for c in row]) for row in pexp]).astype(np.uint8)
        assert np.array_equal(pch, nxt)
        prv = np.stack([np.stack([s.tiles for s in env.prev_state(pst[:64], a)]) for a in range(4)], 1)
        if n == 4:
            assert sha(P).startswith("8fb38fb7ebc8ba32"), sha(P)
            assert sha(nxt).startswith("e968dd97e9a852bf"), sha(nxt)
            golden["puzzle15_synth1000_in_sha256"] = np.array(sha(P))
            golden["puzzle15_synth1000_next4_sha256"] = np.array(sha(nxt))
        golden["puzzle%d_synth64_in" % (D - 1)] = P[:64]
        golden["puzzle%d_synth64_children" % (D - 1)] = nxt[:64]
        golden["puzzle%d_synth64_prev" % (D - 1)] = prv.astype(np.uint8)
        fl = [c for row in pexp[:64] for c in row]
        golden["puzzle%d_synth64_is_solved" % (D - 1)] = env.is_solved(fl)
        golden["puzzle%d_synth64_nnet_in" % (D - 1)] = env.state_to_nnet_input(fl)[0].astype(np.uint8)
        g = env.generate_goal_states(1)
        ge, _ = env.expand(g)
        golden["puzzle%d_goal_children" % (D - 1)] = np.stack([c.tiles for c in ge[0]]).astype(np.uint8)
        golden["puzzle%d_goal_children_is_solved" % (D - 1)] = env.is_solved(ge[0])

    # ------------------------------------------------- shipped test sets
    d = pickle.load(open(REF + "/data/cube3/test/data_0.pkl", "rb"))
    cs = np.stack([s.colors for s in d["states"]]).astype(np.uint8)
    sol = [[c3.moves.index("%s%i" % (f, n)) for f, n in s] for s in d["solutions"]]
    # verify every shipped optimal solution with the reference env
    for s0, mv in zip(d["states"], sol):
        st = Cube3State(s0.colors.astype(np.uint8))
        for a in mv:
            st = c3.next_state([st], a)[0][0]
        assert c3.is_solved([st])[0]
    golden["cube3_test_states"] = cs
    golden["cube3_test_opt_len"] = np.array([len(s) for s in sol], np.int32)
    maxl = max(len(s) for s in sol)
    solarr = -np.ones((len(sol), maxl), np.int8)
    for i, s in enumerate(sol):
        solarr[i, : len(s)] = s
    golden["cube3_test_opt_moves"] = solarr
    for name, n in (("puzzle15", 4), ("puzzle24", 5), ("puzzle35", 6), ("puzzle48", 7)):
        d = pickle.load(open(REF + "/data/%s/test/data_0.pkl" % name, "rb"))
        ts = np.stack([s.tiles for s in d["states"]]).astype(np.uint8)
        golden[name + "_test_states"] = ts
        if "solutions" in d:
            env = NPuzzle(n)
            ps = [[env.moves.index(m) for m in s] for s in d["solutions"]]
            for s0, mv in zip(d["states"], ps):
                st = NPuzzleState(s0.tiles.astype(env.dtype))
                for a in mv:
                    st = env.next_state([st], a)[0][0]
                assert env.is_solved([st])[0]
            golden[name + "_test_opt_len"] = np.array([len(s) for s in ps], np.int32)
            ml = max(len(s) for s in ps)
            arr = -np.ones((len(ps), ml), np.int8)
            for i, s in enumerate(ps):
                arr[i, : len(s)] = s
            golden[name + "_test_opt_moves"] = arr

    # ------------------------------------------------- published A* outcomes
    pat = re.compile(r"State: (\d+), SolnCost: ([\d.]+), # Moves: (\d+), # Nodes Gen: ([\d,]+), Time: ([\d.]+)")
    for name in ("cube3", "puzzle15", "puzzle24", "puzzle35", "puzzle48"):
        rows = []
        for line in open(REF + "/results/%s/output.txt" % name):
            m = pat.search(line)
            if m:
                rows.append((int(m.group(3)), int(m.group(4).replace(",", "")), float(m.group(5))))
        r = np.array(rows, np.float64)
        golden["published_%s_len" % name] = r[:, 0].astype(np.int32)
        golden["published_%s_nodes" % name] = r[:, 1].astype(np.int64)
        golden["published_%s_time" % name] = r[:, 2].astype(np.float32)

    # ------------------------------------------------- heuristic forward
    def det_weights(model, seed):
        """Deterministic NumPy PCG64 weights incl. non-trivial BN running stats."""
        rng = np.random.default_rng(seed)
        sd = model.state_dict()
        new = {}
        for k, v in sd.items():
            shp = tuple(v.shape)
            if k.endswith("num_batches_tracked"):
                new[k] = torch.tensor(7, dtype=torch.long)
            elif k.endswith("running_var"):
                new[k] = torch.tensor(rng.uniform(0.5, 1.5, shp).astype(np.float32))
            elif k.endswith("running_mean"):
                new[k] = torch.tensor(rng.normal(0, 0.1, shp).astype(np.float32))
            elif k.endswith("weight") and len(shp) == 2:
                new[k] = torch.tensor((rng.normal(0, 1.0, shp) / np.sqrt(shp[1])).astype(np.float32))
            elif k.endswith("weight"):  # BN gamma
                new[k] = torch.tensor(rng.uniform(0.8, 1.2, shp).astype(np.float32))
            else:  # biases / BN beta
                new[k] = torch.tensor(rng.normal(0, 0.1, shp).astype(np.float32))
        model.load_state_dict(new)
        return {k: v.numpy() for k, v in new.items()}

    torch.set_num_threads(1)
    tiny = ResnetModel(54, 6, 64, 32, 2, 1, True)
    tw = det_weights(tiny, 1234)
    tiny.eval()
    xin = (synth_states(256, 54, 5) // 9).astype(np.uint8)
    with torch.no_grad():
        tout = tiny(torch.tensor(xin)).numpy()[:, 0]
    np.savez_compressed(os.path.join(OUT, "tiny_resnet.npz"), x=xin, y=tout.astype(np.float32),
                        **{"w:" + k: v for k, v in tw.items()})
    # full cube3 architecture: weights regenerated from the seed on both sides
    full = c3.get_nnet_model()
    det_weights(full, 2024)
    full.eval()
    xin2 = (synth_states(64, 54, 6) // 9).astype(np.uint8)
    with torch.no_grad():
        fout = full(torch.tensor(xin2)).numpy()[:, 0]
    golden["cube3_resnet_seed2024_x"] = xin2
    golden["cube3_resnet_seed2024_y"] = fout.astype(np.float32)
    # puzzle15 architecture
    p15 = NPuzzle(4).get_nnet_model()
    det_weights(p15, 2025)
    p15.eval()
    xin3 = synth_states(64, 16, 7)
    with torch.no_grad():
        pout = p15(torch.tensor(xin3)).numpy()[:, 0]
    golden["puzzle15_resnet_seed2025_x"] = xin3
    golden["puzzle15_resnet_seed2025_y"] = pout.astype(np.float32)

    # ------------------------------------------------- dummy heuristics
    golden["heur_mod97_cube3_synth64"] = heur_mod97(S[:64])
    golden["heur_knuth3_cube3_synth64"] = heur_knuth3(S[:64])

    # ------------------------------------------------- reference python A* traces
    def run_ref_astar(env, mk_state, get_arr, root_arr, heur, w, B, max_itr=100000):
        def hfn(states, is_nnet_format=False):
            arr = np.stack([get_arr(s) for s in states]).astype(np.uint8)
            return np.maximum(heur(arr).astype(np.float64), 0.0)

        astar = AStar([mk_state(root_arr)], env, hfn, [w])
        trace = []
        itr = 0
        while not min(astar.has_found_goal()):
            astar.step(hfn, B)
            inst = astar.instances[0]
            trace.append((len(inst.open_set), len(inst.closed_dict), inst.num_nodes_generated))
            itr += 1
            assert itr < max_itr
        goal = astar.get_goal_node_smallest_path_cost(0)
        _, moves, pc = get_path(goal)
        return np.array(trace, np.int64), np.array(moves, np.int32), float(pc), astar.get_num_nodes_generated(0)

    def scramble(env, goal_arr, mk_state, get_arr, moves):
        st = mk_state(goal_arr.copy())
        for a in moves:
            st = env.next_state([st], a)[0][0]
        return get_arr(st).astype(np.uint8)

    cases = []
    c3mk = lambda a: Cube3State(a.astype(np.uint8))  # noqa: E731
    c3get = lambda s: s.colors  # noqa: E731
    for ci, (scr, w, B, hname) in enumerate([
        ([0, 5, 7, 2], 0.8, 50, "mod97"),
        ([1, 3, 8, 10, 4], 0.6, 37, "mod97"),
        ([0, 5, 7, 2], 0.8, 1, "knuth3"),
        ([11, 2, 6, 9, 0, 5], 0.8, 200, "knuth3"),
        ([4, 9, 1], 1.0, 1000, "knuth3"),
        ([], 0.8, 10, "mod97"),
    ]):
        root = scramble(c3, c3.goal_colors, c3mk, c3get, scr)
        heur = heur_mod97 if hname == "mod97" else heur_knuth3
        tr, mv, pc, nn = run_ref_astar(c3, c3mk, c3get, root, heur, w, B)
        print("cube3 trace", scr, w, B, hname, "->", mv.tolist(), pc, nn, len(tr))
        key = "astar_py_cube3_%d" % ci
        golden[key + "_root"] = root
        golden[key + "_cfg"] = np.array([w, B, 0 if hname == "mod97" else 1], np.float64)
        golden[key + "_trace"] = tr
        golden[key + "_moves"] = mv
        golden[key + "_result"] = np.array([pc, nn], np.float64)
        cases.append(key)
    p15e = NPuzzle(4)
    pmk = lambda a: NPuzzleState(a.astype(np.uint8))  # noqa: E731
    pget = lambda s: s.tiles  # noqa: E731
    for ci, (scr, w, B, hname) in enumerate([
        ([1, 3, 1, 1, 3, 0, 2, 0, 3, 1], 0.8, 100, "knuth3"),
        ([1, 1, 3, 3, 0, 2], 0.6, 7, "mod97"),
    ]):
        root = scramble(p15e, p15e.goal_tiles, pmk, pget, scr)
        heur = heur_mod97 if hname == "mod97" else heur_knuth3
        tr, mv, pc, nn = run_ref_astar(p15e, pmk, pget, root, heur, w, B)
        print("puzzle15 trace", scr, w, B, hname, "->", mv.tolist(), pc, nn, len(tr))
        key = "astar_py_puzzle15_%d" % ci
        golden[key + "_root"] = root
        golden[key + "_cfg"] = np.array([w, B, 0 if hname == "mod97" else 1], np.float64)
        golden[key + "_trace"] = tr
        golden[key + "_moves"] = mv
        golden[key + "_result"] = np.array([pc, nn], np.float64)
        cases.append(key)
    golden["astar_py_cases"] = np.array(cases)

    # ------------------------------------------------- AVI update step
    # reference updaters/updater.py:gbfs_update + search_utils.bellman with a deterministic heuristic, eps_max = 0.
    # A revisited state triggers a random move (gbfs.py:112-116): np.random.choice is stubbed to return child 0
    # while the fixture is recorded, so the seen-state logic itself is pinned deterministically.
    from updaters.updater import gbfs_update
    from utils import search_utils

    def upd_case(env, mk_state, get_arr, roots, heur, steps):
        def hfn(states, is_nnet_format=False):
            arr = np.stack([get_arr(s) for s in states]).astype(np.uint8)
            return np.maximum(heur(arr).astype(np.float64), 0.0)
        outs = []
        real_choice = np.random.choice
        calls = [0]

        def stub_choice(n, *a, **k):  # the RNG source is stubbed (not the reference): "random child" := child 0
            calls[0] += 1
            return 0
        np.random.choice = stub_choice
        try:
            for sd in (1, 2):
                np.random.seed(sd)
                su, ctg, solved = gbfs_update([mk_state(r.copy()) for r in roots], env, steps, hfn, 0.0)
                outs.append((np.stack([get_arr(s) for s in su]).astype(np.uint8), np.asarray(ctg, np.float64),
                             np.asarray(solved, bool)))
        finally:
            np.random.choice = real_choice
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
        print("   (random-child draws stubbed to child 0: %d per run)" % (calls[0] // 2))
        return outs[0]

    rng = np.random.default_rng(99)
    # cube3: 96 states, 0..4 random moves from the goal (some are solved / get solved within the steps)
    roots = []
    for i in range(96):
        st = Cube3State(c3.goal_colors.copy())
        for a in rng.integers(0, 12, size=int(rng.integers(0, 5))):
            st = c3.next_state([st], int(a))[0][0]
        roots.append(st.colors.astype(np.uint8))
    roots = np.stack(roots)
    golden["avi_cube3_roots"] = roots
    for steps in (1, 3):
        su, ctg, sv = upd_case(c3, c3mk, c3get, roots, heur_knuth3, steps)
        golden["avi_cube3_steps%d_states" % steps] = su
        golden["avi_cube3_steps%d_ctg" % steps] = ctg
        golden["avi_cube3_steps%d_solved" % steps] = sv
        print("avi cube3 steps", steps, su.shape, ctg[:4], sv.sum())
    # bellman alone (search_utils.py:16-32) on the synthetic states
    hfn64 = lambda sts, is_nnet_format=False: np.maximum(  # noqa: E731
        heur_mod97(np.stack([s.colors for s in sts]).astype(np.uint8)).astype(np.float64), 0.0)
    bk, _, _ = search_utils.bellman([Cube3State(x.copy()) for x in S[:64]], hfn64, c3)
    golden["avi_cube3_bellman_synth64_mod97"] = np.asarray(bk, np.float64)
    # puzzle15
    proots = []
    for i in range(64):
        st = NPuzzleState(p15e.goal_tiles.copy())
        for a in rng.integers(0, 4, size=int(rng.integers(0, 6))):
            st = p15e.next_state([st], int(a))[0][0]
        proots.append(st.tiles.astype(np.uint8))
    proots = np.stack(proots)
    golden["avi_puzzle15_roots"] = proots
    su, ctg, sv = upd_case(p15e, pmk, pget, proots, heur_knuth3, 2)
    golden["avi_puzzle15_steps2_states"] = su
    golden["avi_puzzle15_steps2_ctg"] = ctg
    golden["avi_puzzle15_steps2_solved"] = sv

    np.savez_compressed(os.path.join(OUT, "golden.npz"), **golden)
    print("wrote", os.path.join(OUT, "golden.npz"), "with", len(golden), "arrays")


if __name__ == "__main__":
    main()
