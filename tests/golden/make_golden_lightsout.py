#!/usr/bin/env python3
"""LightsOut fixtures recorded by IMPORTING the reference (build container only; /root/reference never travels):

    python tests/golden/make_golden_lightsout.py   ->  tests/golden/lightsout.npz

Data only — inputs and what the reference's own functions returned for them:
  environments/lights_out.py   LightsOut(7): move_matrix, next_state, expand, is_solved, state_to_nnet_input, goal states
  search_methods/astar.py      AStar (python BWAS) traces on LightsOut with the deterministic heuristics shared with
                               include/dca.h (DCA_HEUR_MOD97 / DCA_HEUR_KNUTH3)
"""
import os
import sys

import numpy as np

np.float = float  # noqa: the reference targets numpy 1.22
np.int = int  # noqa

sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.dirname(os.path.abspath(__file__))

from environments.lights_out import LightsOut, LOState  # noqa: E402
from search_methods.astar import AStar, get_path  # noqa: E402
from make_golden import heur_knuth3, heur_mod97  # noqa: E402


def main():
    g = {}
    env = LightsOut(7)
    g["move_matrix"] = env.move_matrix.astype(np.int16)
    rng = np.random.default_rng(11)
    S = rng.integers(0, 2, size=(64, 49)).astype(np.uint8)
    g["states"] = S
    states = [LOState(s.copy()) for s in S]
    nxt = np.zeros((49, 64, 49), np.uint8)
    for a in range(49):
        ns, tc = env.next_state(states, a)
        assert all(t == 1.0 for t in tc)
        nxt[a] = np.stack([x.tiles for x in ns])
        ps = env.prev_state(ns, a)
        assert all(np.array_equal(p.tiles, s) for p, s in zip(ps, S))  # every move is its own inverse
    g["next_state_all_actions"] = nxt
    exp, tcs = env.expand(states[:8])
    g["expand_children_8"] = np.stack([np.stack([c.tiles for c in row]) for row in exp]).astype(np.uint8)
    assert all(np.all(t == 1.0) for t in tcs)
    goal = env.generate_goal_states(3, np_format=True)
    g["goal"] = goal.astype(np.uint8)
    one = [LOState(goal[0].copy())] + [env.next_state([LOState(goal[0].copy())], a)[0][0] for a in (0, 24)]
    g["is_solved_probe_states"] = np.stack([s.tiles for s in one]).astype(np.uint8)
    g["is_solved_probe"] = env.is_solved(one)
    g["nnet_input_64"] = env.state_to_nnet_input(states)[0]
    g["num_moves"] = np.array(env.get_num_moves())
    net = env.get_nnet_model()
    g["nnet_dims"] = np.array([net.state_dim, net.one_hot_depth, net.fc1.out_features, net.fc2.out_features, net.num_resnet_blocks])

    def run_ref_astar(root_arr, heur, w, B):
        def hfn(sts, is_nnet_format=False):
            arr = np.stack([s.tiles for s in sts]).astype(np.uint8)
            return np.maximum(heur(arr).astype(np.float64), 0.0)
        astar = AStar([LOState(root_arr.copy())], env, hfn, [w])
        trace = []
        while not min(astar.has_found_goal()):
            astar.step(hfn, B)
            inst = astar.instances[0]
            trace.append((len(inst.open_set), len(inst.closed_dict), inst.num_nodes_generated))
            assert len(trace) < 100000
        goal_node = astar.get_goal_node_smallest_path_cost(0)
        _, moves, pc = get_path(goal_node)
        return np.array(trace, np.int64), np.array(moves, np.int32), float(pc), astar.get_num_nodes_generated(0)

    cases = []
    for ci, (scr, w, B, hname) in enumerate([([3, 17, 40, 22], 0.2, 100, "knuth3"), ([8, 30, 45], 0.8, 7, "mod97"),
                                             ([0, 6, 42, 48, 24], 0.5, 1000, "knuth3"), ([], 0.2, 5, "mod97")]):
        st = LOState(goal[0].copy())
        for a in scr:
            st = env.next_state([st], a)[0][0]
        root = st.tiles.astype(np.uint8)
        tr, mv, pc, nn = run_ref_astar(root, heur_mod97 if hname == "mod97" else heur_knuth3, w, B)
        print("lightsout7 trace", scr, w, B, hname, "->", mv.tolist(), pc, nn, len(tr))
        key = "astar_py_lightsout7_%d" % ci
        g[key + "_root"] = root
        g[key + "_cfg"] = np.array([w, B, 0 if hname == "mod97" else 1], np.float64)
        g[key + "_trace"] = tr
        g[key + "_moves"] = mv
        g[key + "_result"] = np.array([pc, nn], np.float64)
        cases.append(key)
    g["astar_py_cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "lightsout.npz"), **g)
    print("wrote", os.path.join(OUT, "lightsout.npz"), {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
