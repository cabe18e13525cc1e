"""Pin the CPU oracle (oracle/) against the reference's own outputs.

Fixtures in tests/golden/ were produced by importing the reference (make_golden.py);
the cpp-semantics known answers are the rows of SURVEY.md Appendix A (outputs of the
reference binary recorded during the survey).  CPU only.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import np_oracle as no
from tests.conftest import synth_states

PUZZLES = [("puzzle15", 4), ("puzzle24", 5), ("puzzle35", 6), ("puzzle48", 7)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------ tables
def test_cube3_perm_table(golden):
    perm = golden["cube3_perm"]
    assert sha(perm) == "d090eb61b95bda1d675b1cd60ae1c355c9eee2caaeed1ffb72dabbaeb5e55742"
    assert np.array_equal(no.cube3_perm_table(), perm)
    assert np.array_equal(co.cube3_perm_table(), perm)
    assert perm[0, :10].tolist() == [2, 5, 8, 1, 4, 7, 0, 3, 6, 9]
    for a in range(12):
        assert (perm[a] != np.arange(54)).sum() == 20
        assert np.array_equal(perm[a][perm[a ^ 1]], np.arange(54))  # rev(a) = a^1
    # scatter pairs of the reference (cube3.py:167) define the same map
    for a in range(12):
        p = np.arange(54)
        p[golden["cube3_rotate_idxs_new"][a]] = golden["cube3_rotate_idxs_old"][a]
        assert np.array_equal(p, perm[a])


@pytest.mark.parametrize("name,n", PUZZLES)
def test_npuzzle_swap_table(golden, name, n):
    g = golden["npuzzle_swap_%d" % n]
    assert np.array_equal(no.npuzzle_swap_table(n), g)
    assert np.array_equal(co.npuzzle_swap_table(n), g)
    if n == 4:
        assert g[0].tolist() == [4, 0, 1, 0] and g[15].tolist() == [15, 11, 15, 14]


# ------------------------------------------------------------------ cube3 ops
def test_cube3_synth1000_sha(golden):
    S = synth_states(1000, 54, 0)
    assert sha(S) == str(golden["cube3_synth1000_in_sha256"])
    want = str(golden["cube3_synth1000_children_sha256"])
    assert sha(no.cube3_expand(S)) == want
    ch, _, _ = co.expand("cube3", S)
    assert sha(ch) == want


def test_cube3_ops_vs_reference(golden):
    S = golden["cube3_synth64_in"]
    ch_ref = golden["cube3_synth64_children"]
    ch, sv, hs = co.expand("cube3", S)
    assert np.array_equal(ch, ch_ref)
    assert np.array_equal(no.cube3_expand(S), ch_ref)
    flat = ch_ref.reshape(-1, 54)
    assert np.array_equal(sv, golden["cube3_synth64_is_solved"])
    assert np.array_equal(co.is_solved("cube3", flat), golden["cube3_synth64_is_solved"])
    assert np.array_equal(no.cube3_nnet_input(flat), golden["cube3_synth64_nnet_in"])
    assert np.array_equal(co.nnet_input("cube3", flat), golden["cube3_synth64_nnet_in"])
    assert np.all(golden["cube3_synth64_tc"] == 1.0)
    assert np.array_equal(hs, no.hash64(flat))
    assert np.array_equal(co.hash64(flat), no.hash64(flat))
    for a in range(12):
        assert np.array_equal(co.next_state("cube3", S, a), golden["cube3_synth64_next_state"][a])
        assert np.array_equal(no.cube3_next_state(S, a), golden["cube3_synth64_next_state"][a])
        assert np.array_equal(no.cube3_prev_state(S, a), golden["cube3_synth64_prev_state"][a])
        assert np.array_equal(co.next_state("cube3", S, a ^ 1), golden["cube3_synth64_prev_state"][a])


def test_cube3_goal(golden):
    goal = np.arange(54, dtype=np.uint8)[None]
    ch, sv, _ = co.expand("cube3", goal)
    assert np.array_equal(ch[0], golden["cube3_goal_children"])
    assert not sv.any() and not golden["cube3_goal_children_is_solved"].any()
    assert co.is_solved("cube3", goal)[0] and golden["cube3_goal_is_solved"][0]
    assert no.cube3_is_solved(goal)[0]


def test_cube3_properties():
    S = synth_states(512, 54, 3)
    for a in range(12):
        n1 = co.next_state("cube3", S, a)
        assert np.array_equal(co.next_state("cube3", n1, a ^ 1), S)  # move o inverse = id
        n4 = S
        for _ in range(4):
            n4 = co.next_state("cube3", n4, a)
        assert np.array_equal(n4, S)  # 4x same quarter turn = id
    ch, _, _ = co.expand("cube3", S)
    for a in range(12):
        assert np.array_equal(ch[:, a], co.next_state("cube3", S, a))  # expand == stack of next_state


def test_cube3_known_answers(golden):
    """Every shipped optimal solution of data/cube3/test solves its state (1000/1000, 20637 moves)."""
    st = golden["cube3_test_states"].copy()
    mv = golden["cube3_test_opt_moves"]
    assert int(golden["cube3_test_opt_len"].sum()) == 20637
    for t in range(mv.shape[1]):
        for a in range(12):
            sel = mv[:, t] == a
            if sel.any():
                st[sel] = co.next_state("cube3", st[sel], a)
    assert co.is_solved("cube3", st).all()


# ------------------------------------------------------------------ puzzles
@pytest.mark.parametrize("name,n", PUZZLES)
def test_npuzzle_ops_vs_reference(golden, name, n):
    P = golden[name + "_synth64_in"]
    ch_ref = golden[name + "_synth64_children"]
    ch, sv, hs = co.expand(name, P)
    assert np.array_equal(ch, ch_ref)
    assert np.array_equal(no.npuzzle_expand(P, n), ch_ref)
    flat = ch_ref.reshape(-1, n * n)
    assert np.array_equal(sv, golden[name + "_synth64_is_solved"])
    assert np.array_equal(no.nnet_input(name, flat), golden[name + "_synth64_nnet_in"])
    assert np.array_equal(co.nnet_input(name, flat), golden[name + "_synth64_nnet_in"])
    assert np.array_equal(hs, no.hash64(flat))
    for a in range(4):
        assert np.array_equal(no.npuzzle_prev_state(P, n, a), golden[name + "_synth64_prev"][:, a])
        assert np.array_equal(co.next_state(name, P, a ^ 1), golden[name + "_synth64_prev"][:, a])
    goal = np.concatenate((np.arange(1, n * n), [0])).astype(np.uint8)[None]
    gch, gsv, _ = co.expand(name, goal)
    assert np.array_equal(gch[0], golden[name + "_goal_children"])
    assert np.array_equal(gsv, golden[name + "_goal_children_is_solved"])
    assert co.is_solved(name, goal)[0]


def test_puzzle15_config1_sha(golden):
    """BASELINE.json configs[0]: puzzle15 next_state on 1k random states (CPU plumbing)."""
    P = synth_states(1000, 16, 0)
    assert sha(P) == str(golden["puzzle15_synth1000_in_sha256"])
    nxt = np.stack([co.next_state("puzzle15", P, a) for a in range(4)], 1)
    assert sha(nxt) == str(golden["puzzle15_synth1000_next4_sha256"])
    assert sha(no.npuzzle_expand(P, 4)) == str(golden["puzzle15_synth1000_next4_sha256"])


@pytest.mark.parametrize("name,n", PUZZLES[:2])
def test_npuzzle_known_answers(golden, name, n):
    st = golden[name + "_test_states"].copy()
    mv = golden[name + "_test_opt_moves"]
    for t in range(mv.shape[1]):
        for a in range(4):
            sel = mv[:, t] == a
            if sel.any():
                st[sel] = co.next_state(name, st[sel], a)
    assert co.is_solved(name, st).all()


# ------------------------------------------------------------------ reference C++ envs
def test_reference_cpp_envs_agree(golden):
    """oracle/_ref = the reference's cpp/environments.cpp compiled in place."""
    if co.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    S = synth_states(256, 54, 11)
    ch, sv = co.ref_expand("cube3", S)
    och, osv, _ = co.expand("cube3", S)
    assert np.array_equal(ch, och) and np.array_equal(sv, osv)
    for name, n in PUZZLES:
        P = synth_states(128, n * n, 12)
        ch, sv = co.ref_expand(name, P)
        och, osv, _ = co.expand(name, P)
        assert np.array_equal(ch, och) and np.array_equal(sv, osv)
        goal = np.concatenate((np.arange(1, n * n), [0])).astype(np.uint8)[None]
        # no-op moves at the goal (blank in the corner: U and L) leave it solved
        assert np.array_equal(co.ref_expand(name, goal)[1], golden[name + "_goal_children_is_solved"])
        assert co.ref_expand(name, goal)[1].tolist() == [True, False, True, False]


# ------------------------------------------------------------------ one-hot / heuristics
def test_onehot_and_builtin_heuristics(golden):
    S = golden["cube3_synth64_in"]
    idx = no.cube3_nnet_input(S)
    oh = no.onehot(idx, 6)
    assert np.array_equal(co.onehot_f32(idx, 6), oh)
    assert oh.sum() == 64 * 54 and oh.shape == (64, 324)
    assert np.array_equal(co.heur_builtin(0, S), golden["heur_mod97_cube3_synth64"])
    assert np.array_equal(co.heur_builtin(1, S), golden["heur_knuth3_cube3_synth64"])
    for hid in range(4):
        assert np.array_equal(co.heur_builtin(hid, S), no.heur_builtin(hid, S))


def test_resnet_forward_restatement(tiny_resnet, golden):
    w = {k[2:]: tiny_resnet[k] for k in tiny_resnet.files if k.startswith("w:")}
    y32 = no.resnet_forward(w, tiny_resnet["x"], 6, 2, np.float32)
    y64 = no.resnet_forward(w, tiny_resnet["x"], 6, 2, np.float64)
    assert np.max(np.abs(y32 - tiny_resnet["y"])) < 1e-5
    assert np.max(np.abs(y64 - tiny_resnet["y"])) < 1e-5
    # full cube3 architecture, weights regenerated from the seed
    shapes = no.resnet_shapes(54, 6, 5000, 1000, 4)
    wf = no.resnet_det_weights(shapes, 2024)
    yf = no.resnet_forward(wf, golden["cube3_resnet_seed2024_x"], 6, 4, np.float64)
    ref = golden["cube3_resnet_seed2024_y"]
    assert np.max(np.abs(yf - ref)) < 1e-5 * max(1.0, np.abs(ref).max())


def test_first_layer_as_an_embedding_sum_reproduces_the_reference_outputs(tiny_resnet, golden):
    """The identity dca_l1_embed rests on (SURVEY 8(f)-2's "embedding-sum layer 1"): onehot(s) . W1^T = the sum of one gathered
    weight column per position.  The oracle's restatement of it (BatchNorm folded, positions in ascending order) gives the
    reference-recorded network outputs — float64 and float32 — for cube3 and for the puzzle geometries it is used on."""
    w = {k[2:]: tiny_resnet[k] for k in tiny_resnet.files if k.startswith("w:")}
    for dt in (np.float64, np.float32):
        y = no.resnet_forward(w, tiny_resnet["x"], 6, 2, dt, l1="embed")
        assert np.max(np.abs(y - tiny_resnet["y"])) < 1e-5
    a = no.l1_embedding_sum(w, tiny_resnet["x"], 6, np.float64)
    x = no.onehot(tiny_resnet["x"], 6, np.float64)
    g = w["bn1.weight"] / np.sqrt(w["bn1.running_var"].astype(np.float64) + 1e-5)
    ref = np.maximum((x @ w["fc1.weight"].astype(np.float64).T + w["fc1.bias"] - w["bn1.running_mean"]) * g + w["bn1.bias"], 0)
    assert np.max(np.abs(a - ref)) < 1e-12
    wf = no.resnet_det_weights(no.resnet_shapes(54, 6, 5000, 1000, 4), 2024)
    yf = no.resnet_forward(wf, golden["cube3_resnet_seed2024_x"], 6, 4, np.float64, l1="embed")
    assert np.max(np.abs(yf - golden["cube3_resnet_seed2024_y"])) < 1e-5
    nets = np.load(os.path.join(os.path.dirname(__file__), "golden", "nets.npz"))
    for name, dim, seed in (("puzzle48", 7, 2026), ("puzzle24", 5, 2027)):
        D = dim * dim
        wp = no.resnet_det_weights(no.resnet_shapes(D, D, 5000, 1000, 4), seed)
        xk = nets["%s_resnet_seed%d_x" % (name, seed)]
        y = no.resnet_forward(wp, xk, D, 4, np.float64, l1="embed")
        assert np.max(np.abs(y - nets["%s_resnet_seed%d_y" % (name, seed)])) < 1e-5
        assert np.max(np.abs(y - no.resnet_forward(wp, xk, D, 4, np.float64))) < 1e-9


# ------------------------------------------------------------------ BWAS, python semantics
def _py_cases(golden):
    return [str(k) for k in golden["astar_py_cases"]]


def test_astar_py_vs_reference_traces(golden):
    for key in _py_cases(golden):
        env = "cube3" if "cube3" in key else "puzzle15"
        root = golden[key + "_root"]
        w, B, hid = golden[key + "_cfg"]
        trace = golden[key + "_trace"]
        pc, nn = golden[key + "_result"]
        r = co.astar(env, root, float(w), int(B), co.SEM_PY, heur_builtin_id=int(hid), trace_cap=len(trace) + 8)
        assert r["solved"]
        assert r["moves"] == golden[key + "_moves"].tolist(), key
        assert r["nodes_generated"] == int(nn), key
        assert r["path_cost"] == pc
        assert r["iterations"] == len(trace)
        assert np.array_equal(r["trace"], trace), key


def test_astar_py_numpy_restatement_small(golden):
    for key in _py_cases(golden):
        trace = golden[key + "_trace"]
        if len(trace) > 200:
            continue
        env = "cube3" if "cube3" in key else "puzzle15"
        w, B, hid = golden[key + "_cfg"]
        r = no.astar_py(env, golden[key + "_root"], lambda s: no.heur_builtin(int(hid), s), float(w), int(B))
        assert r["moves"] == golden[key + "_moves"].tolist()
        assert np.array_equal(r["trace"], trace)


def test_astar_py_callback_heuristic(golden):
    key = "astar_py_cube3_0"
    w, B, hid = golden[key + "_cfg"]
    r = co.astar("cube3", golden[key + "_root"], float(w), int(B), co.SEM_PY,
                 heur_fn=lambda s: no.heur_builtin(int(hid), s), trace_cap=64)
    assert r["moves"] == golden[key + "_moves"].tolist()
    assert np.array_equal(r["trace"], golden[key + "_trace"])


# ------------------------------------------------------------------ BWAS, cpp semantics
def _scramble(env, moves):
    if env == "cube3":
        s = np.arange(54, dtype=np.uint8)[None]
    else:
        n = {"puzzle15": 4, "puzzle48": 7}[env]
        s = np.concatenate((np.arange(1, n * n), [0])).astype(np.uint8)[None]
    for a in moves:
        s = co.next_state(env, s, a)
    return s[0]


# rows of SURVEY.md Appendix A: outputs of the reference binary with heuristic KNUTH3
CPP_KNOWN = [
    ("cube3", [0, 5, 7, 2], 0.8, 50, [3, 6, 4, 1], 3241, 8),
    ("cube3", [1, 3, 8, 10, 4], 0.6, 37, [5, 9, 11, 0, 2], 5234485, 11792),
    ("cube3", [11, 2, 6, 9, 0, 5], 0.8, 200, [4, 1, 8, 7, 3, 10], 5360737, 2237),
    ("puzzle15", [1, 3, 1, 1, 3, 0, 2, 0, 3, 1], 0.8, 100, [0, 2, 1, 3, 1, 2, 0, 0, 2, 0], 9445, 29),
]


@pytest.mark.parametrize("env,scr,w,B,soln,nodes,iters", CPP_KNOWN)
def test_astar_cpp_known_answers(env, scr, w, B, soln, nodes, iters):
    root = _scramble(env, scr)
    if env == "puzzle15":
        assert root.tolist() == [1, 6, 2, 4, 5, 0, 7, 8, 9, 3, 10, 11, 13, 14, 15, 12]
    r = co.astar(env, root, w, B, co.SEM_CPP, heur_builtin_id=1)
    assert r["solved"]
    assert r["moves"] == soln
    assert r["nodes_generated"] == nodes
    assert r["iterations"] == iters


def test_astar_cpp_puzzle48_known_answer():
    root = _scramble("puzzle48", [1, 1, 3, 1, 3, 3, 0, 2, 1, 3, 0, 0])
    r = co.astar("puzzle48", root, 0.6, 64, co.SEM_CPP, heur_builtin_id=1)
    assert r["moves"] == [1, 2, 1, 3, 0, 2, 1, 2, 0, 2, 0, 0]
    # SURVEY: binary 345 897 / survey restatement 345 889 (heap tie order); same libstdc++ heap here
    assert r["nodes_generated"] in (345897, 345889)


# ------------------------------------------------------------------ AVI update step ((f)-1)
def test_avi_update_restatement_vs_reference(golden):
    kn = lambda s: no.heur_builtin(1, s)  # noqa: E731
    for steps in (1, 3):
        su, ctg, sv = no.gbfs_update("cube3", golden["avi_cube3_roots"], steps, kn)
        assert np.array_equal(su, golden["avi_cube3_steps%d_states" % steps])
        assert np.array_equal(ctg, golden["avi_cube3_steps%d_ctg" % steps])
        assert np.array_equal(sv, golden["avi_cube3_steps%d_solved" % steps])
    su, ctg, sv = no.gbfs_update("puzzle15", golden["avi_puzzle15_roots"], 2, kn)
    assert np.array_equal(su, golden["avi_puzzle15_steps2_states"])
    assert np.array_equal(ctg, golden["avi_puzzle15_steps2_ctg"])
    assert np.array_equal(sv, golden["avi_puzzle15_steps2_solved"])
    bk, _, _ = no.bellman("cube3", golden["cube3_synth64_in"], lambda s: no.heur_builtin(0, s))
    assert np.array_equal(bk, golden["avi_cube3_bellman_synth64_mod97"])


def test_split_operand_layers_are_fp32_accurate_by_construction():
    """The arithmetic behind the device's fp32 parity mode, restated on the host (oracle/np_oracle.py): three fp16 products
    over split operands reproduce an fp32-class GEMM, and three bf16 planes reproduce fp32 weights (one-hot layer 1)."""
    rng = np.random.default_rng(0)
    x = np.maximum(rng.standard_normal((64, 1000)), 0).astype(np.float32) * 1.3   # post-ReLU activations
    w = (rng.standard_normal((96, 1000)) * 0.03).astype(np.float32)
    w[:8] *= 1.0e3
    w[8:16] *= 1.0e-3                                                             # unit magnitudes spread over 1e6
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    y3 = no.f16x3_linear(x, w)
    y32 = (x @ w.T).astype(np.float64)                                             # a plain fp32 GEMM for comparison
    colmax = np.abs(ref).max(axis=0)
    e3 = (np.abs(y3 - ref) / colmax).max()
    e32 = (np.abs(y32 - ref) / colmax).max()
    assert e3 < 2e-6 and e3 < 8 * e32 + 1e-7, (e3, e32)
    for planes, tol in ((1, 2.0 ** -8), (2, 2.0 ** -16), (3, 2.0 ** -23)):
        ps = no.bf16_planes(w, planes)
        assert all(np.all((p.view(np.uint32) & 0xFFFF) == 0) for p in ps)           # every plane is a bf16 number
        assert (np.abs(sum(ps) - w) / np.abs(w)).max() <= tol
    # one-hot rows times three planes == the fp32 weights' own row sums, to fp32 accuracy
    idx = rng.integers(0, 6, size=(32, 54))
    oh = no.onehot(idx.astype(np.uint8), 6).astype(np.float64)
    w1 = (rng.standard_normal((40, 324)) * 0.2).astype(np.float32)
    got = sum(oh @ p.astype(np.float64).T for p in no.bf16_planes(w1, 3))
    want = oh @ w1.astype(np.float64).T
    assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()


def test_resnet_restatement_puzzle_nets_and_trained_magnitudes():
    """tests/golden/nets.npz (recorded from the reference's ResnetModel): the puzzle architectures (one-hot depth n*n,
    n_puzzle.py:94-98) and a cube3 network rescaled to trained-network output magnitudes (|h| 21-29)."""
    nets = np.load(os.path.join(os.path.dirname(__file__), "golden", "nets.npz"))
    for name, dim, seed in (("puzzle48", 7, 2026), ("puzzle24", 5, 2027)):
        D = dim * dim
        w = no.resnet_det_weights(no.resnet_shapes(D, D, 5000, 1000, 4), seed)
        y = no.resnet_forward(w, nets["%s_resnet_seed%d_x" % (name, seed)], D, 4, np.float64)
        assert np.max(np.abs(y - nets["%s_resnet_seed%d_y" % (name, seed)])) < 1e-5
    for seed in (2028, 2029, 2030):
        key = "cube3_big_seed%d" % seed
        w = no.resnet_det_weights(no.resnet_shapes(54, 6, 5000, 1000, 4), seed)
        s, t = np.float64(nets[key + "_out_scale"]), np.float64(nets[key + "_out_shift"])
        w["fc_out.weight"] = (w["fc_out.weight"] * np.float32(s)).astype(np.float32)
        w["fc_out.bias"] = (w["fc_out.bias"] * np.float32(s) + np.float32(t)).astype(np.float32)
        y64 = no.resnet_forward(w, nets[key + "_x"], 6, 4, np.float64)
        assert 20.0 < y64.min() and y64.max() < 30.0
        # the oracle's fp64 evaluation IS the fixture's fp64 yardstick (same weights, same arithmetic) ...
        assert np.max(np.abs(y64 - nets[key + "_y64"])) < 1e-9
        # ... and the reference's own fp32 forward sits within the north star's 1e-5 of it even at this magnitude
        assert np.max(np.abs(nets[key + "_y32"] - y64)) < 1e-5
    # puzzle48 at ITS trained magnitudes (|h| 100-280): the reference's fp32 forward is NOT within 1e-5 absolute of float64
    # there (one fp32 ulp is 0.8e-5 .. 3e-5) — it is within 1e-5 * |h|, which is the tolerance the GPU test holds the product to
    key = "puzzle48_big_seed2031"
    w = no.resnet_det_weights(no.resnet_shapes(49, 49, 5000, 1000, 4), 2031)
    s, t = np.float64(nets[key + "_out_scale"]), np.float64(nets[key + "_out_shift"])
    w["fc_out.weight"] = (w["fc_out.weight"] * np.float32(s)).astype(np.float32)
    w["fc_out.bias"] = (w["fc_out.bias"] * np.float32(s) + np.float32(t)).astype(np.float32)
    y64 = no.resnet_forward(w, nets[key + "_x"], 49, 4, np.float64)
    assert 99.0 < y64.min() and y64.max() < 300.0
    assert np.max(np.abs(y64 - nets[key + "_y64"]) / np.abs(y64)) < 1e-11
    d = np.abs(nets[key + "_y32"] - y64)
    assert d.max() > 1e-5 and (d / np.abs(y64)).max() < 1e-5
