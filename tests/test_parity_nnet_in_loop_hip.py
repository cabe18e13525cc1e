"""configs[2] AS WRITTEN: the cost-to-go network in the loop at batch 20 000, followed by the oracle.

The trace tests of test_parity_at_scale_hip.py drive the engine with a built-in heuristic; the path the metric is quoted
on — dedup-first stepping (`k_pack`, `h[kept_pos[j]]` in `k_commit`), the padded `FastResnet` on the hand-written f16x3
kernels, the full-size `ResnetModel(54, 6, 5000, 1000, 4, 1, True)` — was only compared with the oracle at batch 60.  Here:

 (a) cube3, weight 0.8, batch 20 000, synthetic full-size weights, 40 iterations, through the CLI-default path
     (dedup-first + FastResnet f16x3) AND through `--eval_all_children` (reference order, the plain module on the
     one-hot rows the expansion launch writes): |OPEN|, |CLOSED|, nodes generated after every iteration equal
     `oracle.astar(..., heur_fn=closure)` (oracle/dca_oracle.cpp restating astar.py:50-90,180-203,256-333), where the
     closure runs the SAME device network on all 240 000 children of the oracle's batch;
 (b) the same for puzzle48, weight 0.6 (configs[4]'s geometry, train.sh's weight), 20 iterations;
 (c) what (a) rests on, checked directly: a state's heuristic value has the same BITS whatever row, batch size or
     chunking it is evaluated in (packed 197 k rows vs padded 240 k rows vs 1024-row chunks) — every kernel of the
     forward walks K in a fixed order per row, including the output layer (`dca_head_gemv`).
Reference: astar.py:256-317 (AStar.step), nnet_utils.py:156-198 (heuristic closure).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from deepcubea_amd import _lib
    _lib.require_gpu()
    return _lib


@pytest.fixture(scope="module")
def co():
    from oracle import c_oracle
    return c_oracle


def _net(env_name, seed):
    from deepcubea_amd.utils import env_utils
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    env = env_utils.get_environment(env_name)
    net = env.get_nnet_model()  # cube3.py:87-91 / n_puzzle.py:94-98: ResnetModel(D, depth, 5000, 1000, 4, 1, True)
    load_synthetic_weights(net, seed)
    return env, net.cuda().eval()


def _trace_with_network(L, co, env_name, root, w, B, iters, mode, seed):
    """-> (engine trace, oracle trace, network rows the engine evaluated, rows the oracle's closure evaluated)."""
    from deepcubea_amd.search_methods import astar as cli
    from deepcubea_amd.search_methods.engine import BwasEngine
    from deepcubea_amd.utils import nnet_utils
    from deepcubea_amd.utils.pytorch_models import FastResnet
    env, net = _net(env_name, seed)
    _, _, D, A, _ = L.env_ids(env_name)
    M = B * A
    rows = max(10000, cli._MIN_NNET_ROWS)  # what `--nnet_batch_size 10000` (train.sh) becomes in the CLI (_nnet_rows)
    if mode == "dedup_first":  # astar.py's default here: _load_heuristic + BwasEngine(packed=True)
        fast = FastResnet(net).cuda()
        assert fast.split and fast.gemm == "hip"
        hfn = nnet_utils.get_heuristic_fn_dev(fast, clip_zero=False, batch_size=rows)
        if fast.uses_l1_kernel:
            eng = BwasEngine(env_name, w, B, max_nodes=iters * M + (1 << 20), packed=True)
        else:
            eng = BwasEngine(env_name, w, B, max_nodes=iters * M + (1 << 20), packed=True,
                             onehot_dtype=fast.onehot_dtype, onehot_stride=fast.in_pad)
        enc = fast
    else:  # --eval_all_children: the reference's order and the reference's module (BatchNorm unfolded), fp32 one-hot rows
        hfn = nnet_utils.get_heuristic_fn_dev(net, clip_zero=False, batch_size=rows)
        eng = BwasEngine(env_name, w, B, max_nodes=iters * M + (1 << 20), onehot_dtype=torch.float32)
        enc = None
    evaluated = [0]

    def heur(states):  # the oracle's closure: ALL children of its batch, on the same device network
        n = len(states)
        x = torch.from_numpy(np.ascontiguousarray(states // 9 if env_name == "cube3" else states)).cuda()
        evaluated[0] += n
        if enc is None:  # the module takes the uint8 rows (one-hot inside, pytorch_models.py:49-52)
            # same padded batch shape as the engine's fixed M rows, so the library GEMMs see the same problem
            xp = torch.zeros((max(M, n), D), dtype=torch.uint8, device="cuda")
            xp[:n] = x
            return hfn(xp)[:n].cpu().numpy()
        return hfn(x).cpu().numpy()

    ref = co.astar(env_name, root, w, B, co.SEM_PY, heur_fn=heur, max_iters=iters, trace_cap=iters, stop_on_goal=False)
    assert ref["iterations"] == iters
    eng.reset(root)
    eng.root_commit(hfn(eng.root_nnet_in()).to(torch.float32))
    tr = []
    for i in range(iters):
        eng.step(hfn)
        st = eng.status()
        assert not st["failed"], (i, st)
        tr.append((st["open_size"], st["closed_size"], st["nodes_generated"]))
    n_eval = eng.rows_evaluated
    eng.close()
    return np.array(tr, np.int64), ref["trace"], n_eval, evaluated[0]


# ------------------------------------------------------------------------------------------------ (a)
@pytest.mark.parametrize("mode", ["dedup_first", "eval_all_children"])
@torch.no_grad()
def test_cube3_w08_batch_20000_full_network_in_the_loop_vs_oracle(L, co, golden, mode):
    B, w, iters = 20000, 0.8, 40
    root = np.ascontiguousarray(golden["cube3_test_states"][0])
    tr, ref, n_eng, n_ref = _trace_with_network(L, co, "cube3", root, w, B, iters, mode, seed=2024)
    assert tr[-1, 2] >= 30 * B * 12  # full batches: 240 000 children per iteration once OPEN holds 20 000 nodes
    assert np.array_equal(tr[:, 2], ref[:, 2])
    assert np.array_equal(tr, ref), np.argwhere(tr != ref)[:4]
    if mode == "dedup_first":
        assert n_eng < 0.95 * n_ref  # the engine evaluated only the children that survive the CLOSED check
    print("%s: %d iterations, %d nodes generated, engine evaluated %d network rows (oracle closure: %d)"
          % (mode, iters, tr[-1, 2], n_eng, n_ref))


# ------------------------------------------------------------------------------------------------ (b)
@torch.no_grad()
def test_puzzle48_w06_batch_20000_full_network_in_the_loop_vs_oracle(L, co, golden):
    B, w, iters = 20000, 0.6, 20
    root = np.ascontiguousarray(golden["puzzle48_test_states"][3])
    tr, ref, n_eng, n_ref = _trace_with_network(L, co, "puzzle48", root, w, B, iters, "dedup_first", seed=2026)
    assert tr[-1, 2] >= 4 * B * 4
    assert np.array_equal(tr, ref), np.argwhere(tr != ref)[:4]
    assert n_eng < n_ref


# ------------------------------------------------------------------------------------------------ (c)
@pytest.mark.parametrize("env_name,seed", [("cube3", 2024), ("puzzle48", 2026)])
@torch.no_grad()
def test_heuristic_bits_do_not_depend_on_row_or_batch(L, env_name, seed):
    from deepcubea_amd.utils import nnet_utils
    from deepcubea_amd.utils.pytorch_models import FastResnet
    env, net = _net(env_name, seed)
    fast = FastResnet(net).cuda()
    D = L.env_ids(env_name)[2]
    g = torch.Generator().manual_seed(seed)
    n = 240000 if env_name == "cube3" else 80000
    if env_name == "cube3":
        x = torch.randint(0, 6, (n, D), generator=g, dtype=torch.uint8).cuda()
    else:
        x = torch.argsort(torch.rand((n, D), generator=g), dim=1).to(torch.uint8).cuda()
    whole = nnet_utils.get_heuristic_fn_dev(fast)(x)  # ONE call, all rows (the reference order's padded batch)
    assert whole.shape == (n,) and bool(torch.isfinite(whole).all())
    # the CLI's chunking (131 072 rows per call)
    assert torch.equal(nnet_utils.get_heuristic_fn_dev(fast, batch_size=1 << 17)(x), whole)
    # 1024-row chunks (the shape short batches are rounded to)
    assert torch.equal(nnet_utils.get_heuristic_fn_dev(fast, batch_size=1024)(x), whole)
    # a "packed" batch: 82 % of the rows, in another order, rounded up to 1024 rows with zero padding behind them
    perm = torch.randperm(n, generator=g)[: int(0.82 * n)].cuda()
    rows = (perm.numel() + 1023) // 1024 * 1024
    xp = torch.zeros((rows, D), dtype=torch.uint8, device="cuda")
    xp[:perm.numel()] = x[perm]
    packed = nnet_utils.get_heuristic_fn_dev(fast, batch_size=1 << 17)(xp)
    assert torch.equal(packed[:perm.numel()], whole[perm])
    # single rows and ragged tails
    for lo, hi in ((0, 1), (5, 6), (1000, 1003), (n - 257, n)):
        assert torch.equal(nnet_utils.get_heuristic_fn_dev(fast)(x[lo:hi].contiguous()), whole[lo:hi])
    assert fast.split_fallbacks == 0


@torch.no_grad()
def test_head_gemv_matches_float64_and_is_position_independent(L):
    g = torch.Generator().manual_seed(3)
    for dt, tol in ((torch.float32, 2e-6), (torch.bfloat16, 2e-6), (torch.float16, 2e-6)):
        for m, k, n_out in ((1, 1024, 1), (1000, 1024, 1), (4099, 1024, 3), (257, 64, 8), (3, 4, 2)):
            x = torch.randn((m, k), generator=g).to(dt).cuda()
            w = torch.randn((n_out, k), generator=g).cuda()
            b = torch.randn((n_out,), generator=g).cuda()
            y = L.head_gemv(x, w, b)
            ref = x.double() @ w.double().t() + b.double()  # x as stored (exact in float64)
            scale = float((x.double().abs() @ w.double().abs().t()).max()) + 1.0
            assert float((y.double() - ref).abs().max()) <= tol * scale, (dt, m, k, n_out)
            # strided rows (a column window of a wider matrix) and a shuffled batch: same bits per row
            wide = torch.zeros((m, k + 8), dtype=dt, device="cuda")
            wide[:, :k] = x
            assert torch.equal(L.head_gemv(wide[:, :k], w, b), y)
            p = torch.randperm(m, generator=g).cuda()
            assert torch.equal(L.head_gemv(x[p].contiguous(), w, b), y[p])
    assert torch.equal(L.head_gemv(torch.zeros((0, 1024), device="cuda"), torch.zeros((1, 1024), device="cuda"), None),
                       torch.zeros((0, 1), device="cuda"))
