"""GPU parity tests of the BASELINE configs that round 1 left unexercised (VERDICT r01 "next round" item 1):

  (a) the puzzle cost-to-go networks through every device path (layer-1 MFMA kernel where instantiated, one-hot rows
      otherwise, f16x3 split layers) against outputs recorded from the reference's ResnetModel;
  (b) the puzzle48 engine at configs[4]'s batch 20 000 / weight 0.6, first iterations traced against the oracle, PY and CPP;
  (c) the AVI update step on >= 1M puzzle48 states through `Updater.update_dev`, checked by size-independent properties
      and shard consistency;
  (d) the north star's 1e-5 heuristic tolerance at TRAINED-network magnitudes (|h| 21-29), judged against the float64
      evaluation of the same weights (SURVEY §7.3's protocol) — where the f16x3 path has the least headroom.
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


# ------------------------------------------------------------------------------------------------ (a) puzzle networks
def _puzzle_net(dim, seed):
    from deepcubea_amd.utils.pytorch_models import ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    D = dim * dim
    net = ResnetModel(D, D, 5000, 1000, 4, 1, True)  # n_puzzle.py:94-98
    load_synthetic_weights(net, seed)
    return net.eval()


@pytest.mark.parametrize("name,dim,seed,src", [("puzzle15", 4, 2025, "golden"), ("puzzle24", 5, 2027, "nets"),
                                                ("puzzle48", 7, 2026, "nets")])
@torch.no_grad()
def test_puzzle_network_paths_match_reference_within_1e5(L, golden, nets, name, dim, seed, src):
    from deepcubea_amd.utils.pytorch_models import FastResnet, fold_batchnorm
    fx = golden if src == "golden" else nets
    x = torch.tensor(fx["%s_resnet_seed%d_x" % (name, seed)]).cuda()
    ref = fx["%s_resnet_seed%d_y" % (name, seed)]
    tol = 1e-5  # the north star's tolerance; |ref| < 1 here, so absolute == relative-to-max(1,|ref|)
    net = _puzzle_net(dim, seed)
    D = dim * dim
    # reference module layout on the device (library fp32 GEMMs), BatchNorm folded
    y_plain = net.cuda()(x)[:, 0].cpu().numpy()
    y_fold = fold_batchnorm(net).cuda()(x)[:, 0].cpu().numpy()
    assert np.max(np.abs(y_plain - ref)) < tol and np.max(np.abs(y_fold - ref)) < tol
    # CLI default: FastResnet, fp32 parity mode = f16x3 split layers; layer 1 from the uint8 rows where the MFMA kernel
    # is instantiated for the geometry, else from one-hot rows (two-plane f16 library layer 1)
    fast = FastResnet(net).cuda()
    assert fast.split
    y_fast = fast(x)[:, 0].cpu().numpy()
    assert np.max(np.abs(y_fast - ref)) < tol, (name, float(np.max(np.abs(y_fast - ref))))
    y_oh = fast.forward_onehot(fast.encode(x))[:, 0].cpu().numpy()  # the engine's packed one-hot rows take this entry
    assert np.max(np.abs(y_oh - ref)) < tol
    assert fast.split_fallbacks == 0
    if L.l1_supported(D, D):
        assert fast.uses_l1_kernel and fast.l1_planes == 3
    # plain fp32 GEMMs (split off) and a larger, ragged batch of random states: split vs native
    native = FastResnet(net, split=False).cuda()
    assert np.max(np.abs(native(x)[:, 0].cpu().numpy() - ref)) < tol
    g = torch.Generator().manual_seed(seed)
    xb = torch.stack([torch.randperm(D, generator=g) for _ in range(3001)]).to(torch.uint8).cuda()
    d = (fast(xb) - native(xb)).abs().max().item()
    assert d < tol, d
    # non-parity modes stay close
    for dt, lim in ((torch.bfloat16, 5e-2), (torch.float16, 1e-2)):
        yl = FastResnet(net, dt).cuda()(x)[:, 0].float().cpu().numpy()
        assert np.max(np.abs(yl - ref)) < lim


@torch.no_grad()
def test_puzzle48_engine_feeds_the_network_rows_it_claims(L, co, nets):
    """configs[4] plumbing: the engine's packed one-hot rows (2401 wide, stride FastResnet.in_pad) or uint8 rows drive the
    same network to the same values as evaluating the kept children directly."""
    from deepcubea_amd.search_methods.engine import BwasEngine
    from deepcubea_amd.utils import nnet_utils
    from deepcubea_amd.utils.pytorch_models import FastResnet
    net = _puzzle_net(7, 2026)
    fast = FastResnet(net).cuda()
    hfn = nnet_utils.get_heuristic_fn_dev(fast, clip_zero=False, batch_size=1 << 17)
    if fast.uses_l1_kernel:
        eng = BwasEngine("puzzle48", 0.6, 512, max_nodes=1 << 20, packed=True)
    else:
        eng = BwasEngine("puzzle48", 0.6, 512, max_nodes=1 << 20, onehot_dtype=fast.onehot_dtype, packed=True,
                         onehot_stride=fast.in_pad)
    root = nets["puzzle48_resnet_seed2026_x"][0]
    eng.reset(root)
    eng.root_commit(hfn(eng.root_nnet_in()))
    for it in range(6):
        nn, oh, src, rows = eng.pop_expand_packed()
        kept = eng.last_children()[src[:rows].long()].contiguous()
        n = (rows + 1023) // 1024 * 1024
        h = hfn(oh[:n], True) if oh is not None else hfn(nn[:n])
        direct = fast(kept)[:, 0]
        assert rows > 0 and float((h[:rows] - direct).abs().max()) < 1e-5
        eng.commit_packed(h.float().contiguous())
    assert eng.status()["iterations"] == 6
    eng.close()


# ------------------------------------------------------------------------------------------------ (b) puzzle48 engine
@pytest.mark.parametrize("sem", ["py", "cpp"])
def test_puzzle48_engine_batch_20000_first_iterations_vs_oracle(L, co, golden, sem):
    from deepcubea_amd.search_methods.engine import BwasEngine
    B, w, hid, iters = 20000, 0.6, 1, 14  # configs[4]: batch 20 000; train.sh weight 0.6; KNUTH3 (few float32 cost ties)
    root = np.ascontiguousarray(golden["puzzle48_test_states"][3])
    semv, osem = (L.SEM_PY, co.SEM_PY) if sem == "py" else (L.SEM_CPP, co.SEM_CPP)
    ref = co.astar("puzzle48", root, w, B, osem, heur_builtin_id=hid, max_iters=iters, trace_cap=iters, stop_on_goal=False)
    eng = BwasEngine("puzzle48", w, B, max_nodes=1 << 22, semantics=semv)
    eng.reset(root)
    if semv == L.SEM_PY:
        eng.root_commit(L.heuristic_builtin(hid, torch.from_numpy(root[None].copy()).cuda()))
    tr = []
    for _ in range(iters):
        eng.run_builtin(hid, 1)
        st = eng.status()
        tr.append((st["open_size"], st["closed_size"], st["nodes_generated"]))
    tr = np.array(tr, np.int64)
    assert tr[-1, 2] >= 4 * B * 4  # several full-size batches were expanded (80 000 children each)
    assert np.array_equal(tr[:, 2], ref["trace"][:, 2])  # nodes generated per iteration: exact in both semantics
    if sem == "py":
        assert np.array_equal(tr, ref["trace"])
    else:
        # cpp: |OPEN| / |CLOSED| may drift by a few entries where equal float32 costs pop in libstdc++'s heap order
        rel = np.abs(tr[:, :2] - ref["trace"][:, :2]) / np.maximum(ref["trace"][:, :2], 1)
        assert rel.max() < 1e-2, rel.max()
    # the hipGraph replay continues the same search
    eng.run_builtin(hid, 6, use_graph=True)
    st = eng.status()
    assert st["iterations"] == iters + 6 and not st["failed"]
    assert st["nodes_generated"] - (1 if sem == "cpp" else 0) == st["nodes_expanded"] * 4
    eng.close()


# ------------------------------------------------------------------------------------------------ (c) AVI update, 1M states
@torch.no_grad()
def test_avi_update_one_million_puzzle48_states(L, co):
    """configs[4]'s update step at size: 2^20 puzzle48 states through `Updater.update_dev` with the puzzle48 network
    (fp32 parity mode) as the target heuristic.  Properties that hold at any size: a solved state backs up to 0, every
    other target is >= 1 (1 + max(h, 0)); targets equal 1 + min over the children of max(h, 0) recomputed on a sample
    through the ORACLE's expansion; shards regenerate the same states whichever rank produces them (index0 keyed RNG)."""
    from deepcubea_amd.updaters.updater import Updater
    from deepcubea_amd.utils import env_utils, nnet_utils
    from deepcubea_amd.utils.pytorch_models import FastResnet
    env = env_utils.get_environment("puzzle48")
    fast = FastResnet(_puzzle_net(7, 2026)).cuda()
    hfn = nnet_utils.get_heuristic_fn_dev(fast, clip_zero=True, batch_size=1 << 18)
    oh = None if fast.uses_l1_kernel else fast.onehot_dtype
    n = 1 << 20
    upd = Updater(env, n, 1000, hfn, 1, update_batch_size=1 << 17, seed=77, onehot_dtype=oh)
    sn, ctg, sv = upd.update_dev()
    assert sn.shape == (n, 49) and ctg.shape == (n, 1) and sv.shape == (n,)
    ctg = ctg[:, 0]
    goal = torch.tensor(np.concatenate((np.arange(1, 49), [0])).astype(np.uint8), device="cuda")
    is_goal = (sn == goal).all(dim=1)
    assert torch.equal(is_goal, sv.bool()) and int(is_goal.sum()) >= 1  # back_max 1000 over 1M states: ~1000 zero-step walks
    assert float(ctg[is_goal].abs().max()) == 0.0
    assert float(ctg[~is_goal].min()) >= 1.0 and bool(torch.isfinite(ctg).all())
    # every row is a permutation of 0..48 (valid puzzle state) — checked on the device
    assert bool((torch.sort(sn.long(), dim=1).values == torch.arange(49, device="cuda")).all())
    # recompute a sample through the oracle's expansion and the same closure
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(5))[:4096]
    st = sn[idx.cuda()].cpu().numpy()
    ch, _, _ = co.expand("puzzle48", st)
    hc = hfn(torch.from_numpy(ch.reshape(-1, 49)).cuda()).view(-1, 4)
    want = 1.0 + hc.min(dim=1).values
    want[torch.from_numpy((st == goal.cpu().numpy()).all(1)).cuda()] = 0.0
    assert float((want - ctg[idx.cuda()]).abs().max()) < 1e-4
    # shard consistency: ranks 0/1 of a 2-rank split regenerate exactly the halves of the single-rank run
    half = Updater(env, n, 1000, hfn, 1, update_batch_size=1 << 17, seed=77, onehot_dtype=oh)
    for r, (i0, ln) in enumerate(((0, n // 2), (n // 2, n // 2))):
        half.local_n, half.index0, half.rank = ln, i0, r
        s2, c2, v2 = half.update_dev()
        assert torch.equal(s2, sn[i0:i0 + ln]) and torch.equal(v2, sv[i0:i0 + ln])
        assert float((c2[:, 0] - ctg[i0:i0 + ln]).abs().max()) < 1e-4
    # multi-step GBFS on a slice keeps the reference's bookkeeping: a solved instance stops contributing states
    upd3 = Updater(env, 1 << 14, 3, hfn, 3, update_batch_size=1 << 14, seed=78, onehot_dtype=oh)
    s3, c3, v3 = upd3.update_dev()
    assert (1 << 14) <= s3.shape[0] <= 3 * (1 << 14) and c3.shape[0] == s3.shape[0] and int(v3.sum()) >= 1


# ------------------------------------------------------------------------------------------------ (d) trained magnitudes
def _rescaled(net, nets, key):
    """fc_out rescaled exactly as make_golden_nets.py did for the reference's module (float32 arithmetic on both sides)."""
    with torch.no_grad():
        s, t = float(nets[key + "_out_scale"]), float(nets[key + "_out_shift"])
        net.fc_out.weight.copy_((net.fc_out.weight * np.float32(s)).float())
        net.fc_out.bias.copy_((net.fc_out.bias * np.float32(s) + np.float32(t)).float())
    return net.eval()


@pytest.mark.parametrize("seed", [2028, 2029, 2030])
@torch.no_grad()
def test_heuristic_tolerance_at_trained_network_magnitudes(L, nets, seed):
    """|h| = 21..29, THREE weight seeds (one seed at 9.54e-6 under a 1e-5 assert was one data point).  One
    float32 ulp of 25 is 1.9e-6; the reference's own CPU fp32 forward is 6.0e-6 .. 7.2e-6 away from the float64 evaluation of
    its weights (recorded by make_golden_nets.py).

    THE FINDING of round 6, measured on the MI355X and NOT tuned away: the north star's 1e-5 ABSOLUTE holds at these magnitudes
    for one seed of three.  CLI-default path (f16x3 parity mode), max |error| vs float64 / vs the reference's fp32 values:
        seed 2028  9.5e-6 / 9.5e-6      seed 2029  1.06e-5 / 1.34e-5      seed 2030  9.5e-6 / 1.14e-5
    (1.34e-5 = 7 fp32 ulps of 29; on the 32x32x16 form of the kernels, earlier in the round: 8.9e-6 / 9.5e-6, 1.08e-5 / 1.53e-5, 8.9e-6 / 1.14e-5).  The reference's own module run on this GPU's fp32 GEMMs is FURTHER from float64 than that
    (1.48e-5 .. 1.59e-5): two correct fp32 evaluations of this network differ by up to ~1.5e-5 at |h| ~ 25 whatever computes
    them, because each is 0.6 - 1.6e-5 from exact arithmetic.  The tolerance this repo states for the parity mode is therefore
    1e-5 * max(1, |h|) per state (`--nnet_dtype` help) — which is 1e-5 absolute wherever |h| <= 1, the regime of
    every reference-recorded network fixture that IS held to 1e-5 absolute (test_puzzle_network_paths_match_reference_within_1e5,
    test_heuristic_forward_matches_reference_within_1e5) — and what this test asserts at trained magnitudes is
      (a) that relative-to-magnitude tolerance, with regression guards in fp32 ulps of max|h| (vs float64 <= 7 ulps: 1.3e-5;
          vs the reference's fp32 values <= 10 ulps: 1.9e-5), for every device path;
      (b) that the parity mode is at least as close to exact arithmetic as the REFERENCE'S OWN MODULE is when it runs on this
          GPU (library fp32 GEMMs) — the yardstick that does not depend on a chosen number;
      (c) the same bits through the engine's packed path."""
    from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel, fold_batchnorm
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    key = "cube3_big_seed%d" % seed
    net = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(net, seed)
    net = _rescaled(net, nets, key)
    x = torch.tensor(nets[key + "_x"]).cuda()
    y32, y64 = nets[key + "_y32"].astype(np.float64), nets[key + "_y64"]
    assert 20.0 < y64.min() and y64.max() < 30.0
    ref_err = float(np.max(np.abs(y32 - y64)))
    assert ref_err < 1e-5
    paths = {
        "module_fp32": net.cuda(),
        "folded_fp32": fold_batchnorm(net).cuda(),
        "fast_native_fp32": FastResnet(net, split=False).cuda(),
        "fast_f16x3 (CLI default)": FastResnet(net).cuda(),
    }
    errs = {}
    for name, m in paths.items():
        y = m(x)[:, 0].double().cpu().numpy()
        errs[name] = (float(np.max(np.abs(y - y64))), float(np.max(np.abs(y - y32))))
    hmax = float(np.max(np.abs(y32)))
    ulp = float(np.spacing(np.float32(hmax)))
    print("seed %d: max abs error vs float64 / vs reference fp32 (reference fp32 vs float64: %.2e; one fp32 ulp of %.1f = %.2e):"
          % (seed, ref_err, hmax, ulp), errs)
    e64, e32 = errs["fast_f16x3 (CLI default)"]
    print("seed %d: parity mode within the north star's 1e-5 ABSOLUTE at |h| 21-29?  vs float64: %s (%.3e), vs reference fp32: %s "
          "(%.3e = %.1f ulps)" % (seed, e64 <= 1e-5, e64, e32 <= 1e-5, e32, e32 / ulp))
    tol_rel = 1e-5 * max(1.0, hmax)
    for name, (a64, a32) in errs.items():
        assert a64 <= tol_rel and a32 <= tol_rel, (name, a64, a32, tol_rel)      # (a) the stated tolerance ...
        assert a64 <= 10.0 * ulp and a32 <= 12.0 * ulp, (name, a64 / ulp, a32 / ulp)  # ... and the fp32-noise regression guards
    assert e64 <= 7.0 * ulp and e32 <= 10.0 * ulp, (e64 / ulp, e32 / ulp)
    assert e64 <= errs["module_fp32"][0], (e64, errs["module_fp32"][0])         # (b) closer to exact than the reference's module on this GPU
    f = paths["fast_f16x3 (CLI default)"]
    assert f.split and f.split_fallbacks == 0
    # the same network the way the engine's dedup-first stepping calls it: uint8 rows through the heuristic closure, the
    # batch padded to a multiple of 1024 rows with stale rows behind the valid ones (engine.step: `nn[:n]`) — a state's
    # value must not depend on the padding, and stays inside the same tolerance at |h| ~ 25
    from deepcubea_amd.utils import nnet_utils
    hfn = nnet_utils.get_heuristic_fn_dev(f, clip_zero=False, batch_size=1 << 17)
    rows = x.shape[0]
    assert x.dtype == torch.uint8
    pad = (rows + 1023) // 1024 * 1024 + 1024
    xp = torch.randint(0, 6, (pad, x.shape[1]), dtype=torch.uint8, device="cuda")
    xp[:rows] = x
    hp = hfn(xp)[:rows].double().cpu().numpy().reshape(-1)
    e64p, e32p = float(np.max(np.abs(hp - y64))), float(np.max(np.abs(hp - y32)))
    print("seed %d: engine packed path (rows padded to %d): vs float64 %.3e, vs reference fp32 %.3e" % (seed, pad, e64p, e32p))
    assert e64p == e64 and e32p == e32   # (c) the very same values: a state's heuristic does not depend on batch or padding
    assert np.array_equal(hp, f(x)[:, 0].double().cpu().numpy())  # bit-identical to the unpadded evaluation


@torch.no_grad()
def test_heuristic_tolerance_at_puzzle48_magnitudes(L, nets):
    """configs[4] is puzzle48, whose trained cost-to-go is O(100-300).  One float32 ulp is 7.6e-6 at 100 and 3.1e-5 at 280:
    the reference's own fp32 forward is 2.5e-4 away from the float64 evaluation of its weights there (1.6e-6 * |h|, recorded by
    make_golden_nets.py), so NO fp32 result can be within 1e-5 ABSOLUTE of another.  The tolerance of the parity mode at
    these magnitudes is therefore 1e-5 * max(1, |h|) per state — against the reference's fp32 values and against float64 —
    which is what this test asserts (and what DESIGN §2 and `--nnet_dtype`'s help say); the f16x3 path is also required to be
    no further from float64 than 2x the reference's own fp32 forward is."""
    from deepcubea_amd.utils import nnet_utils
    from deepcubea_amd.utils.pytorch_models import FastResnet
    key = "puzzle48_big_seed2031"
    net = _rescaled(_puzzle_net(7, 2031), nets, key)
    x = torch.tensor(nets[key + "_x"]).cuda()
    y32, y64 = nets[key + "_y32"].astype(np.float64), nets[key + "_y64"]
    assert 99.0 < y64.min() and y64.max() < 300.0
    ref_rel = float(np.max(np.abs(y32 - y64) / np.abs(y64)))
    ref_abs = float(np.max(np.abs(y32 - y64)))
    assert ref_abs > 1e-5 and ref_rel < 1e-5   # the reference itself: outside 1e-5 absolute, inside 1e-5 * |h|
    fast = FastResnet(net).cuda()
    assert fast.split
    paths = {"module_fp32": net.cuda(), "fast_native_fp32": FastResnet(net, split=False).cuda(), "fast_f16x3 (CLI default)": fast}
    for name, m in paths.items():
        y = m(x)[:, 0].double().cpu().numpy()
        r64 = float(np.max(np.abs(y - y64) / np.maximum(1.0, np.abs(y64))))
        r32 = float(np.max(np.abs(y - y32) / np.maximum(1.0, np.abs(y32))))
        a64 = float(np.max(np.abs(y - y64)))
        print("puzzle48 |h| 100-280, %s: vs float64 %.3e abs = %.3e * |h|; vs reference fp32 %.3e * |h| "
              "(reference fp32 vs float64: %.3e abs = %.3e * |h|)" % (name, a64, r64, r32, ref_abs, ref_rel))
        assert r64 <= 1e-5 and r32 <= 1e-5, (name, r64, r32)
        if "f16x3" in name:
            assert a64 <= 2.0 * ref_abs, (a64, ref_abs)
    assert fast.split_fallbacks == 0
    # through the heuristic closure on padded uint8 rows, as the engine calls it: same bits
    hfn = nnet_utils.get_heuristic_fn_dev(fast, clip_zero=False, batch_size=1 << 17)
    xp = torch.zeros((2048, 49), dtype=torch.uint8, device="cuda")
    xp[:] = torch.arange(49, dtype=torch.uint8, device="cuda")
    xp[:x.shape[0]] = x
    hp = hfn(xp)[:x.shape[0]].double().cpu().numpy().reshape(-1)
    assert np.array_equal(hp, fast(x)[:, 0].double().cpu().numpy())
