"""CPU-side tests: host logic of the mirror, ABI surface, network definition.  No GPU needed."""
import ctypes as C
import io
import os
import pickle
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from deepcubea_amd import _lib
    # the product ABI (include/dca.h) and the test / tuning / profiling hooks kept apart from it (include/dca_debug.h)
    hdr = open(os.path.join(ROOT, "include", "dca.h")).read()
    dbg = open(os.path.join(ROOT, "include", "dca_debug.h")).read()
    product = set(re.findall(r"\b(dca_[a-z0-9_]+)\s*\(", hdr)) - {"dca_engine"}
    hooks = set(re.findall(r"\b(dca_[a-z0-9_]+)\s*\(", dbg)) - {"dca_engine"}
    assert not (product & hooks) and not any("debug" in n or "variant" in n or "profile" in n for n in product)
    declared = sorted(product | hooks)
    assert sorted(_lib.ABI_SYMBOLS) == declared
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), "libdca_hip.so does not export %s" % name
    assert _lib.lib().dca_abi_version() == 5


def test_host_tables_match_golden(golden):
    from deepcubea_amd import _lib
    assert np.array_equal(_lib.cube3_perm_table(), golden["cube3_perm"])
    for n in (4, 5, 6, 7):
        assert np.array_equal(_lib.npuzzle_swap_table(n), golden["npuzzle_swap_%d" % n])
    with pytest.raises(_lib.DcaError):
        _lib.npuzzle_swap_table(9)


def test_env_registry_and_errors():
    from deepcubea_amd import _lib
    from deepcubea_amd.utils import env_utils
    assert _lib.env_ids("cube3") == (0, 0, 54, 12, 6)
    assert _lib.env_ids("puzzle15") == (1, 4, 16, 4, 16)
    assert _lib.env_ids("puzzle48") == (1, 7, 49, 4, 49)
    with pytest.raises(ValueError):
        _lib.env_ids("sokoban")
    env = env_utils.get_environment("cube3")
    assert env.get_num_moves() == 12 and env.moves[0] == "U-1" and env.moves_rev[0] == "U1"
    assert env_utils.get_environment("puzzle24").dim == 5
    if not torch.cuda.is_available():
        with pytest.raises(_lib.DcaError):  # no silent CPU fallback
            env.next_state(env.generate_goal_states(2), 0)


def test_resnet_matches_reference_outputs(tiny_resnet, golden):
    from deepcubea_amd.utils.pytorch_models import ResnetModel, fold_batchnorm
    from oracle import np_oracle as no
    torch.set_num_threads(2)
    m = ResnetModel(54, 6, 64, 32, 2, 1, True)
    sd = {k[2:]: torch.tensor(tiny_resnet[k]) for k in tiny_resnet.files if k.startswith("w:")}
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    m.eval()
    x = torch.tensor(tiny_resnet["x"])
    with torch.no_grad():
        y = m(x)[:, 0].numpy()
        yf = fold_batchnorm(m)(x)[:, 0].numpy()
        yo = m.forward_onehot(torch.tensor(no.onehot(tiny_resnet["x"], 6)))[:, 0].numpy()
    # tolerance stated by the north star: 1e-5 (|h| here is O(1))
    assert np.max(np.abs(y - tiny_resnet["y"])) < 1e-5
    assert np.max(np.abs(yf - tiny_resnet["y"])) < 1e-5
    assert np.array_equal(yo, y)
    # full cube3 architecture with seed-regenerated weights vs the reference's own forward
    full = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    w = no.resnet_det_weights(no.resnet_shapes(54, 6, 5000, 1000, 4), 2024)
    full.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    full.eval()
    with torch.no_grad():
        yy = full(torch.tensor(golden["cube3_resnet_seed2024_x"]))[:, 0].numpy()
    ref = golden["cube3_resnet_seed2024_y"]
    assert np.max(np.abs(yy - ref)) < 1e-5 * max(1.0, float(np.abs(ref).max()))
    # padded / epilogue-fused inference layout: same function (host arithmetic here; the GPU test repeats it on device)
    from deepcubea_amd.utils.pytorch_models import FastResnet
    for model, xin, want in ((m, tiny_resnet["x"], tiny_resnet["y"]), (full, golden["cube3_resnet_seed2024_x"], ref)):
        fast = FastResnet(model)
        assert fast.in_pad % 64 == 0 and fast.res_pad > fast.res_dim
        oh = np.zeros((xin.shape[0], fast.in_pad), np.float32)
        oh[:, :324] = no.onehot(xin, 6)
        yfast = fast.forward_onehot(torch.tensor(oh))[:, 0].numpy()
        assert np.max(np.abs(yfast - want)) < 1e-5 * max(1.0, float(np.abs(want).max()))


def test_reference_pickles_load_through_the_mirror(tmp_path):
    """data/*/test/data_0.pkl name environments.cube3.Cube3State with int64 colors."""
    import sys
    import types
    from deepcubea_amd.utils import data_utils
    # fabricate a pickle with the reference's class paths
    pkg = types.ModuleType("environments")
    mod = types.ModuleType("environments.cube3")

    class Cube3State:  # noqa
        __slots__ = ['colors', 'hash']

        def __init__(self, colors):
            self.colors = colors
            self.hash = None
    Cube3State.__module__ = "environments.cube3"
    Cube3State.__qualname__ = "Cube3State"
    mod.Cube3State = Cube3State
    sys.modules["environments"] = pkg
    sys.modules["environments.cube3"] = mod
    try:
        blob = pickle.dumps({"states": [Cube3State(np.arange(54, dtype=np.int64))]}, protocol=2)
    finally:
        del sys.modules["environments"], sys.modules["environments.cube3"]
    p = tmp_path / "data_0.pkl"
    p.write_bytes(blob)
    d = data_utils.load_pickle(str(p))
    from deepcubea_amd.environments.cube3 import Cube3State as Mine
    assert isinstance(d["states"][0], Mine) and d["states"][0].colors.dtype == np.uint8
    assert d["states"][0].colors.tolist() == list(range(54))


def test_compare_solutions_report(tmp_path, golden, capsys):
    """§8(f)-3: statistics of scripts/compare_solutions.py + the reader of published output.txt logs."""
    from deepcubea_amd.utils import compare_solutions as cs
    lens, nodes, times = golden["published_cube3_len"], golden["published_cube3_nodes"], golden["published_cube3_time"]
    log = tmp_path / "output.txt"
    with open(log, "w") as f:
        f.write("device: cuda:0, devices: [0], on_gpu: True\n")
        for i in range(len(lens)):
            f.write("State: %i, SolnCost: %.2f, # Moves: %i, # Nodes Gen: %s, Time: %.2f\n"
                    % (i, lens[i], lens[i], format(int(nodes[i]), ","), times[i]))
    pub = cs.load_results(str(log))
    assert np.array_equal(pub["lens"], lens) and np.array_equal(pub["num_nodes_generated"], nodes)
    s = cs.summarize(pub)
    assert abs(s["Lengths"]["mean"] - 21.349) < 1e-3 and s["Lengths"]["min"] == 16 and s["Lengths"]["max"] == 24
    assert abs(s["Nodes Generated"]["mean"] - 8185993) < 1
    import pickle
    opt = golden["cube3_test_opt_len"]
    p1 = tmp_path / "opt.pkl"
    pickle.dump({"states": [None] * 1000, "solutions": [[0] * int(k) for k in opt], "times": [1.0] * 1000,
                 "num_nodes_generated": [10.0] * 1000}, open(p1, "wb"))
    c = cs.compare(cs.load_results(str(p1)), pub)
    assert abs(c["pct_equal"] - 65.0) < 0.05 and c["length_diff"]["max"] == 4  # 65.0 % optimal, +2/+4 otherwise
    cs.main(["--soln1", str(p1), "--soln2", str(log)])
    out = capsys.readouterr().out
    assert "1000 states" in out and "65.00% soln2 equal to soln1" in out and "-Nodes/Sec-" in out


def test_train_nnet_matches_reference_run():
    """Training step against tests/golden/train_nnet.npz, recorded from the reference's own
    nnet_utils.train_nnet (CPU, seeded): same batches, same Adam/lr schedule -> same final weights and last loss."""
    import random
    from deepcubea_amd.utils import nnet_utils
    from deepcubea_amd.utils.pytorch_models import ResnetModel
    torch.set_num_threads(1)
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_nnet.npz"))
    for tag, bn in (("bn", True), ("nobn", False)):
        net = ResnetModel(54, 6, 64, 32, 2, 1, bn)
        net.load_state_dict({k.split(":", 2)[2]: torch.tensor(g[k]) for k in g.files if k.startswith(tag + ":init:")})
        bs, itrs, itr0, lr, lr_d = g[tag + ":args"]
        np.random.seed(7)
        random.seed(7)
        last = nnet_utils.train_nnet(net, [g[tag + ":x"]], g[tag + ":y"], torch.device("cpu"), int(bs), int(itrs),
                                     int(itr0), float(lr), float(lr_d), display=False)
        assert abs(last - float(g[tag + ":last_loss"])) < 1e-5 * max(1.0, abs(last))
        for k, v in net.state_dict().items():
            want = g["%s:final:%s" % (tag, k)]
            assert np.allclose(v.numpy(), want, rtol=1e-5, atol=1e-6), (tag, k)
    with pytest.raises(ValueError):
        nnet_utils.train_nnet(net, [g["nobn:x"][:3]], g["nobn:y"][:3], torch.device("cpu"), 8, 1, 0, 0.1, 1.0, False)


def test_network_weight_layouts_for_the_mfma_paths():
    """Host-side preparation of the heuristic network's device layouts: the bf16 plane tiles of the layer-1 kernel and the
    per-unit-scaled fp16 split weights of the f16x3 layers reconstruct the fp32 weights."""
    from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel, l1_weight_tiles
    g = torch.Generator().manual_seed(0)
    w = torch.randn(128, 324, generator=g) * torch.logspace(-3, 2, 128)[:, None]  # rows spanning 5 decades
    for planes, tol in ((1, 2.0 ** -8), (2, 2.0 ** -16), (3, 2.0 ** -23)):
        t = l1_weight_tiles(w, planes, 336)
        assert t.shape == (2, planes, 42, 64, 8) and t.dtype == torch.bfloat16
        # [ntile][plane][k/8][n][8] -> sum of planes, back to [n_pad, k_pad]
        rec = t.float().sum(dim=1).permute(0, 2, 1, 3).reshape(128, 336)
        assert torch.all(rec[:, 324:] == 0)
        assert float(((rec[:, :324] - w).abs() / w.abs().clamp_min(1e-30)).max()) <= tol
    m = ResnetModel(54, 6, 64, 32, 2, 1, True).eval()
    with torch.no_grad():
        m.bn2.weight[:8] *= 1e3  # spread the folded unit magnitudes
    f = FastResnet(m, gemm="library")  # (each mode keeps only the operand layout it reads)
    assert f.split and len(f.split_w) == 5 and len(f.split_wh) == 0 and f.onehot_dtype == torch.float16 and f.in_pad == 384
    fh = FastResnet(m)  # default: the hand-written kernel's planes
    assert fh.split and len(fh.split_w) == 0 and len(fh.split_wh) == 5 and len(fh.split_wl) == 5
    for w3, wh, wl in zip(f.split_w, fh.split_wh, fh.split_wl):
        v = w3.view(w3.shape[0], -1, 3)
        assert torch.equal(v[:, :, 0], wh) and torch.equal(v[:, :, 2], wl)
    from deepcubea_amd.utils.pytorch_models import fold_batchnorm
    fm = fold_batchnorm(m)
    lins = [fm.fc2] + [l for blk in fm.blocks for l in (blk[0], blk[2])]
    for w3, alpha, bias, lin in zip(f.split_w, f.split_alpha, f.split_b, lins):
        n, k = lin.weight.shape
        v = w3.float().view(w3.shape[0], -1, 3)            # W3[:, 3k..3k+2] = (wh, wh, wl)
        assert torch.equal(v[:, :, 0], v[:, :, 1])
        rec = (v[:, :, 0] + v[:, :, 2]) * alpha[:, None]  # undo the per-unit power-of-two scale
        assert torch.all(torch.log2(alpha) == torch.round(torch.log2(alpha)))
        err = (rec[:n, :k] - lin.weight.detach()).abs().amax(dim=1) / lin.weight.detach().abs().amax(dim=1)
        assert float(err.max()) <= 2.0 ** -20 and torch.all(rec[n:] == 0) and torch.allclose(bias[:n], lin.bias.detach())
    oh = torch.zeros(5, f.in_pad)
    oh[:, ::6] = 1.0
    with torch.no_grad():  # host path of the same module = plain fp32 GEMMs
        assert torch.allclose(f.forward_onehot(oh), m.forward_onehot(oh[:, :324]), atol=1e-5)


def test_layer1_kernel_choice_by_geometry_and_mode():
    """`FastResnet(l1="auto")`: the embedding sum (dca_l1_embed) where the one-hot depth makes it the faster kernel — the
    sliding puzzles in the fp32 parity mode, the larger ones in bf16 too — the one-hot MFMA kernel for cube3; the table handed
    to the kernel is the folded first-layer weight matrix, transposed (bf16 mode: its bf16-rounded values)."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel, fold_batchnorm, L1_EMBED_MIN_DEPTH
    assert L1_EMBED_MIN_DEPTH[torch.float32] == 16 and L1_EMBED_MIN_DEPTH[torch.bfloat16] == 25
    for d, depth in ((54, 6), (16, 16), (25, 25), (36, 36), (49, 49), (49, 6)):
        assert _lib.l1_embed_supported(d, depth)
    assert not _lib.l1_embed_supported(54, 7)

    def net(d, depth):
        m = ResnetModel(d, depth, 130, 64, 1, 1, True).eval()
        with torch.no_grad():
            m.bn1.running_mean.uniform_(-1, 1), m.bn1.running_var.uniform_(0.5, 2.0)
        return m

    cube, p15, p24 = net(54, 6), net(16, 16), net(25, 25)
    assert FastResnet(cube).l1_embed_w is None and FastResnet(cube, torch.bfloat16).l1_embed_w is None
    assert FastResnet(cube, l1="embed").l1_embed_w is not None  # on request (slower there: csrc/dca_embed.hip)
    f15 = FastResnet(p15)
    assert f15.l1_embed_w is not None and f15.uses_l1_kernel and FastResnet(p15, torch.bfloat16).l1_embed_w is None
    assert FastResnet(p15, l1="mfma").l1_embed_w is None
    f24 = FastResnet(p24, torch.bfloat16)
    assert FastResnet(p24).l1_embed_w is not None and f24.l1_embed_w is not None
    w1 = fold_batchnorm(p15).fc1.weight.detach()
    assert f15.l1_embed_w.shape == (256, 192) and torch.equal(f15.l1_embed_w[:, :130], w1.t()) and torch.all(f15.l1_embed_w[:, 130:] == 0)
    w1 = fold_batchnorm(p24).fc1.weight.detach()
    assert torch.equal(f24.l1_embed_w[:, :130], w1.t().to(torch.bfloat16).float())
    with pytest.raises(ValueError):
        FastResnet(ResnetModel(20, 5, 130, 64, 1, 1, True).eval(), l1="embed")
    lo = net(49, 6)  # lightsout7: only the embedding kernel exists for it; taken in the fp32 mode, not in bf16
    assert FastResnet(lo).l1_embed_w is not None and FastResnet(lo).l1_tiles is None and FastResnet(lo).uses_l1_kernel
    assert FastResnet(lo, torch.bfloat16).l1_embed_w is None and not FastResnet(lo, torch.bfloat16).uses_l1_kernel
    assert FastResnet(lo, l1="mfma").l1_embed_w is None


def test_make_batches_drops_the_tail_like_the_reference():
    from deepcubea_amd.utils import nnet_utils
    np.random.seed(0)
    b = nnet_utils.make_batches(10, 4)
    assert [len(x) for x in b] == [4, 4] and len(set(np.concatenate(b).tolist())) == 8
    assert nnet_utils.make_batches(3, 4) == []


def test_cli_surfaces_match_the_reference_parsers():
    """Every option of the reference's `search_methods/astar.py` and `ctg_approx/avi.py` parsers (tests/golden/cli_flags.json,
    recorded by intercepting the reference's own parse_args) exists here with the same flag, type, requiredness and default
    — except `--language`, whose default is this package's only core (`hip`)."""
    import json
    from deepcubea_amd.ctg_approx import avi
    from deepcubea_amd.search_methods import astar
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "cli_flags.json")))
    for name, parser in (("astar", astar.build_parser()), ("avi", avi.build_parser())):
        mine = {a.dest: a for a in parser._actions if a.option_strings and a.dest != "help"}
        for dest, want in ref[name].items():
            assert dest in mine, (name, dest)
            a = mine[dest]
            assert sorted(a.option_strings) == want["flags"] and bool(a.required) == want["required"], (name, dest)
            assert getattr(a.type, "__name__", None) == want["type"] and type(a).__name__ == want["action"], (name, dest)
            if (name, dest) != ("astar", "language"):
                assert a.default == want["default"], (name, dest, a.default, want["default"])
    assert astar.build_parser().get_default("language") == "hip" and ref["astar"]["language"]["default"] == "python"


def test_updater_shards_states_like_split_evenly(monkeypatch):
    """Updater: rank r owns a contiguous slice of the update's states, sizes as misc_utils.split_evenly (misc_utils.py:29-36),
    and its counter-RNG stream starts at the slice's first global index (no GPU needed for the bookkeeping)."""
    from deepcubea_amd.search_methods import sharding
    from deepcubea_amd.updaters.updater import Updater

    class _Env:
        pass

    seen = []
    for world, n in ((1, 10), (3, 10), (8, 50_000_003)):
        tot, nxt = 0, 0
        for rank in range(world):
            monkeypatch.setattr(sharding, "world_info", lambda w=world, r=rank: (w, r))
            u = Updater(_Env(), n, 30, None, 1)
            assert u.index0 == nxt and u.local_n in (n // world, n // world + 1)
            nxt += u.local_n
            tot += u.local_n
            seen.append(u.local_n)
        assert tot == n
    assert seen[1:4] == [4, 3, 3]
    assert Updater(_Env(), 10, 30, None, 1, update_method="astar").method == "ASTAR"  # updater.py:70-71
    with pytest.raises(ValueError):
        Updater(_Env(), 10, 30, None, 1, update_method="bfs")  # updater.py:72-73 "Unknown update method"


def test_results_pickle_names_the_reference_classes(tmp_path):
    """results.pkl must load in the reference tree — `dump_pickle` writes this package's State classes
    under `environments.cube3.Cube3State` / `environments.n_puzzle.NPuzzleState` (same __slots__ on both sides)."""
    import pickletools
    from deepcubea_amd.environments.cube3 import Cube3State
    from deepcubea_amd.environments.n_puzzle import NPuzzleState
    from deepcubea_amd.utils import data_utils
    res = {"states": [Cube3State(np.arange(54, dtype=np.uint8))], "paths": [[NPuzzleState(np.arange(16, dtype=np.uint8))]],
           "solutions": [[1, 2]], "times": [0.5], "num_nodes_generated": [12]}
    p = tmp_path / "results.pkl"
    data_utils.dump_pickle(res, str(p))
    names = [arg for op, arg, _ in pickletools.genops(p.read_bytes()) if op.name == "GLOBAL"]
    assert "environments.cube3 Cube3State" in names and "environments.n_puzzle NPuzzleState" in names
    assert not any(n.startswith("deepcubea_amd") for n in names)
    # a reader that only knows the reference's module layout (stand-in classes with the reference's __slots__)
    import sys
    import types
    mods = {}
    for mname, cname, slot in (("environments.cube3", "Cube3State", "colors"), ("environments.n_puzzle", "NPuzzleState", "tiles")):
        cls = type(cname, (), {"__slots__": [slot, "hash"], "__module__": mname})
        mod = types.ModuleType(mname)
        setattr(mod, cname, cls)
        mods[mname] = mod
    mods["environments"] = types.ModuleType("environments")
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        back = pickle.load(open(p, "rb"))
    finally:
        for k, v in saved.items():
            if v is None:
                del sys.modules[k]
            else:
                sys.modules[k] = v
    assert type(back["states"][0]).__module__ == "environments.cube3" and back["states"][0].colors.tolist() == list(range(54))
    assert back["paths"][0][0].tiles.tolist() == list(range(16)) and back["solutions"] == [[1, 2]]
    # and back through this package's loader
    mine = data_utils.load_pickle(str(p))
    assert isinstance(mine["states"][0], Cube3State) and mine["states"][0] == res["states"][0]


def test_compare_solutions_refuses_misaligned_sets():
    from deepcubea_amd.utils import compare_solutions as cs
    a = {"lens": np.array([5, 6, 7, 8]), "times": np.ones(4), "num_nodes_generated": np.ones(4) * 10}
    b = {"lens": np.array([7, 8]), "times": np.ones(2), "num_nodes_generated": np.ones(2) * 10}
    with pytest.raises(ValueError):
        cs.compare(a, b)
    c = cs.compare(a, b, offset=2)  # a --start_idx 2 run
    assert c["num_states"] == 2 and c["pct_equal"] == 100.0
    with pytest.raises(ValueError):
        cs.compare(a, b, offset=3)


def test_fp8_network_bookkeeping_on_the_host():
    """Fp8Resnet's host-side preparation (no GPU, no kernels): FastResnet's bias folding undone into explicit biases, one e4m3
    weight matrix + one scale per output unit per dense layer.  Evaluated here in float64 WITHOUT activation quantisation, the
    dequantised layers must reproduce the fp32 network up to the e4m3 rounding of the weights (3 mantissa bits: a few percent
    after averaging over the fan-in) — an unfolding or scale mix-up would be off by O(1)."""
    import torch
    from deepcubea_amd.utils.pytorch_models import FastResnet, Fp8Resnet, ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    net = ResnetModel(54, 6, 300, 120, 3, 1, True)
    load_synthetic_weights(net, 7)
    net.eval()
    f8 = Fp8Resnet(net)
    ref = FastResnet(net, torch.float32, split=False)
    assert len(f8.w8) == 1 + 2 * 3 and all(w.dtype == torch.float8_e4m3fn for w in f8.w8)
    assert all(float(w.float().abs().max()) <= 448.0 for w in f8.w8)
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 6, (64, 54), dtype=torch.uint8, generator=g)
    oh = torch.nn.functional.one_hot(x.long(), 6).double().view(64, -1)
    want = ref.forward_onehot(torch.nn.functional.pad(oh.float(), (0, ref.in_pad - oh.shape[1])))[:, 0].double()
    # float64 evaluation of the prepared layers: layer 1 from the kept bf16-rounded fp32 weights, then (w8 * scale, bias)
    h = torch.relu(oh @ f8._w1.double().t() + f8._b1.double())
    W = [w.float().double() * s.double()[:, None] for w, s in zip(f8.w8, f8.w_scale)]
    B = [b.double() for b in f8.bias]
    xx = torch.relu(h @ W[0].t() + B[0])
    for i in range(3):
        hh = torch.relu(xx @ W[1 + 2 * i].t() + B[1 + 2 * i])
        xx = torch.relu(xx + hh @ W[2 + 2 * i].t() + B[2 + 2 * i])
    got = (xx @ f8.base.w_out.double().t() + f8.base.b_out.double())[:, 0]
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) / scale < 0.08
    assert float(torch.corrcoef(torch.stack([got, want]))[0, 1]) > 0.995
    with pytest.raises(RuntimeError):
        f8(x)  # no host path: the fp8 mode runs on the GPU only


def test_graph_chunk_plan_covers_any_run_and_repeats_in_a_steady_search():
    """dca_engine_run_builtin cuts a run of iterations into hipGraph chunks (host logic, no device): the chunks tile the run
    exactly, never exceed 64 iterations, mark as rebase iterations exactly the first 8 of a search and every 16th after, are
    cut at the period boundaries — and a steady search therefore replays the same few (length, pattern) graphs."""
    from deepcubea_amd import _lib
    L = _lib.lib()

    def plan(start, run):
        out, it, left = [], start, run
        while left > 0:
            n, mask = C.c_int(0), C.c_uint64(0)
            _lib.check(L.dca_engine_plan_chunk(C.c_int64(it), int(left), C.byref(n), C.byref(mask)), "dca_engine_plan_chunk")
            assert 1 <= n.value <= min(left, 64)
            for i in range(n.value):
                assert bool((mask.value >> i) & 1) == ((it + i) < 8 or (it + i) % 16 == 0), (it, i)
            assert mask.value >> n.value == 0
            out.append((n.value, mask.value))
            it, left = it + n.value, left - n.value
        return out

    assert plan(0, 8) == [(8, 0xFF)]                       # the ramp: every iteration rebases
    assert plan(8, 20) == [(8, 0), (12, 1)]                # the driver's window (bench.py --steps 20 --warmup 5): two graphs
    assert plan(8, 200)[:2] == [(8, 0), (64, 0x0001000100010001)]
    for start, run in [(0, 1), (0, 300), (5, 40), (16, 16), (17, 15), (17, 16), (31, 1), (100, 999), (8, 64)]:
        chunks = plan(start, run)
        assert sum(n for n, _ in chunks) == run
        bounds = np.cumsum([start] + [n for n, _ in chunks])[1:-1]
        # an interior cut sits on a period boundary (or ends the ramp), unless the run itself ends there
        assert all(b % 16 == 0 or b == 8 for b in bounds), (start, run, chunks)
    # a long steady search: after the first partial period every chunk is one of at most two shapes
    assert len(set(plan(8, 5000)[1:-1])) == 1


def test_power_of_two_operand_scaling_keeps_fp32_products_on_fp16_planes():
    """The arithmetic behind `_lib.linear_train` (training-step GEMMs through dca_f16x3_gemm), restated in numpy: an operand is
    multiplied by the power of two that brings its largest magnitude into [2^14, 2^15) (exact), split into hi = fp16(x) and
    lo = fp16(x - hi); the product is xh*wh + xh*wl + xl*wh accumulated in fp32.  Against float64: elements within 2^-10 of the
    largest keep all 22 bits of the split; smaller ones are held to 2^-39 of the largest (fp16 subnormal steps) — and an
    UNSCALED split of 1e-7-sized gradients flushes them, which is why the scaling is there."""
    rng = np.random.default_rng(5)

    def pow2_scale(amax):
        return 2.0 ** (14 - np.floor(np.log2(amax)))

    def planes(x):
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
        return hi.astype(np.float64), lo.astype(np.float64)

    g = (rng.standard_normal((64, 512)) * 1e-7 * np.exp2(-20.0 * rng.random((64, 1)))).astype(np.float32)  # gradients, 20 binades of rows
    w = (rng.standard_normal((512, 128)) / 512 ** 0.5).astype(np.float32)
    w[::7] *= 1e-3
    sg = pow2_scale(np.abs(g).max())
    sw = pow2_scale(np.abs(w).max(axis=0))  # one scale per output unit (the GEMM's rows of W^T)
    gs, ws = (g * np.float32(sg)).astype(np.float32), (w * sw.astype(np.float32)[None, :]).astype(np.float32)
    assert np.array_equal(gs.astype(np.float64), g.astype(np.float64) * sg)  # a power of two moves the exponent only
    assert 2.0 ** 14 <= np.abs(gs).max() < 2.0 ** 15 and np.all((np.abs(ws).max(axis=0) >= 2.0 ** 14) & (np.abs(ws).max(axis=0) < 2.0 ** 15))
    gh, gl = planes(gs)
    wh, wl = planes(ws)
    # per element: the split's error is at most 2^-22 of the element (both planes normal) or half a subnormal step
    err = np.abs(gh + gl - gs.astype(np.float64))
    assert np.all(err <= np.maximum(np.abs(gs) * 2.0 ** -21, 2.0 ** -25))
    got = (gh @ wh + gh @ wl + gl @ wh) / sg / sw[None, :]
    want = g.astype(np.float64) @ w.astype(np.float64)
    rows = np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)
    assert rows.max() < 1e-5 and np.median(rows) < 1e-6, (rows.max(), np.median(rows))
    # the same split without the scaling: 1e-7-sized values sit in fp16's subnormal range (or below it)
    uh, ul = planes(g)
    assert np.abs(uh + ul - g.astype(np.float64)).max() / np.abs(g).max() > 1e-3


def test_auto_instances_rule(monkeypatch):
    """--instances_per_gpu auto (search_methods/astar.py:auto_instances): enough instances that a launch / a network call
    has a chip-filling amount of work, bounded by the states at hand and by the HBM the K node pools need; tie-heavy integer
    built-ins stay single-instance."""
    from types import SimpleNamespace as NS
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods import astar
    from deepcubea_amd.search_methods.engine import BwasEngine
    from deepcubea_amd.utils import env_utils
    env = env_utils.get_environment("cube3")
    hbm = [288 << 30]  # an empty MI355X; auto_max_nodes itself needs the device (torch.cuda.mem_get_info)
    monkeypatch.setattr(BwasEngine, "auto_max_nodes", staticmethod(
        lambda env_name, batch_size, num_instances=1, fraction=0.8, sharers=1:
        min(int(hbm[0] // sharers * fraction) // (BwasEngine.bytes_per_node(env_name) * num_instances), 0x7FFFFF00 // 2)))
    mk = lambda **kw: NS(env="cube3", max_nodes=str(1 << 20), instances_per_gpu="auto", **kw)
    assert astar.auto_instances(mk(batch_size=20000), env, 1000, None) == 1          # 240 000 rows per call already
    assert astar.auto_instances(mk(batch_size=10000), env, 1000, None) == 2          # train.sh's batch: two fill a GEMM
    assert astar.auto_instances(mk(batch_size=20000), env, 1000, _lib.HEUR_HASHU01) == 16
    assert astar.auto_instances(mk(batch_size=20000), env, 3, _lib.HEUR_HASHU01) == 3  # never more than the states at hand
    assert astar.auto_instances(mk(batch_size=100), env, 1000, _lib.HEUR_HASHU01) == 16
    assert astar.auto_instances(mk(batch_size=10000), env_utils.get_environment("puzzle15"), 500, _lib.HEUR_MANHATTAN) == 1
    a = mk(batch_size=20000)
    a.instances_per_gpu = "5"
    assert astar.auto_instances(a, env, 1000, None) == 5
    # an explicit --max_nodes is ids PER SEARCH and every instance owns a pool — `auto` must not turn a command
    # that fitted at K = 1 into K pools that do not fit.  8e8 ids x ~200 B = 160 GB: one pool fits 288 GB, two do not
    big = mk(batch_size=10000)
    big.max_nodes = "800000000"
    assert astar.auto_instances(big, env, 1000, None) == 1
    assert astar.auto_instances(mk(batch_size=20000, ), env, 1000, _lib.HEUR_HASHU01) == 16  # 2^20 ids x 16: no constraint
    mid = mk(batch_size=20000)
    mid.max_nodes = str(1 << 27)   # 16 pools of 2^27 ids = 430 GB: the largest K whose pools fit 80 % of the HBM
    k = astar.auto_instances(mid, env, 1000, _lib.HEUR_HASHU01)
    assert 1 < k < 16 and BwasEngine.auto_max_nodes("cube3", 20000, k) >= (1 << 27) > BwasEngine.auto_max_nodes("cube3", 20000, k + 1)
    hbm[0] = 36 << 30                # an eighth of the device (eight ranks sharing it): --max_nodes auto keeps pools >= 2^27 ids
    auto = mk(batch_size=20000)
    auto.max_nodes = "auto"
    assert astar.auto_instances(auto, env, 1000, _lib.HEUR_HASHU01) == 1


def test_graft_entry_build_passes_on_this_box():
    """`__graft_entry__.build()` is the driver's "does it build" check (hipcc cross-compiles without a GPU): it must succeed here —
    including its own assertion on the library's ABI version, which round 5 bumped without touching that file until the GPU box's
    smoke test said so."""
    import importlib
    ge = importlib.import_module("__graft_entry__")
    ge.build()
    from deepcubea_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dca.h")).read()
    assert "#define DCA_ABI_VERSION %d" % _lib.lib().dca_abi_version() in hdr


def test_ranks_sharing_a_gpu_are_not_guessed_from_world_size(monkeypatch):
    """launchers that set RANK / WORLD_SIZE / LOCAL_RANK but no LOCAL_WORLD_SIZE (srun, mpirun) made every rank of a
    multi-node job believe WORLD_SIZE / devices ranks share its GPU — node pools cut to 1/nodes of the HBM, grid-wide tie
    refinement off.  Without LOCAL_WORLD_SIZE (and without a process group to count over) the deployment model is assumed: one
    rank per GPU; with it, the local ranks are mapped onto the visible devices; a count taken over the process group wins."""
    import torch
    from deepcubea_amd.search_methods import sharding
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sharding, "_sharers", None)
    monkeypatch.setenv("WORLD_SIZE", "32")   # four nodes of eight
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert sharding.ranks_on_my_device() == 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert sharding.ranks_on_my_device() == 1
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)  # a test box: eight local ranks on one GPU
    monkeypatch.setenv("LOCAL_RANK", "0")
    assert sharding.ranks_on_my_device() == 8
    monkeypatch.setattr(sharding, "_sharers", 2)  # counted over the process group: (host, device) pairs equal to mine
    assert sharding.ranks_on_my_device() == 2


def test_sharers_are_counted_by_physical_device_not_by_index(monkeypatch):
    """under srun on AMD every task sees ITS GPU as device 0 (ROCR_VISIBLE_DEVICES binding, HIP_ / CUDA_VISIBLE_DEVICES
    unset): an (index, masks) key made all ranks of a node compare equal.  The key is the device's UUID (or PCI address)."""
    from types import SimpleNamespace as NS
    import torch
    from deepcubea_amd.search_methods import sharding
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: NS(uuid="GPU-aaaa", pci_bus_id=5))
    a = sharding._device_identity()
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: NS(uuid="GPU-bbbb", pci_bus_id=6))
    b = sharding._device_identity()
    assert a != b and a[1] == "uuid"            # two tasks, both "device 0", two GPUs
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: NS(pci_domain_id=0, pci_bus_id=5, pci_device_id=0))
    c = sharding._device_identity()
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: NS(pci_domain_id=0, pci_bus_id=6, pci_device_id=0))
    assert c != sharding._device_identity() and c[1] == "pci"
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: NS())   # a runtime that reports neither
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "3")
    d = sharding._device_identity()
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4")
    assert d != sharding._device_identity() and d[1] == "index"


def test_bench_reads_the_committed_pmc_traffic_of_this_round():
    """bench.py's `roofline.traffic` figures come from the rocprofv3 PMC passes committed under profiles/ (counters cannot be read
    inside the timed process): the engine kernels' and the gather kernel's entries of THIS round (6) are there and name their source."""
    import importlib
    bench = importlib.import_module("bench")
    t = bench.pmc_traffic("k_expand", "cube3", 20000)
    assert t is not None and 3e7 < t < 1.2e8 and bench.PMC_SOURCE["k_expand"].startswith("profiles/r06_pmc_traffic.json")
    bench.expand_pmc_traffic("f32", 1_000_000)
    assert bench.PMC_SOURCE["expand_fused_kernel<cube3,f32>"].startswith("profiles/r06_expand_pmc_traffic.json")
    for oh, alg in (("f32", 16_254_000_000), ("bf16", 8_478_000_000)):
        b = bench.expand_pmc_traffic(oh, 1_000_000)
        assert b is not None and alg <= b <= 1.03 * alg, (oh, b)  # no re-reads: measured traffic = algorithmic bytes (+ hash / solved)
    assert bench.expand_pmc_traffic("f32", 12345) is None
