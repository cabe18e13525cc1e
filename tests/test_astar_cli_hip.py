"""GPU end-to-end tests of the `--language hip` driver and of the heuristic service on PyTorch-ROCm."""
import os
import pickle
import re
import sys
import types

import numpy as np

from deepcubea_amd.utils import data_utils
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port() -> str:
    """A port nobody listens on right now (fixed ports collide when an earlier rendezvous still lingers)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def _ref_pickle(path, arrays):
    """A states pickle with the REFERENCE's class path (environments.cube3.Cube3State, int64 colors)."""
    pkg, mod = types.ModuleType("environments"), types.ModuleType("environments.cube3")

    class Cube3State:  # noqa
        __slots__ = ['colors', 'hash']

        def __init__(self, colors):
            self.colors = colors
            self.hash = None
    Cube3State.__module__, Cube3State.__qualname__ = "environments.cube3", "Cube3State"
    mod.Cube3State = Cube3State
    sys.modules["environments"], sys.modules["environments.cube3"] = pkg, mod
    try:
        blob = pickle.dumps({"states": [Cube3State(a.astype(np.int64)) for a in arrays]}, protocol=2)
    finally:
        del sys.modules["environments"], sys.modules["environments.cube3"]
    open(path, "wb").write(blob)


def test_heuristic_forward_matches_reference_within_1e5(golden, tiny_resnet):
    """north star: heuristic values within 1e-5 of the reference (fp32 both sides)."""
    from deepcubea_amd.utils.pytorch_models import ResnetModel, fold_batchnorm
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    from deepcubea_amd.utils import nnet_utils
    from deepcubea_amd import _lib
    m = ResnetModel(54, 6, 64, 32, 2, 1, True)
    m.load_state_dict({k[2:]: torch.tensor(tiny_resnet[k]) for k in tiny_resnet.files if k.startswith("w:")})
    m = m.cuda().eval()
    x = torch.tensor(tiny_resnet["x"]).cuda()
    hfn = nnet_utils.get_heuristic_fn_dev(m, batch_size=100)
    y = hfn(x).cpu().numpy()
    assert np.max(np.abs(y - tiny_resnet["y"])) < 1e-5
    yo = hfn(_lib.onehot(x, 6, torch.float32), True).cpu().numpy()
    assert np.array_equal(yo, y)
    full = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(full, 2024)
    full = full.cuda().eval()
    ref = golden["cube3_resnet_seed2024_y"]
    with torch.no_grad():
        yy = full(torch.tensor(golden["cube3_resnet_seed2024_x"]).cuda())[:, 0].cpu().numpy()
        yf = fold_batchnorm(full)(torch.tensor(golden["cube3_resnet_seed2024_x"]).cuda())[:, 0].cpu().numpy()
    tol = 1e-5 * max(1.0, float(np.abs(ref).max()))
    assert np.max(np.abs(yy - ref)) < tol and np.max(np.abs(yf - ref)) < tol
    # padded / epilogue-fused inference layout (the CLI default): same function, fp32
    from deepcubea_amd.utils.pytorch_models import FastResnet
    fast = FastResnet(full).cuda()
    ys = fast(torch.tensor(golden["cube3_resnet_seed2024_x"]).cuda())[:, 0].cpu().numpy()
    assert np.max(np.abs(ys - ref)) < tol
    yt = FastResnet(m).cuda()(x)[:, 0].cpu().numpy()
    assert np.max(np.abs(yt - tiny_resnet["y"])) < 1e-5
    # reference-signature closure (lists of State objects -> float64)
    from deepcubea_amd.utils import env_utils
    env = env_utils.get_environment("cube3")
    states = env.np_to_states(golden["cube3_synth64_in"][:16])
    h = nnet_utils.get_heuristic_fn(full, torch.device("cuda"), env, clip_zero=True, batch_size=7)(states)
    assert h.dtype == np.float64 and h.shape == (16,) and (h >= 0).all()


@pytest.mark.parametrize("mode", ["dedup_first", "eval_all_children"])
def test_cli_end_to_end(tmp_path, capsys, mode):
    from deepcubea_amd.search_methods import astar
    from deepcubea_amd.utils import env_utils, nnet_utils
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    from oracle import c_oracle as co
    scr = [[0, 5, 7], [1, 3, 8, 10], [], [4, 9]]
    roots = []
    for mv in scr:
        s = np.arange(54, dtype=np.uint8)[None]
        for a in mv:
            s = co.next_state("cube3", s, a)
        roots.append(s[0])
    spath = str(tmp_path / "states.pkl")
    _ref_pickle(spath, roots)
    rdir = str(tmp_path / "res")
    B, w = 60, 0.8
    astar.main(["--states", spath, "--model", "synthetic:11", "--env", "cube3", "--weight", str(w), "--batch_size",
                str(B), "--results_dir", rdir, "--language", "hip", "--nnet_batch_size", "1024", "--max_nodes",
                str(1 << 20)] + (["--eval_all_children"] if mode == "eval_all_children" else []))
    sys.stdout = sys.__stdout__
    res = data_utils.load_pickle(os.path.join(rdir, "results.pkl"))
    assert sorted(res.keys()) == ["num_nodes_generated", "paths", "solutions", "states", "times"]  # astar.py:382-397
    log = open(os.path.join(rdir, "output.txt")).read()
    lines = re.findall(r"State: (\d+), SolnCost: ([\d.]+), # Moves: (\d+), # Nodes Gen: ([\d,]+), Time: ([\d.]+)", log)
    assert len(lines) == 4  # astar.py:449-452 line format
    env = env_utils.get_environment("cube3")
    # same search with the CPU oracle driven by the SAME device heuristic (padded to the engine's row count)
    nnet = env.get_nnet_model()
    load_synthetic_weights(nnet, 11)
    nnet = nnet.cuda().eval()
    if mode == "dedup_first":  # the CLI's default network layout; batches are rounded up to 1024 rows
        from deepcubea_amd.utils.pytorch_models import FastResnet
        hfn = nnet_utils.get_heuristic_fn_dev(FastResnet(nnet).cuda(), batch_size=1024)
        M = 1024
    else:
        hfn = nnet_utils.get_heuristic_fn_dev(nnet, batch_size=1024)
        M = B * 12

    def heur(states):
        x = torch.zeros((max(M, len(states)), 54), dtype=torch.uint8, device="cuda")
        x[:len(states)] = torch.from_numpy(np.ascontiguousarray(states // 9)).cuda()
        return hfn(x)[:len(states)].cpu().numpy()

    for i, root in enumerate(roots):
        soln = res["solutions"][i]
        s = root[None].copy()
        for a in soln:
            s = co.next_state("cube3", s, a)
        assert co.is_solved("cube3", s)[0]
        assert len(res["paths"][i]) == len(soln) + 1
        assert int(lines[i][2]) == len(soln) and int(lines[i][3].replace(",", "")) == res["num_nodes_generated"][i]
        ref = co.astar("cube3", root, w, B, co.SEM_PY, heur_fn=heur)
        assert len(soln) == len(ref["moves"])
        assert res["num_nodes_generated"][i] == ref["nodes_generated"], (i, res["num_nodes_generated"][i], ref)


@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp8", "fp8mx"])
def test_cli_non_parity_dtypes_produce_valid_solutions(tmp_path, dt):
    """`--nnet_dtype bf16 | fp16 | fp8 | fp8mx` through the CLI (argument parsing, network construction, engine wiring): the
    non-parity modes promise valid solutions in the reference's result format, nothing about node counts."""
    from deepcubea_amd.search_methods import astar
    from oracle import c_oracle as co
    scr = [[0, 5, 7], [1, 3, 8, 10], [4, 9, 2, 6]]
    roots = []
    for mv in scr:
        s = np.arange(54, dtype=np.uint8)[None]
        for a in mv:
            s = co.next_state("cube3", s, a)
        roots.append(s[0])
    spath = str(tmp_path / "states.pkl")
    _ref_pickle(spath, roots)
    rdir = str(tmp_path / "res")
    astar.main(["--states", spath, "--model", "synthetic:11", "--env", "cube3", "--weight", "0.8", "--batch_size", "200",
                "--results_dir", rdir, "--language", "hip", "--nnet_batch_size", "4096", "--max_nodes", str(1 << 21),
                "--nnet_dtype", dt])
    sys.stdout = sys.__stdout__
    res = data_utils.load_pickle(os.path.join(rdir, "results.pkl"))
    assert len(res["solutions"]) == len(roots)
    for root, soln, path in zip(roots, res["solutions"], res["paths"]):
        s = root[None].copy()
        for a in soln:
            s = co.next_state("cube3", s, a)
        assert co.is_solved("cube3", s)[0] and len(path) == len(soln) + 1


def test_nnet_bf16_mode_still_solves(tmp_path):
    """bf16 heuristic = explicitly non-parity mode: only validity of the solution is asserted."""
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods.engine import BwasEngine
    from deepcubea_amd.utils import nnet_utils
    from deepcubea_amd.utils.pytorch_models import ResnetModel, fold_batchnorm
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    from oracle import c_oracle as co
    m = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(m, 3)
    m = fold_batchnorm(m).cuda().eval()
    hfn = nnet_utils.get_heuristic_fn_dev(m, autocast_dtype=torch.bfloat16)
    s = np.arange(54, dtype=np.uint8)[None]
    for a in [2, 7, 9, 4]:
        s = co.next_state("cube3", s, a)
    eng = BwasEngine("cube3", 0.8, 200, max_nodes=1 << 21, onehot_dtype=torch.bfloat16)
    res = eng.solve(s[0], hfn, max_iters=3000)
    assert res["solved"]
    t = s.copy()
    for a in res["moves"]:
        t = co.next_state("cube3", t, a)
    assert co.is_solved("cube3", t)[0]
    eng.close()


@pytest.mark.parametrize("scaling", ["tensor", "block"])
def test_nnet_fp8_mode_still_solves(tmp_path, scaling):
    """fp8 heuristic (Fp8Resnet: e4m3 layers, per-tensor scales = `--nnet_dtype fp8`, block scales = `fp8mx`) = explicitly
    non-parity mode: only validity of the solution is asserted; the mode must actually have switched to the e4m3 kernels on
    the way."""
    from deepcubea_amd.search_methods.engine import BwasEngine
    from deepcubea_amd.utils import nnet_utils
    from deepcubea_amd.utils.pytorch_models import Fp8Resnet, ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    from oracle import c_oracle as co
    m = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(m, 3)
    fast = Fp8Resnet(m.eval(), scaling=scaling).cuda()
    hfn = nnet_utils.get_heuristic_fn_dev(fast)
    s = np.arange(54, dtype=np.uint8)[None]
    for a in [2, 7, 9, 4]:
        s = co.next_state("cube3", s, a)
    eng = BwasEngine("cube3", 0.8, 200, max_nodes=1 << 21, packed=True)
    res = eng.solve(s[0], hfn, max_iters=3000)
    assert res["solved"]
    assert fast.scaling == scaling
    # per-tensor: calibrated on the first batch of >= 1024 real rows; block-scaled (dca_gemm8_mx): nothing to calibrate
    assert (fast.layer_scale is not None) == (scaling == "tensor")
    t = s.copy()
    for a in res["moves"]:
        t = co.next_state("cube3", t, a)
    assert co.is_solved("cube3", t)[0]
    eng.close()


def test_cli_instances_per_gpu(tmp_path):
    """--instances_per_gpu K: K scrambles stepped by one engine, one network call per iteration for all of them."""
    from deepcubea_amd.search_methods import astar
    from oracle import c_oracle as co
    scr = [[0, 5, 7], [1, 3, 8, 10], [], [4, 9], [2, 6, 11]]
    roots = []
    for mv in scr:
        s = np.arange(54, dtype=np.uint8)[None]
        for a in mv:
            s = co.next_state("cube3", s, a)
        roots.append(s[0])
    spath = str(tmp_path / "states.pkl")
    _ref_pickle(spath, roots)
    outs = {}
    for k in (1, 3, "auto"):
        rdir = str(tmp_path / ("res%s" % k))
        astar.main(["--states", spath, "--model", "synthetic:11", "--env", "cube3", "--weight", "0.8", "--batch_size",
                    "60", "--results_dir", rdir, "--nnet_batch_size", "1000", "--max_nodes", str(1 << 20),
                    "--instances_per_gpu", str(k), "--debug"])
        outs[k] = data_utils.load_pickle(os.path.join(rdir, "results.pkl"))
    for i, root in enumerate(roots):
        for k in (1, 3, "auto"):
            s = root[None].copy()
            for a in outs[k]["solutions"][i]:
                s = co.next_state("cube3", s, a)
            assert co.is_solved("cube3", s)[0]
        assert len(outs[1]["solutions"][i]) == len(outs[3]["solutions"][i]) == len(scr[i])
    # results.pkl does not depend on K: same moves, same node counts, same paths — only the per-state
    # wall times differ.  `auto` (the default) steps all five together here: 60 x 12 children per instance fill nothing.
    for k in (3, "auto"):
        assert outs[k]["solutions"] == outs[1]["solutions"]
        assert outs[k]["num_nodes_generated"] == outs[1]["num_nodes_generated"]
        assert [len(p) for p in outs[k]["paths"]] == [len(p) for p in outs[1]["paths"]]


def test_cli_two_ranks_sharded(tmp_path):
    """N>1 path end to end on the GPU box: two ranks under torch.distributed.run (sharing the one GPU here; one GPU
    per rank on a multi-GPU node), scrambles drawn from the shared work queue, rank 0 writes the merged results.pkl."""
    import subprocess
    from oracle import c_oracle as co
    scr = [[0, 5, 7], [1, 3, 8, 10], [], [4, 9], [2, 6, 11]]
    roots = []
    for mv in scr:
        s = np.arange(54, dtype=np.uint8)[None]
        for a in mv:
            s = co.next_state("cube3", s, a)
        roots.append(s[0])
    spath = str(tmp_path / "states.pkl")
    _ref_pickle(spath, roots)
    rdir = str(tmp_path / "res2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", _free_port(), "-m", "deepcubea_amd.search_methods.astar", "--states", spath,
           "--model_dir", "synthetic:11", "--env", "cube3", "--weight", "0.8", "--batch_size", "60", "--results_dir", rdir,
           "--nnet_batch_size", "1000", "--max_nodes", str(1 << 20)]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        # Two fresh processes bringing up HIP contexts and a gloo rendezvous next to this one on the same GPU: one failure in
        # ~10 suite runs was seen on the pool with nothing of the ranks' own in the launcher's summary.  Show it, then try
        # ONCE more on another port — a second failure is a failure.
        print("first attempt failed (rc %d):\n%s" % (out.returncode, out.stderr[-12000:]))
        cmd[cmd.index("--master-port") + 1] = _free_port()
        out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-12000:]  # (the ranks' own tracebacks come before the launcher's summary)
    res = data_utils.load_pickle(os.path.join(rdir, "results.pkl"))
    assert len(res["solutions"]) == 5 and len(res["times"]) == 5 and len(res["num_nodes_generated"]) == 5
    for i, r0 in enumerate(roots):
        s = r0[None].copy()
        for a in res["solutions"][i]:
            s = co.next_state("cube3", s, a)
        assert co.is_solved("cube3", s)[0] and len(res["solutions"][i]) == len(scr[i])
    assert os.path.isfile(os.path.join(rdir, "output.txt"))  # rank 0's log (the states it drew from the shared queue)


def test_bench_two_ranks_real_engine_over_gloo():
    """The N > 1 path of bench.py with the REAL engine (tests/test_bench_cpu.py only drives `selftest`):
    `bench.py --gpus 2` re-executes itself under torch.distributed.run, two ranks (sharing GPU 0 here, one GPU each on a
    multi-GPU node), every rank its own search replica on its own scrambles, the timing barrier / max / sum over gloo, and
    the sharded leg drawing 2 x 2 puzzle15 scrambles from the shared work queue.  Same launch pattern as the reference's
    per-GPU fan-out (nnet_utils.py:292-301)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "5", "--warmup",
           "2", "--nnet-steps", "0", "--no-cpu-baseline", "--concurrent", "0", "--queue-states", "2", "--no-expand-block"]
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-8000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["scaling"] == "weak" and j["unit"] == "nodes expanded/s"
    assert len(j["per_rank_value"]) == 2 and all(v > 0 for v in j["per_rank_value"])
    # value = nodes expanded by BOTH ranks / max-over-ranks time: no less than either rank alone could report
    assert j["value"] >= 0.9 * max(j["per_rank_value"]) and j["value"] <= 1.1 * sum(j["per_rank_value"])
    assert j["config"]["parallelism"].endswith("x2")
    q = j["sharded_queue"]
    assert q["states"] == 4 and len(q["states_per_rank"]) == 2 and sum(q["states_per_rank"]) == 4
    assert "roofline" in j and "engine_onehot_f32" in j
