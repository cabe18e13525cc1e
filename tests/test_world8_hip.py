"""World-8 dry runs on ONE GPU.  No 8-GPU node has been available to any round, so no scaling curve can be
measured — but everything an 8-rank run does except owning eight devices can be exercised: the rendezvous of eight ranks, the
count of ranks sharing a device (by physical identity), node pools cut to 1/8 of the HBM, the timing barrier / max / sum over
eight ranks, the shared work queue's draws summing to the state count, the merge order of `results.pkl` — and the reference's
pinning pattern (a worker writes its visibility mask before its first device call and addresses `cuda:0`,
utils/nnet_utils.py:208-209, 292-301)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from deepcubea_amd.utils import data_utils

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _free_port() -> str:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def _ref_pickle(path, arrays):
    """A states pickle with the REFERENCE's class path (environments.cube3.Cube3State, int64 colors)."""
    import pickle
    import types
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


def test_bench_eight_ranks_share_one_gpu():
    """`bench.py --gpus 8 --dist-backend gloo`: eight ranks under torch.distributed.run on GPU 0, the real engine in each."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--steps", "5", "--warmup", "2",
           "--nnet-steps", "0", "--no-cpu-baseline", "--concurrent", "0", "--queue-states", "4", "--no-expand-block"]
    out = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-12000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 5 and j["scaling"] == "weak" and j["unit"] == "nodes expanded/s"
    assert len(j["per_rank_value"]) == 8 and all(v > 0 for v in j["per_rank_value"])
    # value = nodes expanded by ALL ranks / max-over-ranks time
    assert 0.9 * max(j["per_rank_value"]) <= j["value"] <= 1.1 * sum(j["per_rank_value"])
    assert j["config"]["parallelism"].endswith("x8")
    # every rank found the other seven on its device (counted over the process group by device UUID / PCI address)
    assert j["ranks_sharing_gpu"] == [8] * 8
    q = j["sharded_queue"]
    assert q["states"] == 32 and len(q["states_per_rank"]) == 8 and sum(q["states_per_rank"]) == 32  # every draw exactly once
    assert "roofline" in j and "engine_onehot_f32" in j


def test_cli_eight_ranks_forty_scrambles(tmp_path):
    """The CLI under `torch.distributed.run --nproc-per-node 8` on 40 shallow cube3 scrambles: eight ranks on GPU 0, the
    network replicated eight times, pools at 1/8 of the device, states drawn from the shared queue, merged in state order."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(8)
    scr = [[int(a) for a in rng.integers(0, 12, size=int(rng.integers(0, 5)))] for _ in range(40)]
    roots = []
    for mv in scr:
        s = np.arange(54, dtype=np.uint8)[None]
        for a in mv:
            s = co.next_state("cube3", s, a)
        roots.append(s[0])
    spath = str(tmp_path / "states.pkl")
    _ref_pickle(spath, roots)
    rdir = str(tmp_path / "res8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), "-m", "deepcubea_amd.search_methods.astar", "--states", spath, "--model_dir",
           "synthetic:11", "--env", "cube3", "--weight", "0.8", "--batch_size", "60", "--results_dir", rdir,
           "--nnet_batch_size", "1000", "--instances_per_gpu", "1", "--debug"]  # --max_nodes auto: the 1/8 share
    out = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1500)
    if out.returncode != 0:  # (see test_cli_two_ranks_sharded: one retry on another port, a second failure is a failure)
        print("first attempt failed (rc %d):\n%s" % (out.returncode, out.stderr[-12000:]))
        cmd[cmd.index("--master-port") + 1] = _free_port()
        out = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-12000:]
    res = data_utils.load_pickle(os.path.join(rdir, "results.pkl"))
    assert len(res["solutions"]) == 40 and len(res["times"]) == 40 and len(res["num_nodes_generated"]) == 40
    for i, r0 in enumerate(roots):  # merged in STATE order whichever rank solved which
        s = r0[None].copy()
        for a in res["solutions"][i]:
            s = co.next_state("cube3", s, a)
        assert co.is_solved("cube3", s)[0] and len(res["solutions"][i]) <= len(scr[i])
        assert np.array_equal(np.asarray(res["states"][i].colors, dtype=np.uint8), r0)
    # --debug: every rank logs to stdout; the queue handed out each of the 40 indices exactly once
    drawn = sorted(int(m) for m in re.findall(r"^State: (\d+),", out.stdout, flags=re.M))
    assert drawn == list(range(40)), drawn
    assert "8 ranks share this GPU" in out.stdout and "1/8 of its memory" in out.stdout


def test_worker_pinned_by_visibility_mask_addresses_cuda0(tmp_path):
    """The reference's pinning pattern: a spawned worker writes `<X>_VISIBLE_DEVICES=<k>` before its first device call and then
    uses `cuda:0` (nnet_utils.py:208-209).  For every GPU k of this box (one here, eight on a node) and both spellings of the
    mask: the worker sees ONE device, as cuda:0, it is physically device k of the unmasked view, and a search on it equals the
    oracle.  A mask with no device behind it makes the product refuse loudly (no fallback)."""
    from deepcubea_amd.search_methods import sharding
    ndev = torch.cuda.device_count()
    assert ndev >= 1
    worker = os.path.join(ROOT, "tests", "_pinned_gpu_worker.py")
    # this process may itself run behind a mask (the box's launcher): logical device k is then the k-th entry of that mask,
    # and that entry is what a worker has to write to reach the same physical GPU
    outer = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES") or ""
    outer = [v.strip() for v in outer.split(",") if v.strip()]
    for k in range(min(ndev, 8)):
        torch.cuda.set_device(k)
        want = [str(x) for x in sharding._device_identity()[1:]]
        gpu_num = outer[k] if k < len(outer) else str(k)
        for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            outp = str(tmp_path / ("pin_%s_%d.json" % (var, k)))
            env = _env()
            env.pop("HIP_VISIBLE_DEVICES", None)
            env.pop("CUDA_VISIBLE_DEVICES", None)
            r = subprocess.run([sys.executable, worker, outp, var, gpu_num], cwd=ROOT, env=env, capture_output=True, text=True,
                               timeout=600)
            assert r.returncode == 0, r.stderr[-6000:]
            rec = json.load(open(outp))
            assert rec["device_count"] == 1 and rec["current_device"] == 0 and rec["alloc_device"] == 0, rec
            assert rec["solved"], rec
            if want[0] in ("uuid", "pci"):  # a physical identity: the masked worker's cuda:0 IS device k
                assert rec["identity"] == want, (rec["identity"], want)
    torch.cuda.set_device(0)
    outp = str(tmp_path / "pin_none.json")
    r = subprocess.run([sys.executable, worker, outp, "HIP_VISIBLE_DEVICES", "-1"], cwd=ROOT, env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-6000:]
    rec = json.load(open(outp))
    assert rec["device_count"] == 0 and rec["require_gpu"] == "DcaError", rec
