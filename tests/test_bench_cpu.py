"""CPU tests of bench.py's launcher plumbing (no GPU work): `--gpus N` must really start N ranks — directly (bench.py
re-executes itself under torch.distributed.run) or when the driver launches it under torch.distributed.run — and rank 0
prints ONE JSON line with n_gpus = N (`--gpus` was parsed and never used)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _free_port() -> str:
    """A port nobody listens on right now (fixed ports collide when an earlier rendezvous still lingers)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])



def _json_line(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_flag_spawns_one_rank_per_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "selftest", "--gpus", "2",
                        "--dist-backend", "gloo", "--steps", "3", "--warmup", "0"], capture_output=True, text=True,
                       timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 0
    assert line["per_rank_value"] == [0.0, 1.0]  # both ranks took part
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["vs_baseline"] is None


def test_driver_style_launch_and_rank_count_check():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--workload", "selftest",
           "--dist-backend", "gloo"]
    r = subprocess.run(cmd + ["--gpus", "2"], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 2
    # --gpus must match the ranks the launcher started
    r = subprocess.run(cmd + ["--gpus", "4"], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode != 0 and "--gpus 4" in (r.stderr + r.stdout)


def test_single_process_default():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "selftest"], capture_output=True,
                       text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_engine_bytes_match_survey_8d():
    sys.path.insert(0, ROOT)
    import bench
    b = bench.engine_bytes("cube3", 20000, 0)
    assert b["per_expansion_8d"] == 702 + 12 * 64          # children only + engine bookkeeping
    assert bench.engine_bytes("cube3", 20000, 4)["per_expansion_8d"] == 16254 + 12 * 64  # fp32 one-hot
    assert bench.engine_bytes("puzzle48", 20000, 4)["per_expansion_8d"] == 38661 + 4 * 64
