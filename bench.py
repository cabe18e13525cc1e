#!/usr/bin/env python3
"""bench.py — throughput of the MI355X-native BWAS hot path (driver contract in the task statement).

    python bench.py --gpus N --steps K --warmup W            (N>1: launched under torch.distributed.run)

Workloads:
  astar   (default)  BASELINE.json configs[2] shape: cube3 weighted A*, w=0.8, batch 20 000, one search
                     instance per GPU, device-resident engine.  One step = one full BWAS iteration
                     (pop 20 000 -> expand 240 000 children + is_solved + hash -> heuristic -> cost ->
                     CLOSED dedup -> push).  `value` uses the built-in hash-derived heuristic
                     (SURVEY §8d "engine-only"); the same line carries `end_to_end_nnet` = the same loop with
                     the 14.7M-parameter ResNet heuristic on PyTorch-ROCm (synthetic weights).
  expand             BASELINE.json configs[1]: fused next_state + one-hot kernel on 1M synthetic cube3
                     states (one step = one launch over the 1M parents).

Multi-GPU: the path shards per search instance — every rank runs its own replica on its own
scrambles, no collective on the data path; only the timing barrier uses RCCL.  scaling = weak.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (≈6.3 TB/s achievable)
# SURVEY §8d algorithmic bytes per cube3 expansion with the fp32 fused one-hot:
#   54 (parent read) + 12*54 (children u8) + 12*324*4 (one-hot f32) = 16 254 B
CUBE3_EXPAND_BYTES_F32 = 54 + 12 * 54 + 12 * 324 * 4


def synth_states(n: int, d: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.permuted(np.tile(np.arange(d, dtype=np.uint8), (n, 1)), axis=1)


def dist_setup(n_gpus: int):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    return world, rank, local


def barrier(world: int):
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x: float, world: int) -> float:
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, world: int) -> float:
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


# --------------------------------------------------------------------------------------------------
# workload: expand (configs[1])
# --------------------------------------------------------------------------------------------------
def run_expand(args, world, rank):
    from deepcubea_amd import _lib
    n = args.n
    S = torch.from_numpy(synth_states(n, 54, rank)).cuda()
    out = {
        "children": torch.empty((n, 12, 54), dtype=torch.uint8, device="cuda"),
        "onehot": torch.empty((n * 12, 324), dtype=torch.float32, device="cuda"),
        "solved": torch.empty((n * 12,), dtype=torch.uint8, device="cuda"),
        "hash": torch.empty((n * 12,), dtype=torch.int64, device="cuda"),
    }
    e, d = _lib.ENV_CUBE3, 0

    def step():
        _lib.expand_fused(e, d, S, out=out)

    for _ in range(args.warmup):
        step()
    barrier(world)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    barrier(world)
    wall = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    wall = max_over_ranks(wall, world)
    total_exp = sum_over_ranks(float(n * args.steps), world)
    achieved = CUBE3_EXPAND_BYTES_F32 * n / (kern_ms * 1e-3) / 1e9
    res = {
        "value": total_exp / wall,
        "ms_per_step": wall / args.steps * 1e3,
        "config": {"workload": "cube3 fused next_state+one-hot(f32)+is_solved+hash kernel, %d synthetic states "
                               "(BASELINE configs[1])" % n, "states": n, "moves": 12, "onehot": "f32",
                   "parallelism": "replica-per-gpu x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "expand_fused_kernel<cube3,f32>", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "bytes_per_launch": CUBE3_EXPAND_BYTES_F32 * n, "kernel_ms": kern_ms},
    }
    return res


def cpu_baseline_expand(seconds_budget: float = 12.0):
    """Reference cpp/environments.cpp (oracle/_ref, kind=reference) driven like the OpenMP expand loop of
    cpp/parallel_weighted_astar.cpp:217-230, on a bounded sample of the same synthetic states."""
    from oracle import c_oracle as co
    kind = "reference" if co.ref_lib() is not None else "port"
    fn = co.ref_expand if kind == "reference" else (lambda env, s: co.expand(env, s)[:2])
    cores = co.num_threads()
    n = 50_000
    S = synth_states(n, 54, 0)
    fn("cube3", S[:1000])
    t0 = time.perf_counter()
    fn("cube3", S)
    dt = time.perf_counter() - t0
    reps = max(1, int(seconds_budget / max(dt, 1e-3)))
    reps = min(reps, 40)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn("cube3", S)
    dt = (time.perf_counter() - t0) / reps
    return {"value": n / dt, "unit": "nodes expanded/s", "cores": cores, "kind": kind,
            "sample": "%d x expand of %d synthetic cube3 states (children only, no one-hot), "
                      "OpenMP over parents like cpp/parallel_weighted_astar.cpp:217-230" % (reps, n)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="expand", choices=["expand"])
    ap.add_argument("--n", type=int, default=1_000_000, help="expand: synthetic states per launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20
    if args.warmup is None:
        args.warmup = 3
    world, rank, local = dist_setup(args.gpus)
    res = run_expand(args, world, rank)
    line = {
        "metric": "A* nodes expanded/sec on cube3",
        "value": res["value"],
        "unit": "nodes expanded/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": res["config"],
        "roofline": res["roofline"],
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_expand()
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
