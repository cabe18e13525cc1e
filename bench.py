#!/usr/bin/env python3
"""bench.py — throughput of the MI355X-native BWAS hot path (driver contract in the task statement).

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the driver) or
directly — bench.py then re-executes itself under torch.distributed.run, one rank per GPU (rank r pinned to GPU r, the
process group bound to that device).  `--gpus` must equal the number of ranks.

Workloads:
  astar   (default)  BASELINE.json configs[2] geometry — the config the metric "A* nodes expanded/sec on cube3, batch
                     20k" is quoted on: cube3 weighted A*, w=0.8, batch 20 000, one search instance per GPU on the
                     device-resident engine.  One step = one full BWAS iteration: pop 20 000 by (cost, push order) ->
                     expand 240 000 children (+is_solved, hash, node fields) -> heuristic -> cost -> CLOSED dedup ->
                     push.  `value` is measured with the built-in hash-derived heuristic 10+5*u01(hash) (SURVEY §8d
                     "engine-only") over episodes of W untimed + K timed iterations on fresh test-set scrambles, repeated
                     until >= 0.25 s were timed.  The same JSON line carries: `roofline` (dominant launch + every launch
                     against the same roofline, timed by device-side stamps inside the replayed hipGraph over one more
                     episode of the timed shape), `roofline_iteration` (SURVEY §8(d) bytes x batch / ms_per_step),
                     `engine_onehot_f32` (the same iteration with the fp32 one-hot rows fused into the expansion launch),
                     `expand_1M` (BASELINE configs[1]: the gather kernel on 1M states, fp32 and bf16 one-hot rows, each
                     with its HBM roofline), `end_to_end_nnet` (the 14.7M-parameter ResNet heuristic in the loop: fp32 parity mode, bf16 on the
                     library / on the hand-written layer kernel, reference order; synthetic weights — the reference's
                     checkpoints are not in the mount), `concurrent_instances`, `sharded_queue` (configs[3]'s path: 32
                     shipped puzzle15 scrambles per rank drawn from the shared work queue and searched to completion) and
                     `cpu_baseline` (the C++/OpenMP port + the reference's own compiled environments.cpp on the expansion).
  expand             BASELINE.json configs[1]: fused next_state + one-hot(f32) + is_solved + hash kernel on
                     1M synthetic cube3 states (one step = one launch over the 1M parents).
  avi / train        SURVEY §8(f) rows: AVI update step (configs[4]) and the training step.

Multi-GPU: the path shards per search instance — every rank runs its own replica on its own
scrambles, no collective on the data path; only the timing barrier / reductions use RCCL.  scaling = weak.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (≈6.3 TB/s achievable)
MIN_TIMED_S = 0.25     # the astar legs repeat their K-step region on fresh scrambles until this much time was timed
STATE_DIM = {"cube3": 54, "puzzle15": 16, "puzzle24": 25, "puzzle35": 36, "puzzle48": 49}


def synth_states(n: int, d: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.permuted(np.tile(np.arange(d, dtype=np.uint8), (n, 1)), axis=1)


_BACKEND = "nccl"


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside a launcher: re-exec under torch.distributed.run, one rank per GPU
    (the reference pins a worker to its GPU the same way, one process each: nnet_utils.py:292-301)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dist_setup(backend: str = "nccl"):
    """One rank per GPU: rank r is pinned to device LOCAL_RANK before anything touches HIP, and the process group is
    bound to that device (RCCL never has to guess it).  gloo lets several ranks share a device (CPU-side smoke tests)."""
    global _BACKEND
    _BACKEND = backend
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            if local >= ndev:
                raise SystemExit("bench.py: rank %d wants GPU %d but only %d are visible (one rank per GPU)" % (rank, local, ndev))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            if ndev:
                torch.cuda.set_device(local % ndev)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            if ndev:
                # ranks sharing a GPU (a test box) are COUNTED over the process group by physical device identity, like the
                # CLI does: the grid-wide refinement of giant tie bins needs every workgroup of a launch resident, which two
                # processes on one device cannot promise each other, and node pools take 1/sharers of the HBM
                from deepcubea_amd import _lib
                from deepcubea_amd.search_methods import sharding
                if sharding.count_sharers(world) > 1:
                    _lib.check(_lib.lib().dca_debug_tune(5, 1), "dca_debug_tune")
    elif ndev:
        torch.cuda.set_device(0)
    return world, rank, local


def barrier(world: int):
    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        if _BACKEND == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
    if gpu:
        torch.cuda.synchronize()


def reduce_ranks(x: float, world: int, op: str) -> float:
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda" if _BACKEND == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def gather_ranks(x: float, world: int):
    if world == 1:
        return [x]
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda" if _BACKEND == "nccl" else "cpu")
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


PMC_SOURCE = {}  # kernel -> which committed PMC summary its traffic figure came from (and the episode shape it was taken at)


def pmc_traffic(kernel: str, env: str, B: int):
    """HBM bytes per launch from the rocprofv3 PMC passes of THIS command (`tools/pmc_summary.py` writes
    profiles/r03_pmc_traffic.json from `rocprofv3 --pmc ... -- python bench.py`): counters cannot be read from inside
    the process, so the entry is matched on kernel, environment and batch size and otherwise left null."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if d.get("env") == env and d.get("batch_size") == B and kernel in d.get("kernels", {}):
            PMC_SOURCE[kernel] = "profiles/%s (%s)" % (name, d.get("shape", "rocprofv3 PMC passes of bench.py at 100-step episodes"))
            return d["kernels"][kernel]
    return None


def test_root(idx: int, env: str = "cube3") -> np.ndarray:
    """Scramble `idx` of the shipped test set of `env` (data/<env>/test, kept as a fixture)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
    st = g[env + "_test_states"]
    return np.ascontiguousarray(st[idx % st.shape[0]])


def run_selftest(args, world, rank):
    """No GPU work: exercises the launcher / rendezvous / timing-barrier / reduction plumbing and the JSON contract
    (tests/test_bench_cpu.py runs it with 2 gloo ranks on the CPU box)."""
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    barrier(world)
    wall = reduce_ranks(time.perf_counter() - t0, world, "max")
    units = reduce_ranks(float(args.steps * (rank + 1)), world, "sum")
    return {"value": units / wall, "ms_per_step": wall / args.steps * 1e3,
            "config": {"workload": "selftest (no GPU work)", "parallelism": "x%d" % world},
            "per_rank_value": gather_ranks(float(rank), world)}


# --------------------------------------------------------------------------------------------------
# workload: astar (configs[2] geometry, engine + heuristic)
# --------------------------------------------------------------------------------------------------
def engine_bytes(env: str, B: int, onehot_bytes: int):
    """Algorithmic bytes per launch of the iteration's kernels and SURVEY §8(d)'s per-expansion
    figure for the whole iteration: n^2-free, each array the kernel must touch counted once.  The colour-index
    (network-input) rows are not part of it: with a built-in heuristic nobody reads them and the launch no longer writes
    them (round 5; they were 13 MB of the 43.5 MB counted for this launch in round 4)."""
    D = STATE_DIM[env]
    A = 12 if env == "cube3" else 4
    depth = 6 if env == "cube3" else D
    M = B * A
    per_child_expand = D + 8 + 4 + 4 + 4 + 1 + 1 + 1 + D * depth * onehot_bytes
    return {
        "expand": B * (D + 4 + 4) + M * per_child_expand,
        # hash 8, own row D, slot compare-and-swap 16 + 8, chain hook 8, next/slot/v0/flags 13, representative's row
        # for the ~15 % duplicates
        "probe": M * (8 + D + 24 + 8 + 13 + 0.15 * D),
        "commit": M * (1 + 4 + 4 + 1 + 4 + 1 + 0.85 * 16),
        "per_expansion_8d": (D + A * D + A * D * depth * onehot_bytes) + A * 64,
    }


def run_astar_leg(args, world, rank, onehot_dtype, min_timed_s, profile_iters):
    """Episodes of [reset on a fresh scramble, W untimed iterations, K timed iterations], repeated until min_timed_s of
    timed region accumulated (a batch-20 000 iteration is ~0.1 ms: K alone would be a few milliseconds).  A real cube3
    search lasts a few hundred iterations (results/cube3: <= 6.1e7 nodes), so episodes — not one endless search — are
    the workload.  Every timed region is bracketed by barrier + synchronize on all ranks."""
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods.engine import BwasEngine
    B, w = args.batch_size, args.weight
    sem = _lib.SEM_CPP if args.semantics == "cpp" else _lib.SEM_PY
    hid = _lib.HEUR_HASHU01
    A = 12 if args.env == "cube3" else 4
    fill = 8  # iterations a fresh search needs before every pop is a full batch (12^5 > 20 000)
    warm = max(args.warmup, fill)
    profile_iters = min(args.steps, 256) if profile_iters > 0 else 0  # the profile covers the timed window's iterations
    iters_cap = warm + args.steps + 24
    max_nodes = max(1 << 20, iters_cap * B * A + (1 << 16))
    eng = BwasEngine(args.env, w, B, max_nodes=max_nodes, semantics=sem, onehot_dtype=onehot_dtype)
    graph = not args.no_graph
    total_t, total_exp, total_gen, local_exp, episodes, dev_ms = 0.0, 0.0, 0.0, 0.0, 0, 0.0
    st1 = None
    while True:
        root = test_root(rank + world * episodes, args.env)
        eng.reset(root)
        if sem == _lib.SEM_PY:
            eng.root_commit(_lib.heuristic_builtin(hid, torch.from_numpy(root[None].copy()).cuda()))
        eng.run_builtin(hid, warm, use_graph=graph)
        st0 = eng.status()
        barrier(world)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        eng.run_builtin(hid, args.steps, use_graph=graph)
        e1.record()
        barrier(world)
        wall = time.perf_counter() - t0
        st1 = eng.status()
        assert not st1["failed"] and not st1["done"], "benchmark search ended early: %r" % (st1,)
        assert st1["iterations"] - st0["iterations"] == args.steps
        total_t += reduce_ranks(wall, world, "max")
        total_exp += reduce_ranks(float(st1["nodes_expanded"] - st0["nodes_expanded"]), world, "sum")
        total_gen += float(st1["nodes_generated"] - st0["nodes_generated"])
        local_exp += float(st1["nodes_expanded"] - st0["nodes_expanded"])
        dev_ms += e0.elapsed_time(e1)
        episodes += 1
        # every rank takes the same decision (total_t is already the max over ranks)
        if total_t >= min_timed_s or episodes >= 400:
            break
    res = {"value": total_exp / total_t, "ms_per_step": total_t / (episodes * args.steps) * 1e3, "episodes": episodes,
           "warmup_effective": warm,
           "timed_s": total_t, "device_ms_per_step": dev_ms / (episodes * args.steps), "local_value": local_exp / total_t,
           "open_size_end": st1["open_size"], "closed_size_end": st1["closed_size"],
           "nodes_generated_timed_rank0": total_gen}
    if profile_iters > 0:
        # One more episode of exactly the timed shape — fresh scramble, the same W untimed iterations, then the K iterations
        # that the timed regions cover (rebase iterations included) — replayed one at a time with the device-side stamps
        # switched on (dca.h: profile_builtin), so that sum(span) + sum(gap) reconciles with ms_per_step.
        root = test_root(rank + world * episodes, args.env)
        eng.reset(root)
        if sem == _lib.SEM_PY:
            eng.root_commit(_lib.heuristic_builtin(hid, torch.from_numpy(root[None].copy()).cuda()))
        eng.run_builtin(hid, warm, use_graph=graph)
        prof = eng.profile_builtin(hid, profile_iters, use_graph=graph)
        res["profile"] = prof
        res["profile_iters"] = profile_iters
        dbg = eng.debug()
        res["front_n"], res["n_ord"] = dbg["front_n"], dbg["n_ord"]
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return res


def run_astar(args, world, rank):
    B, w = args.batch_size, args.weight
    A = 12 if args.env == "cube3" else 4
    leg = run_astar_leg(args, world, rank, None, MIN_TIMED_S, args.profile_iters)
    per_rank = gather_ranks(leg["local_value"], world)
    res = {
        "value": leg["value"],
        "ms_per_step": leg["ms_per_step"],
        "config": {"workload": "%s BWAS iteration on the device-resident engine, batch %d, weight %.2f, "
                               "%s semantics, heuristic = built-in 10+5*u01(hash) (engine-only, SURVEY §8d), no one-hot "
                               "rows (see engine_onehot_f32 for the north star's fused one-hot); BASELINE configs[2] "
                               "geometry; %d episode(s) of %d timed steps on fresh test-set scrambles: the timed window is "
                               "iterations %d..%d of each search (--warmup %d%s: a search needs 8 iterations before every pop "
                               "is a full batch, and the engine's first 8 iterations are rebase iterations; later every 16th is)"
                               % (args.env, B, w, args.semantics, leg["episodes"], args.steps, leg["warmup_effective"],
                                  leg["warmup_effective"] + args.steps - 1, args.warmup,
                                  " raised to %d" % leg["warmup_effective"] if leg["warmup_effective"] != args.warmup else ""),
                   "env": args.env, "batch_size": B, "weight": w, "children_per_step": B * A, "semantics": args.semantics,
                   "hipgraph": not args.no_graph, "parallelism": "one search instance per GPU x%d" % world,
                   "episodes": leg["episodes"], "timed_s": leg["timed_s"],
                   "untimed_iterations_per_episode": leg["warmup_effective"],  # max(--warmup, 8): until every pop is a full batch
                   "open_size_end": leg["open_size_end"], "closed_size_end": leg["closed_size_end"],
                   "device_ms_per_step": leg["device_ms_per_step"]},
    }
    alg = engine_bytes(args.env, B, 0)
    if "profile" in leg:
        span, gap = leg["profile"]["span_ms"], leg["profile"]["gap_ms"]
        n_front, n_ord = leg["front_n"] + 8 * B, max(leg["n_ord"], B)  # (FRONT as stored: live entries + tombstones)
        # k_sel_collect reads FRONT's keys and moves the batch's bins (key 8 + id 4 + slot 4 + bin 2, read and written);
        # k_rank reads those and writes the batch in pop order; k_sel_scan walks the 4096-bin histogram.  The rebase pass
        # (slot "sel_hist", every 8th iteration: its span here is averaged over all profiled iterations) is left out.
        alg_k = dict(alg, sel_collect=8.0 * n_front + 36.0 * n_ord, rank=18.0 * n_ord + 16.0 * B, sel_scan=4096 * 16.0)
        if "probe" not in span:
            # four launches per iteration: the CLOSED probe runs inside the expansion launch, from the rows the tile holds in
            # LDS — its own row (D) is no longer re-read and the hash (8 written + 8 read) never leaves the registers
            alg_k["expand"] = alg["expand"] + alg["probe"] - B * A * (STATE_DIM[args.env] + 16)
        cand = {k: v for k, v in span.items() if k in alg_k and not k.startswith("per_")}
        dom = max(cand, key=cand.get)  # the launch that takes the most time, whatever bounds it
        ach = alg_k[dom] / (span[dom] * 1e-3) / 1e9
        res["roofline"] = {
            "bound": "hbm", "kernel": "k_" + dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic("k_" + dom, args.env, B),
            "traffic_source": None,  # filled below: which committed PMC summary, taken at which episode shape
            "bytes_per_launch": alg_k[dom], "kernel_ms": span[dom],
            "timing": "device wall-clock stamps written by every workgroup of the launch INSIDE the replayed hipGraph "
                      "(dca_engine_profile_builtin; max end - min start, averaged over the %d iterations of one "
                      "more episode of the timed shape: same warm-up, same K iterations, rebase iterations included); HIP "
                      "events bracket the K-step regions (device_ms_per_step)" % leg["profile_iters"],
            # every launch of the iteration against the same roofline (algorithmic bytes / its own span): the table the
            # single `kernel` above is the longest row of
            "launch_rooflines": {"k_" + k: {"bytes_per_launch": alg_k[k], "kernel_ms": round(span[k], 5),
                                            "achieved_GBs": round(alg_k[k] / (span[k] * 1e-3) / 1e9, 1),
                                            "frac": round(alg_k[k] / (span[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for k in cand},
            "launch_span_ms": {k: round(v, 5) for k, v in span.items()},
            "launch_gap_ms": {k: round(v, 5) for k, v in gap.items()},
            "sum_span_ms": sum(v for k, v in span.items() if not k.startswith("rank_")),
            "sum_gap_ms": sum(v for k, v in gap.items() if not k.startswith("rank_")),
            "unprofiled_ms": leg["ms_per_step"] - sum(v for k, v in span.items() if not k.startswith("rank_"))
            - sum(v for k, v in gap.items() if not k.startswith("rank_")),  # iteration-to-iteration hand-over inside the graph chain
            # plain iteration: k_sel_collect (histogram scan fused in), k_rank, k_expand (+ CLOSED probe), k_commit; the
            # sel_hist / sel_scan spans are phases of the rebase iterations' extra launches, averaged over all iterations
            "launches_per_iteration": len([k for k in span if not k.startswith(("refill", "rank_", "sel_hist", "sel_scan"))]),
            "launches_every_16th_iteration_extra": 5,
            "graph_launches": "runs of iterations are replayed as chunk hipGraphs of up to 64 iterations (dca_engine_run_builtin)",
            "note": "k_%s is the longest launch of the iteration; it is bound by dependent memory round trips (hash-table "
                    "probe / rank chains), not by HBM bandwidth, so its fraction of the HBM peak is low by construction — "
                    "with the fp32 one-hot rows written by the same launch it is HBM-write bound (engine_onehot_f32.roofline_expand)" % dom,
        }
    if "roofline" in res:
        res["roofline"]["traffic_source"] = PMC_SOURCE.get(res["roofline"]["kernel"])
    it_bytes = alg["per_expansion_8d"] * B
    res["roofline_iteration"] = {"bound": "hbm", "what": "whole BWAS iteration: SURVEY §8(d) bytes per expansion "
                                 "(%d B, no one-hot) x batch / ms_per_step" % alg["per_expansion_8d"],
                                 "achieved": it_bytes / (leg["ms_per_step"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": it_bytes / (leg["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    res["per_rank_value"] = per_rank
    if args.onehot_leg:
        oh = run_astar_leg(args, world, rank, torch.float32, MIN_TIMED_S, args.profile_iters)
        algo = engine_bytes(args.env, B, 4)
        ob = algo["per_expansion_8d"] * B
        leg_o = {"value": oh["value"], "unit": "nodes expanded/s", "ms_per_step": oh["ms_per_step"], "episodes": oh["episodes"],
                 "what": "same iteration with the fp32 one-hot rows of all %d children written by the expansion launch "
                         "(north star: one-hot fused into the same launch)" % (B * A),
                 "roofline_iteration": {"bytes_per_expansion_8d": algo["per_expansion_8d"],
                                        "achieved": ob / (oh["ms_per_step"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": ob / (oh["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        if "profile" in oh and "expand" in oh["profile"]["span_ms"]:
            sp = oh["profile"]["span_ms"]["expand"]
            leg_o["roofline_expand"] = {"kernel": "k_expand<cube3,onehot f32>", "bytes_per_launch": algo["expand"],
                                        "kernel_ms": sp, "achieved": algo["expand"] / (sp * 1e-3) / 1e9,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": algo["expand"] / (sp * 1e-3) / 1e9 / HBM_PEAK_GBS}
        res["engine_onehot_f32"] = leg_o
    if args.concurrent != "0" and args.env == "cube3":
        from deepcubea_amd import _lib
        sem = _lib.SEM_CPP if args.semantics == "cpp" else _lib.SEM_PY
        sweep = {}
        ks = [int(x) for x in args.concurrent.split(",") if int(x) > 1]
        for k in ks:  # (the device-side launch profile rides on the largest K: the leg the CLI's `auto` picks)
            sweep[str(k)] = run_astar_concurrent(args, world, rank, sem, _lib.HEUR_HASHU01, k, profile=(k == max(ks)))
        if sweep:
            best = max(sweep, key=lambda k: sweep[k]["value"])
            res["concurrent_instances"] = dict(sweep[best], sweep={k: {"value": v["value"], "ms_per_step": v["ms_per_step"],
                                                                       "roofline_frac": v["roofline_iteration"]["frac"]}
                                                                   for k, v in sweep.items()},
                                               note="K searches share every launch (grid.y = instance): the CLI's "
                                                    "--instances_per_gpu auto picks K from this kind of sweep "
                                                    "(search_methods/astar.py:auto_instances)")
    if args.expand_block:
        res["expand_1M"] = expand_block(args, world, rank)
    if args.queue_states > 0:
        res["sharded_queue"] = run_sharded_queue(args, world, rank)
    return res


def run_sharded_queue(args, world, rank):
    """BASELINE configs[3]'s sharding path at a size that means something: `--queue-states` x world shipped puzzle15 test
    scrambles drawn from the shared work queue (search_methods/sharding.WorkQueue: an atomic counter in the process
    group's store, no collective), each searched TO COMPLETION with the built-in Manhattan heuristic (batch 10 000,
    weight 0.8): per-state cost varies ~50x (1e6 .. 7e7 nodes), which is what the queue is there to balance (the
    published cube3 searches vary 40x, results/cube3/output.txt).  value = nodes expanded by all ranks / wall time,
    resets, ramp-up and the tie-heavy integer-cost f-levels included."""
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods import sharding
    from deepcubea_amd.search_methods.engine import BwasEngine
    env, B, w, hid = "puzzle15", 10000, 0.8, _lib.HEUR_MANHATTAN
    n_states = args.queue_states * world
    # (a pool of 2^27 ids — or this rank's share of the device when ranks share a GPU)
    eng = BwasEngine(env, w, B, max_nodes=min(1 << 27, BwasEngine.auto_max_nodes(env, B, 1, sharers=sharding.ranks_on_my_device())))
    queue = sharding.WorkQueue(n_states, world, rank, key="dca_bench_queue")
    barrier(world)
    t0 = time.perf_counter()
    mine, expanded, generated, lens = 0, 0, 0, 0
    while True:
        nxt = queue.next(1)
        if not nxt:
            break
        res = eng.solve_builtin(test_root(nxt[0], env), hid, chunk=32, use_graph=not args.no_graph)
        assert res["solved"], res
        expanded += res["nodes_expanded"]
        generated += res["nodes_generated"]
        lens += len(res["moves"])
        mine += 1
    barrier(world)
    wall = reduce_ranks(time.perf_counter() - t0, world, "max")
    total = reduce_ranks(float(expanded), world, "sum")
    total_gen = reduce_ranks(float(generated), world, "sum")
    counts = gather_ranks(float(mine), world)
    eng.close()
    torch.cuda.empty_cache()
    return {"value": total / wall, "unit": "nodes expanded/s", "nodes_generated_per_s": total_gen / wall, "states": n_states,
            "env": env, "batch_size": B, "weight": w, "heuristic": "built-in Manhattan distance", "states_per_rank": counts,
            "seconds": wall, "mean_solution_length": reduce_ranks(float(lens), world, "sum") / max(n_states, 1),
            "how": "per-instance sharding through the shared work queue; every search runs to completion (resets, ramp-up, "
                   "refills, tie groups included)"}


def run_astar_concurrent(args, world, rank, sem, hid, k, profile=False):
    """k independent search instances per GPU stepped together by ONE engine (grid.y = instance; finer per-instance
    sharding, like the reference's AStar stepping a list of instances): a batch-20 000 iteration is launch/latency
    bound and leaves most of the chip idle.  Reported next to the single-instance `value`, never instead of it."""
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods.engine import BwasEngine
    B, w = args.batch_size, args.weight
    steps, warm = min(args.steps, 50), max(args.warmup, 8)  # (K node pools: episodes of at most 50 timed iterations)
    eng = BwasEngine("cube3", w, B, max_nodes=max(1 << 20, (steps + warm + 8) * B * 12 + (1 << 16)), semantics=sem,
                     num_instances=k)
    wall_sum, total, episodes = 0.0, 0.0, 0
    while True:  # episodes of [reset K fresh scrambles, W untimed, K timed iterations] like the single-instance leg
        for i in range(k):
            root = test_root((rank + world * episodes) * k + i)
            eng.reset(root, i)
            if sem == _lib.SEM_PY:
                eng.root_commit(_lib.heuristic_builtin(hid, torch.from_numpy(root[None].copy()).cuda()), i)
        eng.run_builtin(hid, warm, use_graph=not args.no_graph)
        st0 = [eng.status(i) for i in range(k)]
        barrier(world)
        t0 = time.perf_counter()
        eng.run_builtin(hid, steps, use_graph=not args.no_graph)
        barrier(world)
        wall = time.perf_counter() - t0
        st1 = [eng.status(i) for i in range(k)]
        expanded = sum(b["nodes_expanded"] - a["nodes_expanded"] for a, b in zip(st0, st1))
        assert all(not s["failed"] and not s["done"] for s in st1)
        wall_sum += reduce_ranks(wall, world, "max")
        total += reduce_ranks(float(expanded), world, "sum")
        episodes += 1
        if wall_sum >= MIN_TIMED_S / 2 or episodes >= 100:
            break
    ms_step = wall_sum / (steps * episodes) * 1e3
    out = {"instances_per_gpu": k, "value": total / wall_sum, "unit": "nodes expanded/s",
           "ms_per_step": ms_step, "episodes": episodes, "steps_per_episode": steps,
           "how": "one engine, every kernel launched once for all instances (grid.y = instance)"}
    # the same yardstick as the single-search headline ("so that 4.4e8 has a fraction next to it"):
    # SURVEY §8(d) bytes per expansion x batch x K / ms_per_step, and — one more episode of the timed shape with the device
    # stamps on, every instance's workgroups stamping — the launches' envelopes against their algorithmic bytes x K
    alg = engine_bytes("cube3", B, 0)
    itb = alg["per_expansion_8d"] * B * k
    out["roofline_iteration"] = {"bound": "hbm", "bytes_per_step": itb, "achieved": itb / (ms_step * 1e-3) / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": itb / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if profile and args.profile_iters > 0:
        for i in range(k):
            root = test_root((rank + world * episodes) * k + i)
            eng.reset(root, i)
            if sem == _lib.SEM_PY:
                eng.root_commit(_lib.heuristic_builtin(hid, torch.from_numpy(root[None].copy()).cuda()), i)
        eng.run_builtin(hid, warm, use_graph=not args.no_graph)
        prof = eng.profile_builtin(hid, steps, use_graph=not args.no_graph)
        span = prof["span_ms"]
        alg_k = {"expand": (alg["expand"] + alg["probe"] - B * 12 * (54 + 16)) * k, "commit": alg["commit"] * k}
        out["launch_span_ms"] = {kk: round(v, 5) for kk, v in span.items()}
        out["launch_gap_ms"] = {kk: round(v, 5) for kk, v in prof["gap_ms"].items()}
        out["launch_rooflines"] = {"k_" + kk: {"bytes_per_launch": alg_k[kk], "kernel_ms": round(span[kk], 5),
                                                "achieved_GBs": round(alg_k[kk] / (span[kk] * 1e-3) / 1e9, 1),
                                                "frac": round(alg_k[kk] / (span[kk] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                "traffic": (lambda t: None if t is None else t * k)(pmc_traffic("k_" + kk, "cube3", B)),
                                                "traffic_how": "the single-instance launch's PMC traffic x K (same per-instance work)"}
                                   for kk in alg_k if kk in span and span[kk] > 0}
    eng.close()
    torch.cuda.empty_cache()
    return out


def run_astar_nnet(args, world, rank, dtype_name: str, eval_all_children: bool = False, gemm16: str = "hip"):
    """Same loop, heuristic = ResNet(54*6 -> 5000 -> 1000 -> 4 res blocks -> 1) on PyTorch-ROCm, synthetic weights
    (numpy PCG64 seed 2024).  Default = the CLI's default path: dedup-first engine stepping (only the children that
    survive the CLOSED check are evaluated — same search, astar.py:272-282) + the padded / epilogue-fused network
    layout (FastResnet).  eval_all_children = the reference's order on the plain BN-folded network."""
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods.engine import BwasEngine
    from deepcubea_amd.utils import nnet_utils
    from deepcubea_amd.utils.pytorch_models import FastResnet, Fp8Resnet, ResnetModel, fold_batchnorm
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    from deepcubea_amd.utils import env_utils
    B, w = args.batch_size, args.weight
    # `steps` timed iterations per round, rounds repeated until MIN_TIMED_S were timed; one whole round untimed first: the rows
    # handed to the network change from step to step (padded to 1024), and a first-time shape costs the caching allocator a
    # device allocation — seen as a one-off ~130 ms stall inside the first timed round of some runs (fp32 leg: 28 instead of 17 ms)
    steps, warm = args.nnet_steps, 2 + args.nnet_steps
    env = env_utils.get_environment(args.env)
    A = env.get_num_moves()
    model = env.get_nnet_model()  # cube3: ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(model, 2024)
    macs = sum(m.in_features * m.out_features for m in model.modules() if isinstance(m, torch.nn.Linear))
    fp8_scaling = "block" if dtype_name == "fp8mx" else "tensor"
    if dtype_name == "fp8mx":
        dtype_name = "fp8"
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16, "fp8": torch.bfloat16}[dtype_name]
    max_rounds = 64
    cap = max(1 << 20, (steps * max_rounds + warm + 12) * B * A + 64 * B * A // 2)
    if eval_all_children:
        model = fold_batchnorm(model).cuda().eval()
        hfn = nnet_utils.get_heuristic_fn_dev(model, clip_zero=False, batch_size=args.nnet_batch_size,
                                              autocast_dtype=None if dt == torch.float32 else dt)
        eng = BwasEngine(args.env, w, B, max_nodes=cap, onehot_dtype=dt)
    else:
        fast = (Fp8Resnet(model, scaling=fp8_scaling) if dtype_name == "fp8" else FastResnet(model, dt, gemm16=gemm16)).cuda()
        hfn = nnet_utils.get_heuristic_fn_dev(fast, clip_zero=False, batch_size=args.nnet_batch_size)
        if fast.uses_l1_kernel:  # layer 1 = the library's one-hot MFMA kernel on the packed uint8 rows
            eng = BwasEngine(args.env, w, B, max_nodes=cap, packed=True)
        else:
            eng = BwasEngine(args.env, w, B, max_nodes=cap, onehot_dtype=fast.onehot_dtype, packed=True,
                             onehot_stride=fast.in_pad)
    root = test_root(rank, args.env)
    eng.reset(root)
    eng.root_commit(hfn(eng.root_nnet_in()))
    # fill OPEN past one batch quickly with the cheap heuristic so every timed step is a full batch
    for _ in range(64):
        eng.run_builtin(_lib.HEUR_HASHU01, 1)
        if eng.status()["open_size"] >= 3 * B:
            break
    for _ in range(warm):
        eng.step(hfn)
    st0 = eng.status()
    rows0 = eng.rows_evaluated
    wall, rounds = 0.0, 0
    while True:  # (every rank takes the same decision: `wall` is the max over ranks)
        barrier(world)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step(hfn)
        barrier(world)
        wall += reduce_ranks(time.perf_counter() - t0, world, "max")
        rounds += 1
        if wall >= MIN_TIMED_S or rounds >= max_rounds:
            break
    st1 = eng.status()
    assert not st1["failed"] and not st1["done"], "benchmark search ended early: %r" % (st1,)
    steps = steps * rounds
    expanded = st1["nodes_expanded"] - st0["nodes_expanded"]
    rows = (eng.rows_evaluated - rows0) / steps
    total_exp = reduce_ranks(float(expanded), world, "sum")
    # layer 1 as an embedding sum (dca_l1_embed: the sliding puzzles) issues no MFMA flops: the roofline below counts the dense layers only
    l1_embed = (not eval_all_children) and (getattr(fast, "l1_embed_w", None) is not None
                                            or getattr(fast, "l1_embed_w8", None) is not None)
    if l1_embed:
        macs -= model.fc1.in_features * model.fc1.out_features
    flops = 2.0 * macs * rows
    eng.close()
    torch.cuda.empty_cache()
    peak = (157.3 if eval_all_children else 2500.0) if dtype_name == "fp32" else (5000.0 if dtype_name == "fp8" else 2500.0)
    issued = flops * (3.0 if (dtype_name == "fp32" and not eval_all_children) else 1.0) / (wall / steps) / 1e12
    roof = {"bound": "mfma", "achieved": issued, "peak": peak, "unit": "TFLOP/s", "frac": issued / peak,
            "what": "MFMA flops ISSUED per second over the whole iteration (engine launches and host hand-over included): "
                    "2 x MACs x network rows%s / ms_per_step, against the dense peak of the pipe the layers run on"
                    % (" x 3 (f16x3: three f16 products per fp32-accurate one)" if dtype_name == "fp32" and not eval_all_children else ""),
            "mfma_busy_profile": {"fp32": "profiles/r06_nnet_fp32_pmc_mfma.txt", "fp8": "profiles/r06_nnet_fp8_pmc_mfma.txt"}.get(
                dtype_name if not eval_all_children else "", None),
            "clock_note": "the dense layers run power-limited on random operands (1.3-1.7 GHz shader clock inside their K loops, 2.4 GHz "
                          "nominal: profiles/r04_gemm_timeline.txt) and fetch-bound on the L2 -> LDS operand stream "
                          "(profiles/r05_gemm16_probe.txt: all-zero operands run 32 % faster, the library's kernel 6 %)"}
    return {"value": total_exp / wall, "unit": "nodes expanded/s", "ms_per_step": wall / steps * 1e3,
            "roofline_nnet": roof,
            "steps": steps, "timed_s": wall, "heuristic_dtype": dtype_name + (" (block-scaled, MX-64)" if dtype_name == "fp8" and fp8_scaling == "block" else ""), "weights": "synthetic (numpy PCG64 seed 2024, BN folded)",
            "order": "eval_all_children (reference order)" if eval_all_children else "dedup_first (CLI default)",
            "layer1": "library GEMM on one-hot rows" if eval_all_children or not fast.uses_l1_kernel
            else "dca_l1_embed (embedding sum on the vector pipes: one gathered fp32 weight per position, exact fp32; not in the MFMA flop count)" if l1_embed
            else "dca_l1_onehot_gemm8 (hand-written f8f6f4 MFMA, e4m3 weights)" if getattr(fast, "l1_fp8", False)
            else "dca_l1_onehot_gemm (hand-written MFMA, %d bf16 plane(s)%s)" % ((1, ", e4m3 output" + (" with E8M0 block scales" if fp8_scaling == "block" else "")) if dtype_name == "fp8"
                                                                                else (fast.l1_planes, "")),
            "dense_layers": ("library fp32 GEMMs" if eval_all_children else "dca_f16x3_gemm (hand-written MFMA, epilogue-fused)")
            if dtype_name == "fp32" else (("dca_gemm8_mx (hand-written scaled e4m3 MFMA, one E8M0 scale per row and 64 elements; tail + requantise in the "
                                           "epilogue)" if fp8_scaling == "block" else
                                           "dca_gemm8 (hand-written e4m3 MFMA, per-tensor scales; dequantise + tail + requantise in the epilogue)")
                                          if dtype_name == "fp8" else "dca_gemm16 (hand-written MFMA, epilogue-fused)"
                                          if gemm16 == "hip" else "library (hipBLASLt) GEMMs + clamp pass"),
            "network_rows_per_step": rows, "children_per_step": B * A,
            # fp32 parity mode: batches whose activations left the fp16 range and were redone with fp32 GEMMs (0 expected)
            "split_fallbacks": None if eval_all_children else getattr(fast, "split_fallbacks", None),
            "heuristic_tflops_per_gpu": flops / (wall / steps) / 1e12,
            # fp32 default path = f16x3 split layers: 3 f16 MFMA flops per useful flop -> ceiling 2500/3 "fp32-equivalent"
            "mfma_peak_tflops": (157.3 if eval_all_children else 2500.0 / 3) if dtype_name == "fp32"
            else (5000.0 if dtype_name == "fp8" else 2500.0),
            "mfma_pipe": ("f32-input MFMA" if eval_all_children else "f16/bf16 MFMA, fp32-accurate via operand splitting "
                          "(useful flops = 1/3 of the issued ones)") if dtype_name == "fp32"
            else ("f8f6f4 MFMA (e4m3)" if dtype_name == "fp8" else "f16/bf16 MFMA")}


def cpu_baseline_astar(args, seconds_budget: float = 20.0):
    """The build's C++/OpenMP restatement of cpp/parallel_weighted_astar.cpp (oracle, kind=port) on the same
    root, batch size, weight and built-in heuristic, for a bounded number of iterations."""
    from oracle import c_oracle as co
    root = test_root(0)
    cores = co.num_threads()
    sem = co.SEM_CPP if args.semantics == "cpp" else co.SEM_PY
    iters = 6
    r = co.astar("cube3", root, args.weight, args.batch_size, sem, heur_builtin_id=2, max_iters=iters,
                 stop_on_goal=False)
    per = r["seconds"] / max(r["iterations"], 1)
    iters2 = int(min(max(seconds_budget / max(per, 1e-3), iters), 60))
    r = co.astar("cube3", root, args.weight, args.batch_size, sem, heur_builtin_id=2, max_iters=iters2,
                 stop_on_goal=False)
    return {"value": r["nodes_expanded"] / r["seconds"], "unit": "nodes expanded/s", "cores": cores, "kind": "port",
            "sample": "%d BWAS iterations (batch %d, weight %.2f, %s semantics, same root and built-in heuristic) of "
                      "oracle/dca_oracle.cpp — the C++/OpenMP restatement of cpp/parallel_weighted_astar.cpp "
                      "(std::priority_queue OPEN, hash-map CLOSED, OpenMP expand); the reference binary itself "
                      "needs boost and cannot be built here" % (r["iterations"], args.batch_size, args.weight,
                                                                 args.semantics)}


# --------------------------------------------------------------------------------------------------
# workload: avi (configs[4]'s update step)
# --------------------------------------------------------------------------------------------------
def run_avi(args, world, rank):
    """One step = the AVI update of `--n` training states of `--env` on this GPU: random reverse walks from the goal,
    expansion, ResNet heuristic on every child, Bellman backup (1 GBFS step, eps 0) — the data-generation half of
    ctg_approx/avi.py:do_update.  value = training states produced per second."""
    from deepcubea_amd.updaters.updater import Updater
    from deepcubea_amd.utils import env_utils, nnet_utils
    from deepcubea_amd.utils.pytorch_models import FastResnet
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    env = env_utils.get_environment(args.env)
    model = env.get_nnet_model()
    load_synthetic_weights(model, 2024)
    macs = sum(m.in_features * m.out_features for m in model.modules() if isinstance(m, torch.nn.Linear))
    A = env.get_num_moves()
    back_max = {"cube3": 30, "puzzle15": 500, "puzzle24": 500}.get(args.env, 1000)  # the values of the reference's train.sh
    fast = FastResnet(model, torch.bfloat16 if args.nnet_dtype == "bf16" else torch.float32).cuda()
    hfn = nnet_utils.get_heuristic_fn_dev(fast, clip_zero=False, batch_size=args.nnet_batch_size)
    oh = None if fast.uses_l1_kernel else fast.onehot_dtype  # None: the closure gets the uint8 rows (layer-1 MFMA kernel)
    n = args.n if args.n != 1_000_000 else 200_000

    def step(i):
        upd = Updater(env, n * world, back_max, hfn, 1, update_batch_size=100_000, seed=1000 + i, onehot_dtype=oh)
        return upd.update_dev()

    for i in range(args.warmup):
        step(i)
    barrier(world)
    t0 = time.perf_counter()
    tot = 0
    for i in range(args.steps):
        sn, out, sv = step(100 + i)
        tot += sn.shape[0]
    barrier(world)
    wall = reduce_ranks(time.perf_counter() - t0, world, "max")
    total = reduce_ranks(float(tot), world, "sum")
    if fast.l1_embed_w is not None:  # layer 1 as an embedding sum (dca_l1_embed): no multiply-adds issued for it
        macs -= model.fc1.in_features * model.fc1.out_features
    flops = 2.0 * macs * A * (tot / args.steps)
    published = {"cube3": "1.55e5 states/s on 3 GPUs + 30 CPU procs (saved_models/cube3/output.txt)",
                 "puzzle48": "9.4e4 states/s on 4 GPUs + 30 CPU procs (50M states in 528-530 s, saved_models/puzzle48/output.txt)"}
    return {
        "value": total / wall, "ms_per_step": wall / args.steps * 1e3,
        "config": {"workload": "%s AVI update step (BASELINE configs[4] / avi.py:do_update): generate %d states "
                               "(back_max %d) -> expand -> ResNet heuristic on %d children -> Bellman backup; %s "
                               "heuristic, synthetic weights" % (args.env, n, back_max, A, args.nnet_dtype),
                   "states_per_step_per_gpu": n, "back_max": back_max, "gbfs_steps": 1, "nnet_dtype": args.nnet_dtype,
                   "parallelism": "state shards per GPU x%d" % world,
                   "heuristic_tflops_per_gpu": flops / (wall / args.steps) / 1e12,
                   "layer1": "dca_l1_embed (embedding sum, not in the flop count)" if fast.l1_embed_w is not None
                   else ("dca_l1_onehot_gemm (one-hot MFMA kernel)" if fast.uses_l1_kernel else "library GEMM on one-hot rows"),
                   "reference_published": published.get(args.env)},
    }


# --------------------------------------------------------------------------------------------------
# workload: train (the training step, nnet_utils.train_nnet)
# --------------------------------------------------------------------------------------------------
def run_train(args, world, rank):
    """One step = one Adam iteration of `nnet_utils.train_nnet` on the cube3 network (14.7M parameters, BatchNorm in
    training mode, fp32) at `--train_batch` examples per GPU, training set resident in HBM.  N > 1: the network is
    wrapped in DistributedDataParallel (gradient all-reduce over RCCL), weak scaling."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils import env_utils, nnet_utils
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    env = env_utils.get_environment("cube3")
    model = env.get_nnet_model()
    load_synthetic_weights(model, 2024)
    model = model.cuda()
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[torch.cuda.current_device()])
    B = args.train_batch
    n = B * 8
    states, nb, _ = _lib.generate_states(env._env_id, env._dim, n, 0, 30, 5, rank * n)
    x = _lib.nnet_input(env._env_id, env._dim, states)
    y = nb.float()[:, None].contiguous()  # scramble depth as a stand-in target
    dev = torch.device("cuda", torch.cuda.current_device())
    np.random.seed(1 + rank)
    nnet_utils.train_nnet(net, x, y, dev, B, args.warmup, 0, 1e-3, 0.9999993, display=False)
    barrier(world)
    t0 = time.perf_counter()
    last = nnet_utils.train_nnet(net, x, y, dev, B, args.steps, args.warmup, 1e-3, 0.9999993, display=False)
    barrier(world)
    wall = reduce_ranks(time.perf_counter() - t0, world, "max")
    flops = 3 * 2.0 * 14_621_000 * B  # forward + backward (2x) of the dense layers
    return {"value": B * world * args.steps / wall, "ms_per_step": wall / args.steps * 1e3,
            "config": {"workload": "cube3 cost-to-go network training step (nnet_utils.train_nnet: Adam, MSE, lr*lr_d^itr), "
                                   "batch %d per GPU, fp32, BatchNorm training mode, data resident in HBM" % B,
                       "batch_per_gpu": B, "last_loss": last, "parallelism": "DDP x%d (RCCL gradient all-reduce)" % world,
                       "train_tflops_per_gpu": flops / (wall / args.steps) / 1e12, "mfma_peak_tflops": 157.3,
                       "reference_published": "1.26-1.48e5 samples/s at batch 10000 on 3 GPUs with nn.DataParallel "
                                              "(saved_models/cube3/output.txt)"}}


# --------------------------------------------------------------------------------------------------
# workload: expand (configs[1])
# --------------------------------------------------------------------------------------------------
def run_expand(args, world, rank, onehot=None, steps=None, warmup=None):
    from deepcubea_amd import _lib
    n = args.n
    onehot = onehot or args.onehot
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    S = torch.from_numpy(synth_states(n, 54, rank)).cuda()
    ohdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[onehot]
    esz = 4 if onehot == "f32" else 2
    out = {
        "children": torch.empty((n, 12, 54), dtype=torch.uint8, device="cuda"),
        "onehot": torch.empty((n * 12, 324), dtype=ohdt, device="cuda"),
        "solved": torch.empty((n * 12,), dtype=torch.uint8, device="cuda"),
        "hash": torch.empty((n * 12,), dtype=torch.int64, device="cuda"),
    }
    e, d = _lib.ENV_CUBE3, 0

    def step():
        _lib.expand_fused(e, d, S, out=out)

    for _ in range(warmup):
        step()
    barrier(world)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    barrier(world)
    wall = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    wall = reduce_ranks(wall, world, "max")
    total_exp = reduce_ranks(float(n * steps), world, "sum")
    alg = 54 + 12 * 54 + 12 * 324 * esz  # 16 254 B (f32 one-hot) / 8 478 B (16-bit)
    achieved = alg * n / (kern_ms * 1e-3) / 1e9
    # the store-only yardstick in the SAME process, on the same buffer, right after the kernel (the "96 % of
    # the achievable write bandwidth" of round 1 compared numbers from two boxes): the kernel's output bytes written by
    # k_write_ceiling — plain 16-byte stores, one contiguous MiB per workgroup, nothing gathered
    ceil_bytes = out["onehot"].numel() * esz // 16 * 16
    cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(5, steps // 2))]
    for i in range(2 + len(cev)):
        if i >= 2:
            cev[i - 2][0].record()
        _lib.check(_lib.lib().dca_debug_write_ceiling(C.c_void_p(out["onehot"].data_ptr()), C.c_int64(ceil_bytes),
                                                      C.c_int64(1 << 20), _lib.stream_ptr()), "dca_debug_write_ceiling")
        if i >= 2:
            cev[i - 2][1].record()
    torch.cuda.synchronize()
    ceil_ms = float(np.mean([a.elapsed_time(b) for a, b in cev]))
    write_ceiling = ceil_bytes / (ceil_ms * 1e-3) / 1e9
    del out, S
    torch.cuda.empty_cache()
    return {
        "value": total_exp / wall,
        "ms_per_step": wall / steps * 1e3,
        "config": {"workload": "cube3 fused next_state+one-hot(%s)+is_solved+hash kernel, %d synthetic states "
                               "(BASELINE configs[1])" % (onehot, n), "states": n, "moves": 12,
                   "onehot": onehot, "parallelism": "replica-per-gpu x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "expand_fused_kernel<cube3,%s>" % onehot, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": expand_pmc_traffic(onehot, n),
                     "traffic_source": PMC_SOURCE.get("expand_fused_kernel<cube3,%s>" % onehot),
                     "bytes_per_launch": alg * n, "kernel_ms": kern_ms,
                     "timing": "HIP events around each of the %d launches on the launch stream" % steps,
                     "write_ceiling_GBs": write_ceiling, "frac_of_write_ceiling": achieved / write_ceiling,
                     "write_ceiling_how": "same process, same buffer, right after the timed launches: %d bytes (the kernel's "
                                          "one-hot output) by plain 16-byte stores, one contiguous MiB per workgroup "
                                          "(dca_debug_write_ceiling), %.3f ms per launch" % (ceil_bytes, ceil_ms)},
    }


def expand_pmc_traffic(onehot: str, n: int):
    """HBM bytes per launch of the gather kernel from the committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of
    `bench.py --workload expand` (tools/profile_expand.sh -> profiles/rNN_expand_pmc_traffic.json), matched on the one-hot
    type and the number of states; null when no pass at this shape is committed."""
    for name in ("r06_expand_pmc_traffic.json", "r05_expand_pmc_traffic.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        e = d.get(onehot)
        if e and e.get("states") == n:
            PMC_SOURCE["expand_fused_kernel<cube3,%s>" % onehot] = "profiles/%s (%s)" % (name, d.get("shape", ""))
            return e["hbm_bytes"]
    return None


def expand_block(args, world, rank):
    """BASELINE configs[1] on the default line: the gather kernel — fused next_state + one-hot +
    is_solved + hash over 1M synthetic cube3 states — with fp32 and bf16 one-hot rows, each with its HBM roofline."""
    import copy
    a = copy.copy(args)
    a.n = 1_000_000
    out = {}
    for oh in ("f32", "bf16"):
        r = run_expand(a, world, rank, onehot=oh, steps=20, warmup=3)
        out[oh] = {"value": r["value"], "unit": "nodes expanded/s", "ms_per_launch": r["roofline"]["kernel_ms"],
                   "states": a.n, "roofline": r["roofline"]}
    return out


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
    reps = min(max(1, int(seconds_budget / max(dt, 1e-3))), 40)
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
    ap.add_argument("--workload", default="astar", choices=["astar", "expand", "avi", "train", "selftest"])
    ap.add_argument("--nnet_dtype", default="fp32", choices=["fp32", "bf16"], help="avi: heuristic precision")
    ap.add_argument("--env", default="cube3", choices=["cube3", "puzzle15", "puzzle24", "puzzle35", "puzzle48"],
                    help="astar: environment of the engine-only leg (nnet / concurrency legs are cube3)")
    ap.add_argument("--batch_size", type=int, default=20000)
    ap.add_argument("--weight", type=float, default=0.8)
    ap.add_argument("--semantics", default="py", choices=["py", "cpp"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--profile-iters", type=int, default=20, help="astar: extra iterations replayed with the device-side profile on")
    ap.add_argument("--nnet-steps", type=int, default=4, help="astar: timed steps of the ResNet-heuristic leg (0=skip)")
    ap.add_argument("--nnet_batch_size", type=int, default=245760, help="rows per heuristic call (whole astar batch at once)")
    ap.add_argument("--train_batch", type=int, default=10000, help="train: examples per GPU per step")
    ap.add_argument("--n", type=int, default=1_000_000, help="expand: synthetic states per launch")
    ap.add_argument("--onehot", default="f32", choices=["f32", "bf16", "f16"], help="expand: one-hot element type")
    ap.add_argument("--no-onehot-leg", dest="onehot_leg", action="store_false",
                    help="astar: skip the engine leg with the fused fp32 one-hot rows")
    ap.add_argument("--concurrent", default="2,4,8,16",
                    help="astar: also time K concurrent search instances per GPU for every K of this comma list (0 = skip)")
    ap.add_argument("--no-expand-block", dest="expand_block", action="store_false",
                    help="astar: skip the expand_1M block (BASELINE configs[1]: the gather kernel on 1M states)")
    ap.add_argument("--queue-states", type=int, default=32,
                    help="astar: puzzle15 test scrambles per rank drawn from the shared work queue and searched to completion "
                         "in the sharded leg (0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune", default="", help="diagnostics: dca_debug_tune knobs, e.g. 6=1,3=16384 (A/B runs; not for the record)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # one rank per GPU under torch.distributed.run
    if args.steps is None:
        args.steps = {"astar": 200, "expand": 20, "avi": 3, "train": 30, "selftest": 3}[args.workload]
    if args.warmup is None:
        args.warmup = {"astar": 10, "expand": 3, "avi": 1, "train": 5, "selftest": 0}[args.workload]
    world, rank, local = dist_setup(args.dist_backend)
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if args.tune:
        from deepcubea_amd import _lib
        for kv in args.tune.split(","):
            k, v = kv.split("=")
            _lib.check(_lib.lib().dca_debug_tune(int(k), int(v)), "dca_debug_tune")
    res = {"astar": run_astar, "expand": run_expand, "avi": run_avi, "train": run_train,
           "selftest": run_selftest}[args.workload](args, world, rank)
    line = {
        "metric": {"astar": "A* nodes expanded/sec on %s, batch 20k" % args.env, "expand": "A* nodes expanded/sec on cube3",
                   "avi": "AVI update-step training states generated/sec on %s" % args.env,
                   "train": "cost-to-go network training samples/sec on cube3", "selftest": "selftest units/sec"}[args.workload],
        "value": res["value"],
        "unit": {"avi": "states/s", "train": "samples/s"}.get(args.workload, "nodes expanded/s"),
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"expand": "u8", "train": "f32"}.get(args.workload, "u8 states / f64 cost keys / f32 heuristic"),
        "data": "synthetic",
        "config": res["config"],
    }
    if world > 1 and torch.cuda.is_available():
        from deepcubea_amd.search_methods import sharding
        line["ranks_sharing_gpu"] = [int(v) for v in gather_ranks(float(sharding.ranks_on_my_device()), world)]
    for k in ("roofline", "roofline_iteration", "per_rank_value", "engine_onehot_f32", "concurrent_instances", "sharded_queue",
              "expand_1M"):
        if k in res:
            line[k] = res[k]
    if args.workload == "astar" and args.nnet_steps > 0:
        line["end_to_end_nnet"] = {"fp32": run_astar_nnet(args, world, rank, "fp32"),
                                   "bf16": run_astar_nnet(args, world, rank, "bf16", gemm16="hip"),
                                   "bf16_library_gemm": run_astar_nnet(args, world, rank, "bf16", gemm16="library"),
                                   "fp8_hand_written_gemm": run_astar_nnet(args, world, rank, "fp8"),
                                   "fp8_block_scaled": run_astar_nnet(args, world, rank, "fp8mx"),
                                   "fp32_eval_all_children": run_astar_nnet(args, world, rank, "fp32", True)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload in ("astar", "expand"):
        if args.workload == "expand":
            line["cpu_baseline"] = cpu_baseline_expand()
        elif args.env == "cube3":
            line["cpu_baseline"] = cpu_baseline_astar(args)
            # the REFERENCE's own compiled code beside it (oracle/_ref = cpp/environments.cpp built where it lies): the
            # expansion half of the iteration only — the reference's search core needs boost and cannot be built here
            ref = cpu_baseline_expand(seconds_budget=5.0)
            if ref["kind"] == "reference":
                line["cpu_baseline"]["reference_expand"] = ref
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
