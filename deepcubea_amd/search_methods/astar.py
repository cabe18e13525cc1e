"""Batched weighted A* (BWAS) command line — the `--language hip` driver.

Mirror of the reference CLI `search_methods/astar.py:343-397` (same flags incl. argparse prefix matching such
as `--model` for `--model_dir`, same `results.pkl` keys, same per-state log line, same `--start_idx`
semantics) with the search itself running on the device-resident engine (libdca_hip.so):

    python -m deepcubea_amd.search_methods.astar --states data/cube3/test/data_0.pkl \
        --model_dir saved_models/cube3/current/ --env cube3 --weight 0.6 --batch_size 10000 \
        --results_dir results/cube3/ --language hip --nnet_batch_size 10000

Multi-GPU: launch under `python -m torch.distributed.run --nproc-per-node N ...`; test scrambles are sharded
per instance (a shared work queue by default — per-state cost varies 40x — or state i -> rank i mod N
with --static_shards), every rank holds a replica of the heuristic network and its own OPEN/CLOSED/node pool, and
rank 0 merges the results in state order.  No collective on the data path.

`bwas_hip(args, env, states)` has the signature and return value of the reference's `bwas_python` /
`bwas_cpp` (astar.py:400-568) so it drops into the `--language` switch at astar.py:385-390.
"""
from __future__ import annotations

import os
import pickle
import sys
import time
from argparse import ArgumentParser
from typing import Any, Dict, List, Tuple

import numpy as np
import torch

from .. import _lib
from ..utils import data_utils, env_utils, nnet_utils, search_utils
from . import sharding
from .engine import BwasEngine


_MIN_NNET_ROWS = 1 << 17


def _nnet_rows(args):
    """`--nnet_batch_size` only bounds memory in the reference (astar.py:355, results unchanged).  Its train.sh value
    (10 000 rows) would leave most of an MI355X idle per GEMM and 288 GB of HBM unused, so it is treated as a lower
    bound here: the network sees at least 131 072 rows per call (a whole batch-10 000 iteration)."""
    n = getattr(args, "nnet_batch_size", None)
    return None if n is None else max(int(n), _MIN_NNET_ROWS)


def _load_heuristic(args, env):
    """Device heuristic closure.  `--model_dir synthetic:SEED` builds the environment's network with
    deterministic synthetic weights (the reference checkpoints are not redistributable here)."""
    device, devices, on_gpu = nnet_utils.get_device()
    print("device: %s, devices: %s, on_gpu: %s" % (device, devices, on_gpu))
    if not on_gpu:
        raise _lib.DcaError("--language hip needs an MI355X: no HIP device visible")
    nnet = env.get_nnet_model()
    if str(args.model_dir).startswith("synthetic:"):
        from ..utils.synthetic_weights import load_synthetic_weights
        load_synthetic_weights(nnet, int(str(args.model_dir).split(":", 1)[1]))
        nnet.eval()
    else:
        nnet = nnet_utils.load_nnet("%s/model_state_dict.pt" % args.model_dir, nnet, device=device)
    nnet.to(device)
    dt_name = getattr(args, "nnet_dtype", "fp32")
    if dt_name in ("fp8", "fp8mx") and getattr(args, "eval_all_children", False):
        raise ValueError("--nnet_dtype %s runs on the dedup-first engine path only (drop --eval_all_children)" % dt_name)
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16, "fp8": torch.bfloat16,
          "fp8mx": torch.bfloat16}[dt_name]
    if not getattr(args, "eval_all_children", False):
        # default: padded / epilogue-fused inference layout of the same network, fed by the dedup-first engine
        from ..utils.pytorch_models import FastResnet, Fp8Resnet
        if dt_name in ("fp8", "fp8mx"):
            fast = Fp8Resnet(nnet, scaling="block" if dt_name == "fp8mx" else "tensor").to(device)
        else:
            fast = FastResnet(nnet, dt, gemm16=getattr(args, "gemm16", "hip")).to(device)
        # layer 1 as the library's one-hot MFMA kernel: the engine then hands out uint8 rows only (stride 0 = no one-hot)
        stride = 0 if fast.uses_l1_kernel else fast.in_pad
        args._onehot_dtype = fast.onehot_dtype  # what the engine's pack kernel writes when one-hot rows are needed
        return nnet_utils.get_heuristic_fn_dev(fast, clip_zero=False, batch_size=_nnet_rows(args)), stride
    if getattr(args, "fold_bn", False):
        from ..utils.pytorch_models import fold_batchnorm
        nnet = fold_batchnorm(nnet).to(device)
    ac = None if dt == torch.float32 else dt
    return nnet_utils.get_heuristic_fn_dev(nnet, clip_zero=False, batch_size=_nnet_rows(args), autocast_dtype=ac), None


def _max_nodes(args, instances: int) -> int:
    """`--max_nodes N` or `auto` (default): every generated child takes a node id, and the published cube3 searches reach
    6.1e7 nodes (results/cube3/output.txt) — so the pool is sized from the HBM that is free on this rank's GPU (288 GB on an
    MI355X: ~1e9 ids), not from a constant."""
    v = getattr(args, "max_nodes", "auto")
    if str(v).lower() != "auto":
        return int(v)
    return BwasEngine.auto_max_nodes(args.env, args.batch_size, instances, sharers=sharding.ranks_on_my_device())


# One batch-20 000 cube3 iteration is four dependent launches of 17-29 us that leave most of the 256 CUs idle: K searches
# sharing every launch (grid.y = instance) raise the engine-only rate from 2.1e8 (K = 1) towards 4e8 nodes expanded/s
# (bench.py `concurrent_instances.sweep`); with a network in the loop the iteration is MFMA-bound and K only matters
# while one instance's children are fewer rows than a GEMM needs to fill the chip (_MIN_NNET_ROWS).
_AUTO_CHILDREN_BUILTIN = 3_800_000   # children per launch the sweep still gains up to (B 20 000 x 12 moves -> K = 16: 4.0e8)
_AUTO_MAX_INSTANCES = 16


def auto_instances(args, env, n_states: int, builtin) -> int:
    """`--instances_per_gpu`: the number given, or for `auto` the smallest K whose K x batch x moves children fill the
    chip — _AUTO_CHILDREN_BUILTIN per launch for a built-in heuristic, _MIN_NNET_ROWS per network call otherwise —
    never more than the states this rank can draw, _AUTO_MAX_INSTANCES, or what `--max_nodes` leaves room for.
    Integer-valued built-ins (manhattan, zero) make every f-level ONE cost tie of up to millions of entries, which only a
    single-instance engine refines with its whole grid: those searches stay at K = 1."""
    v = str(getattr(args, "instances_per_gpu", "auto")).lower()
    if v != "auto":
        return max(1, int(v))
    if builtin in (_lib.HEUR_MANHATTAN, _lib.HEUR_ZERO):
        return 1
    per = max(1, int(args.batch_size) * env.get_num_moves())
    want = _AUTO_CHILDREN_BUILTIN if builtin is not None else _MIN_NNET_ROWS
    world, _ = sharding.world_info()
    mine = max(1, -(-int(n_states) // max(world, 1)))
    K = max(1, min(-(-want // per), _AUTO_MAX_INSTANCES, mine))
    # every instance owns a node pool.  `--max_nodes auto`: keep each at >= 2^27 ids (the published cube3 searches reach
    # 6.1e7 nodes); an explicit `--max_nodes N` is ids PER SEARCH and is passed through unchanged, so K shrinks until K
    # pools of N ids fit this rank's share of the HBM (a command that fitted at K = 1 must not fail because of `auto`)
    v = str(getattr(args, "max_nodes", "auto")).lower()
    need = (1 << 27) if v == "auto" else int(v)
    while K > 1 and BwasEngine.auto_max_nodes(args.env, args.batch_size, K, sharers=sharding.ranks_on_my_device()) < need:
        K -= 1
    return K


_BUILTIN = {"manhattan": _lib.HEUR_MANHATTAN, "zero": _lib.HEUR_ZERO, "hashu01": _lib.HEUR_HASHU01}


def bwas_hip(args, env, states: List) -> Tuple[List[List[int]], List[List], List[float], List[int]]:
    """astar.py:400-454 (bwas_python) with the engine in place of AStar.  Returns
    (solns, paths, times, num_nodes_gen) for `states`, in order.
    `--model_dir builtin:manhattan` (or builtin:zero): no network — one of the library's built-in heuristics, evaluated
    inside the expansion launch, the whole search enqueued without host round trips (admissible, consistent Manhattan
    distance on the sliding puzzles: with --weight 1 --semantics cpp the solutions are optimal)."""
    builtin = None
    if str(args.model_dir).startswith("builtin:"):
        name = str(args.model_dir).split(":", 1)[1].lower()
        if name not in _BUILTIN:
            raise ValueError("Unknown built-in heuristic %r (have: %s)" % (name, ", ".join(sorted(_BUILTIN))))
        if name == "manhattan" and not str(args.env).startswith("puzzle"):
            raise ValueError("builtin:manhattan is defined for the sliding puzzles only (it is 0 on %s: "
                             "the search would silently become uniform-cost search)" % args.env)
        builtin = _BUILTIN[name]
        heuristic_fn, onehot_stride = None, None
    else:
        heuristic_fn, onehot_stride = _load_heuristic(args, env)
    sem = _lib.SEM_CPP if getattr(args, "semantics", "py") == "cpp" else _lib.SEM_PY
    oh = getattr(args, "_onehot_dtype", None) or {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16,
                                                   "fp8": torch.bfloat16, "fp8mx": torch.bfloat16}[getattr(args, "nnet_dtype", "fp32")]
    K = auto_instances(args, env, len(states), builtin)
    eng = BwasEngine(args.env, args.weight, args.batch_size, max_nodes=_max_nodes(args, K),
                     semantics=sem, onehot_dtype=None if (onehot_stride == 0 or builtin is not None) else oh,
                     num_instances=K, packed=onehot_stride is not None, onehot_stride=onehot_stride or None)
    world, rank = sharding.world_info()
    local: Dict[int, Tuple[List[int], List, float, int]] = {}
    if getattr(args, "static_shards", False) or world == 1:
        mine = sharding.shard_indices(len(states), world, rank)  # state i -> rank i mod world
        groups = iter([mine[g0:g0 + K] for g0 in range(0, len(mine), K)])
    else:
        queue = sharding.WorkQueue(len(states), world, rank)  # ranks draw the next scramble when they are free
        groups = iter(lambda: queue.next(K), [])
    for group in groups:  # K scrambles stepped together by one engine (one network call per iteration)
        start_time = time.time()
        roots = [np.ascontiguousarray(env._get_arr(states[i]), dtype=np.uint8) for i in group]
        done_at: Dict[int, float] = {}
        if builtin is not None:
            results = eng.solve_many_builtin(roots, builtin, chunk=32, use_graph=True,
                                             on_done=lambda i: done_at.setdefault(i, time.time() - start_time)) \
                if K > 1 else [eng.solve_builtin(roots[0], builtin, chunk=32, use_graph=True)]
        elif K > 1:
            results = eng.solve_many(roots, heuristic_fn, on_done=lambda i: done_at.setdefault(i, time.time() - start_time))
        else:
            results = [eng.solve(roots[0], heuristic_fn)]
        group_time = time.time() - start_time
        for slot, (state_idx, res) in enumerate(zip(group, results)):
            state = states[state_idx]
            if not res["solved"]:
                raise _lib.DcaError("state %d: search stopped without a solution (%s) — raise --max_nodes"
                                    % (state_idx, "node pool exhausted" if res["failed"] else "OPEN empty"))
            soln: List[int] = res["moves"]
            path_cost: float = res["path_cost"]
            num_nodes_gen_idx: int = int(res["nodes_generated"])
            # path of states along the solution (astar.py:213-229 get_path / 534-546 replay)
            path = [state]
            cur = state
            for move in soln:
                cur = env.next_state([cur], move)[0][0]
                path.append(cur)
            # per-state time like astar.py:417,434: with K > 1 the moment THIS instance finished (its search shares
            # every launch with the group's other instances, but it stops generating nodes when it is done)
            solve_time = done_at.get(slot, group_time) if K > 1 else group_time
            assert search_utils.is_valid_soln(state, soln, env)  # astar.py:443
            local[state_idx] = (soln, path, solve_time, num_nodes_gen_idx)
            print("State: %i, SolnCost: %.2f, # Moves: %i, "
                  "# Nodes Gen: %s, Time: %.2f" % (state_idx, path_cost, len(soln), format(num_nodes_gen_idx, ","),
                                                   solve_time))
    eng.close()
    merged = sharding.gather_results(local, len(states), world, rank)
    if merged is None:  # non-zero ranks
        return [], [], [], []
    solns = [merged[i][0] for i in range(len(states))]
    paths = [merged[i][1] for i in range(len(states))]
    times = [merged[i][2] for i in range(len(states))]
    num_nodes_gen = [merged[i][3] for i in range(len(states))]
    return solns, paths, times, num_nodes_gen


def build_parser() -> ArgumentParser:
    parser = ArgumentParser()
    # reference flags (astar.py:346-362)
    parser.add_argument('--states', type=str, required=True, help="File containing states to solve")
    parser.add_argument('--model_dir', type=str, required=True,
                        help="Directory of nnet model, or synthetic:SEED, or builtin:manhattan / builtin:zero")
    parser.add_argument('--env', type=str, required=True, help="Environment: cube3, puzzle15, puzzle24, ...")
    parser.add_argument('--batch_size', type=int, default=1, help="Batch size for BWAS")
    parser.add_argument('--weight', type=float, default=1.0, help="Weight of path cost")
    parser.add_argument('--language', type=str, default="hip", help="hip (the MI355X engine)")
    parser.add_argument('--results_dir', type=str, required=True, help="Directory to save results")
    parser.add_argument('--start_idx', type=int, default=0, help="")
    parser.add_argument('--nnet_batch_size', type=int, default=None,
                        help="How many states the network evaluates at a time (memory only; results unchanged). "
                             "Treated as a lower bound: at least 131072 rows go to the network per call")
    parser.add_argument('--verbose', action='store_true', default=False, help="Set for verbose")
    parser.add_argument('--debug', action='store_true', default=False, help="Set when debugging")
    # engine options
    parser.add_argument('--semantics', type=str, default="py", choices=["py", "cpp"],
                        help="which reference search core to reproduce.  py (default) = search_methods/astar.py, exact: "
                             "pinned node for node to traces recorded from the reference.  cpp = "
                             "cpp/parallel_weighted_astar.cpp, PARITY UNPINNED beyond four recorded answers of the "
                             "reference binary (it needs boost and cannot be rebuilt here): moves, path cost and nodes "
                             "generated match those; |OPEN|/|CLOSED| may differ < 1 %% under float32 cost ties")
    parser.add_argument('--max_nodes', type=str, default="auto",
                        help="node pool capacity (ids per search): a number, or auto = sized from the GPU's free HBM")
    parser.add_argument('--instances_per_gpu', type=str, default="auto",
                        help="scrambles stepped together by one engine (finer per-instance sharding inside a GPU; "
                             "astar.py:232-317 steps a list of instances the same way): a number, or auto = enough "
                             "instances that every launch / network call has a chip-filling amount of work "
                             "(auto_instances).  Results do not depend on it")
    parser.add_argument('--nnet_dtype', type=str, default="fp32", choices=["fp32", "bf16", "fp16", "fp8", "fp8mx"],
                        help="fp32 = parity mode: heuristic values within 1e-5 * max(1, |h|) of the reference's fp32 forward — 1e-5 "
                             "ABSOLUTE for |h| <= 1 (every reference-recorded network fixture); at cube3's trained magnitudes "
                             "|h| ~ 25 measured 0.95 / 1.14 / 1.34e-5 against the reference's fp32 values over three weight seeds "
                             "(5-7 fp32 ulps; the reference's own forward is 0.6-0.7e-5 from float64 there, its module on this "
                             "GPU's fp32 GEMMs 1.5e-5), at puzzle48's |h| ~ 100-300 within 1e-5 * |h| "
                             "(tests/test_parity_configs_hip.py); bf16/fp16 = faster, NOT parity; fp8 = OCP e4m3 operands on the "
                             "hand-written layer kernels (dca_gemm8), one calibrated scale per activation tensor: fastest, "
                             "coarsest; fp8mx = the same with one E8M0 scale per row and 64 elements (nothing to "
                             "calibrate, ~13 %% slower)")
    parser.add_argument('--gemm16', type=str, default="hip", choices=["hip", "library"],
                        help="--nnet_dtype bf16 / fp16: dense layers on the hand-written dca_gemm16 kernel (default) or on the "
                             "library's (hipBLASLt) GEMMs")
    parser.add_argument('--fold_bn', action='store_true', default=False,
                        help="with --eval_all_children: fold BatchNorm into the Linears (always done otherwise)")
    parser.add_argument('--static_shards', action='store_true', default=False,
                        help="multi-GPU: state i -> rank i mod N instead of the shared work queue")
    parser.add_argument('--eval_all_children', action='store_true', default=False,
                        help="reference order (astar.py:272-282): run the network on every child, then drop the "
                             "duplicates.  Default is dedup-first: identical search, fewer network rows")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    world, rank = sharding.init_from_env()
    if rank == 0 and not os.path.exists(args.results_dir):
        os.makedirs(args.results_dir, exist_ok=True)
    results_file: str = "%s/results.pkl" % args.results_dir
    output_file: str = "%s/output.txt" % args.results_dir
    if not args.debug and rank == 0:
        sys.stdout = data_utils.Logger(output_file, "w")

    input_data = data_utils.load_pickle(args.states)
    states = input_data['states'][args.start_idx:]
    env = env_utils.get_environment(args.env)

    results: Dict[str, Any] = dict()
    results["states"] = states
    if args.language == "hip":
        solns, paths, times, num_nodes_gen = bwas_hip(args, env, states)
    else:
        # astar.py:390 — the python / cpp cores are the reference's own; this package only ships hip
        raise ValueError("Unknown language %s" % args.language)
    if rank == 0:
        results["solutions"] = solns
        results["paths"] = paths
        results["times"] = times
        results["num_nodes_generated"] = num_nodes_gen
        data_utils.dump_pickle(results, results_file)  # reference class paths: loads in the reference tree too
    sharding.finalize()


if __name__ == "__main__":
    main()
