"""Approximate value iteration driver — mirror of the reference's `ctg_approx/avi.py` (same flags, same files under
`<save_dir>/<nnet_name>/{current,target}/`, same log lines) around the device update step and the training step:

    python -m deepcubea_amd.ctg_approx.avi --env cube3 --states_per_update 50000000 --batch_size 10000 \
        --nnet_name cube3 --max_itrs 1000000 --loss_thresh 0.1 --back_max 30 --max_update_steps 1

  update   `updaters.Updater` (avi.py:129-159 do_update): states generated, expanded, evaluated by the TARGET network
           (all-zeros heuristic until a target exists, avi.py:219) and backed up on the GPU — no worker processes;
  train    `nnet_utils.train_nnet` (avi.py:238-242) on the device-resident targets;
  test     `updaters.gbfs_test_dev` (avi.py:250-254);
  target   copied from current when the last loss < --loss_thresh (avi.py:262-267).

Multi-GPU: `python -m torch.distributed.run --nproc-per-node N -m deepcubea_amd.ctg_approx.avi ...` — every rank
generates and trains on its share of `--states_per_update` with `--batch_size // N` examples per step; the network is
wrapped in DistributedDataParallel, whose gradient all-reduce over RCCL/xGMI is the framework's only collective and
replaces the reference's `nn.DataParallel` (avi.py:207-208).  `--num_update_procs` is accepted and ignored.
"""
from __future__ import annotations

import os
import pickle
import shutil
import sys
import time
from argparse import ArgumentParser
from typing import Any, Dict, Tuple

import numpy as np
import torch
from torch import nn

from .. import _lib
from ..search_methods import sharding
from ..updaters.updater import Updater, gbfs_test_dev
from ..utils import data_utils, env_utils, nnet_utils


# (flag, type | "flag", default | REQUIRED, help) — names, types and defaults are the reference's (avi.py:21-97; pinned by
# tests/golden/cli_flags.json); `--seed` and `--max_seconds` are the only additions
REQUIRED = object()
_OPTIONS = (
    ("env", str, REQUIRED, "Environment"),
    ("debug", "flag", False, ""),
    ("lr", float, 0.001, "Initial learning rate"),
    ("lr_d", float, 0.9999993, "Learning rate decay: lr * (lr_d ^ itr)"),
    ("max_itrs", int, 1000000, "Maxmimum number of iterations"),
    ("batch_size", int, 1000, "Batch size (global: split evenly over the ranks)"),
    ("single_gpu_training", "flag", False, "accepted for compatibility: one process drives one GPU, more GPUs = more ranks"),
    ("loss_thresh", float, 0.05, "the target network is replaced by the current one when the last loss is below this"),
    ("states_per_update", int, 1000, "training states generated per update"),
    ("epochs_per_update", int, 1, "passes over the generated states per update"),
    ("num_update_procs", int, 1, "ignored: the update runs on the GPU"),
    ("update_nnet_batch_size", int, 10000, "states per heuristic call during the update (memory only)"),
    ("max_update_steps", int, 1, "GBFS steps used to add states to the training set (grows by one per update)"),
    ("update_method", str, "GBFS", "GBFS or ASTAR. If max_update_steps is 1 then either one is the same."),
    ("eps_max", float, 0, "per-instance GBFS eps is uniform in [0, eps_max]"),
    ("num_test", int, 10000, "Number of test states."),
    ("back_max", int, REQUIRED, "Maximum number of backwards steps from goal"),
    ("nnet_name", str, REQUIRED, "Name of neural network"),
    ("update_num", int, 0, "Update number"),
    ("save_dir", str, "saved_models", "Director to which to save model"),
    ("seed", int, 0, "seed of the device state generator (shards are disjoint)"),
    ("max_seconds", float, 0.0, "stop after the update that crosses this much wall time (0 = run to --max_itrs)"),
)


def build_parser() -> ArgumentParser:
    parser = ArgumentParser()
    for name, kind, default, text in _OPTIONS:
        if kind == "flag":
            parser.add_argument("--" + name, action="store_true", default=default, help=text)
        elif default is REQUIRED:
            parser.add_argument("--" + name, type=kind, required=True, help=text)
        else:
            parser.add_argument("--" + name, type=kind, default=default, help=text)
    return parser


def parse_arguments(parser: ArgumentParser, argv=None, rank: int = 0) -> Dict[str, Any]:
    """Derived paths + args.pkl, as avi.py:99-118 leaves them: <save_dir>/<nnet_name>/{target,current}/, output.txt."""
    args = parser.parse_args(argv)
    cfg: Dict[str, Any] = vars(args)
    root = "%s/%s/" % (cfg['save_dir'], cfg['nnet_name'])
    cfg.update(targ_dir="%s/%s/" % (root, 'target'), curr_dir="%s/%s/" % (root, 'current'),
               output_save_loc="%s/output.txt" % root)
    if rank == 0:
        for d in (cfg['targ_dir'], cfg['curr_dir']):
            os.makedirs(d, exist_ok=True)
        where = "%s/args.pkl" % root
        print("Saving arguments to %s" % where)
        with open(where, "wb") as f:
            pickle.dump(args, f, protocol=-1)
        print("Batch size: %i" % cfg['batch_size'])
    return cfg


def copy_files(src_dir: str, dest_dir: str) -> None:
    """Plain files of src_dir -> dest_dir (the current -> target hand-over, avi.py:121-126)."""
    for entry in os.scandir(src_dir):
        if entry.is_file():
            shutil.copy(entry.path, dest_dir)


def load_nnet(nnet_dir: str, env) -> Tuple[nn.Module, int, int]:
    """avi.py:162-173."""
    nnet_file = "%s/model_state_dict.pt" % nnet_dir
    if os.path.isfile(nnet_file):
        nnet = nnet_utils.load_nnet(nnet_file, env.get_nnet_model(), device=torch.device("cpu"))
        itr = pickle.load(open("%s/train_itr.pkl" % nnet_dir, "rb"))
        update_num = pickle.load(open("%s/update_num.pkl" % nnet_dir, "rb"))
    else:
        nnet, itr, update_num = env.get_nnet_model(), 0, 0
    return nnet, int(itr), int(update_num)


def target_heuristic(targ_dir: str, env, device, batch_size: int):
    """The update's heuristic (avi.py:215-224): the target network with clip_zero=True, or all zeros while there is none."""
    targ_file = "%s/model_state_dict.pt" % targ_dir
    if not os.path.isfile(targ_file):
        return lambda x, is_onehot=False: torch.zeros(x.shape[0], dtype=torch.float32, device=x.device)
    from ..utils.pytorch_models import FastResnet
    targ = nnet_utils.load_nnet(targ_file, env.get_nnet_model(), device=torch.device("cpu"))
    fast = FastResnet(targ).to(device)
    fn = nnet_utils.get_heuristic_fn_dev(fast, clip_zero=True, batch_size=batch_size)
    fn.onehot_dtype = None if fast.uses_l1_kernel else fast.onehot_dtype  # None: hand the closure the uint8 rows
    return fn


def do_update(back_max: int, update_num: int, env, max_update_steps: int, update_method: str, num_states: int,
              eps_max: float, heuristic_fn_dev, seed: int, update_batch_size: int):
    """avi.py:129-159 -> (states_nnet u8 [T,D] device, outputs f32 [T,1] device) for this rank."""
    update_steps = min(update_num + 1, max_update_steps)
    num_states = int(np.ceil(num_states / update_steps))
    output_time_start = time.time()
    print("Updating cost-to-go with value iteration")
    if max_update_steps > 1:
        print("Using %s with %i step(s) to add extra states to training set" % (update_method.upper(), update_steps))
    updater = Updater(env, num_states, back_max, heuristic_fn_dev, update_steps, update_method,
                      update_batch_size=update_batch_size, eps_max=eps_max, seed=seed,
                      onehot_dtype=getattr(heuristic_fn_dev, "onehot_dtype", None))
    states_nnet, outputs, is_solved = updater.update_dev()
    if max_update_steps > 1:
        print("%s produced %s states, %.2f%% solved (%.2f seconds)" % (
            update_method.upper(), format(outputs.shape[0], ","), 100.0 * float(is_solved.float().mean()),
            time.time() - output_time_start))
    ctg = outputs[:, 0]
    print("Cost-to-go (mean/min/max): %.2f/%.2f/%.2f" % (float(ctg.mean()), float(ctg.min()), float(ctg.max())))
    return states_nnet, outputs


def main(argv=None):
    # nccl = RCCL over xGMI (one GPU per rank); DCA_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
    world, rank = sharding.init_from_env(os.environ.get("DCA_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo"))
    args_dict = parse_arguments(build_parser(), argv, rank)
    if not args_dict["debug"] and rank == 0:
        sys.stdout = data_utils.Logger(args_dict["output_save_loc"], "a")
    env = env_utils.get_environment(args_dict['env'])
    device, devices, on_gpu = nnet_utils.get_device()
    print("device: %s, devices: %s, on_gpu: %s" % (device, devices, on_gpu))
    if not on_gpu:
        raise _lib.DcaError("avi needs an MI355X: the update step runs on the HIP device (no CPU fallback)")
    device = torch.device("cuda", torch.cuda.current_device())

    nnet, itr, update_num = load_nnet(args_dict['curr_dir'], env)
    update_num = max(update_num, args_dict['update_num'])
    nnet.to(device)
    model = nnet
    if world > 1:
        model = nn.parallel.DistributedDataParallel(nnet, device_ids=[device.index])
    local_batch = max(args_dict['batch_size'] // world, 1)

    t_start = time.time()
    while itr < args_dict['max_itrs']:
        stop = args_dict['max_seconds'] > 0 and time.time() - t_start >= args_dict['max_seconds']
        if world > 1:  # rank 0's clock decides for everybody: a rank that left alone would strand the others in a collective
            flag = torch.tensor([int(stop)], dtype=torch.int32, device=device)
            torch.distributed.broadcast(flag, 0)
            stop = bool(flag.item())
        if stop:
            print("Time budget of %.0f s used (%i iterations, update number %i)" % (args_dict['max_seconds'], itr, update_num))
            break
        # update (targets from the target network)
        hfn = target_heuristic(args_dict['targ_dir'], env, device, args_dict['update_nnet_batch_size'])
        states_nnet, outputs = do_update(args_dict["back_max"], update_num, env, args_dict['max_update_steps'],
                                         args_dict['update_method'], args_dict['states_per_update'], args_dict['eps_max'],
                                         hfn, args_dict['seed'] + 7919 * (itr + 1), max(args_dict['update_nnet_batch_size'], 1) * 10)
        del hfn
        # train.  Under DDP every rank must run the SAME number of steps (each one is a gradient all-reduce), but the
        # shards differ in size: GBFS trajectories end early where an instance solves (--max_update_steps > 1) and
        # states_per_update need not divide evenly.  Agree on the smallest shard and train on that many examples.
        if world > 1:
            cnt = torch.tensor([outputs.shape[0]], dtype=torch.int64, device=device)
            torch.distributed.all_reduce(cnt, op=torch.distributed.ReduceOp.MIN)
            n_common = int(cnt.item())
            if n_common == 0:
                raise RuntimeError("update %d: a rank ended up with no training examples (states_per_update %d over %d "
                                   "ranks) - nothing to train on" % (update_num, args_dict['states_per_update'], world))
            if n_common < outputs.shape[0]:
                states_nnet, outputs = states_nnet[:n_common], outputs[:n_common]
        num_train_itrs = int(args_dict['epochs_per_update'] * np.ceil(outputs.shape[0] / local_batch))
        print("Training model for update number %i for %i iterations" % (update_num, num_train_itrs))
        last_loss = nnet_utils.train_nnet(model, states_nnet, outputs, device, local_batch, num_train_itrs, itr,
                                          args_dict['lr'], args_dict['lr_d'], display=(rank == 0))
        itr += num_train_itrs
        if rank == 0:
            torch.save(nnet.state_dict(), "%s/model_state_dict.pt" % args_dict['curr_dir'])
            pickle.dump(itr, open("%s/train_itr.pkl" % args_dict['curr_dir'], "wb"), protocol=-1)
            pickle.dump(update_num, open("%s/update_num.pkl" % args_dict['curr_dir'], "wb"), protocol=-1)
        # test
        if rank == 0:
            start_time = time.time()
            nnet.eval()
            test_fn = nnet_utils.get_heuristic_fn_dev(nnet, clip_zero=False, batch_size=args_dict['update_nnet_batch_size'])
            gbfs_test_dev(args_dict['num_test'], args_dict['back_max'], env, test_fn,
                          max_solve_steps=min(update_num + 1, args_dict['back_max']), seed=args_dict['seed'] + 1)
            print("Test time: %.2f" % (time.time() - start_time))
        torch.cuda.empty_cache()
        if world > 1:  # every rank takes the same decision from rank 0's loss
            t = torch.tensor([last_loss], device=device)
            torch.distributed.broadcast(t, 0)
            last_loss = float(t.item())
        print("Last loss was %f" % last_loss)
        if last_loss < args_dict['loss_thresh']:
            print("Updating target network")
            if rank == 0:
                copy_files(args_dict['curr_dir'], args_dict['targ_dir'])
            update_num = update_num + 1
            if rank == 0:
                pickle.dump(update_num, open("%s/update_num.pkl" % args_dict['curr_dir'], "wb"), protocol=-1)
        if world > 1:
            torch.distributed.barrier()
    print("Done")
    sharding.finalize()


if __name__ == "__main__":
    main()
