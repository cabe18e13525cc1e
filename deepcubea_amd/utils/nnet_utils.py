"""Heuristic service of the BWAS path — mirror of the reference's `utils/nnet_utils.py:122-221`
(get_device, load_nnet, get_heuristic_fn, load_heuristic_fn).  The queue-based multi-process servers
of nnet_utils.py:224-311 are replaced by one replica per GPU (the search shards per instance).

Two closures are offered:
  get_heuristic_fn(...)      reference signature: list of States (or nnet-format arrays) -> np.float64[n]
  get_heuristic_fn_dev(...)  device-resident: uint8 nnet-input tensor [M,D] (or one-hot rows) -> f32 tensor [M]
                             — what the HIP engine calls between pop_expand and commit, no host copies.
"""
from __future__ import annotations

import os
import re
from collections import OrderedDict
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch import nn


def get_available_gpu_nums() -> List[int]:
    """nnet_utils.py:201-203: the variable is parsed by the program itself, keep its name."""
    devices: Optional[str] = os.environ.get('CUDA_VISIBLE_DEVICES')
    return [int(x) for x in devices.split(',')] if devices else []


def get_device() -> Tuple[torch.device, List[int], bool]:
    """nnet_utils.py:122-130, except that a visible HIP device is used even when
    CUDA_VISIBLE_DEVICES is unset (ROCm boxes usually set HIP_VISIBLE_DEVICES or nothing)."""
    devices = get_available_gpu_nums()
    if torch.cuda.is_available():
        if not devices:
            devices = list(range(torch.cuda.device_count()))
        return torch.device("cuda:0"), devices, True
    return torch.device("cpu"), devices, False


def load_nnet(model_file: str, nnet: nn.Module, device: torch.device = None) -> nn.Module:
    """nnet_utils.py:134-152: strips the DataParallel 'module.' prefix."""
    state_dict = torch.load(model_file, map_location=device) if device is not None else torch.load(model_file)
    nnet.load_state_dict(OrderedDict((re.sub(r'^module\.', '', k), v) for k, v in state_dict.items()))
    nnet.eval()
    return nnet


def get_heuristic_fn_dev(nnet: nn.Module, clip_zero: bool = False, batch_size: Optional[int] = None,
                         autocast_dtype: Optional[torch.dtype] = None):
    """Device closure.  fp32 by default (the 1e-5 parity mode); autocast_dtype=torch.bfloat16 is the
    explicitly non-parity fast mode."""
    nnet.eval()
    in_pad = getattr(nnet, "in_pad", None)  # FastResnet: one-hot rows carry its padded stride and dtype
    takes_valid = bool(getattr(nnet, "takes_valid_rows", False))

    @torch.no_grad()
    def heuristic_fn_dev(x: torch.Tensor, is_onehot: bool = False) -> torch.Tensor:
        n = x.shape[0]
        if is_onehot and in_pad is not None and (x.shape[1] != in_pad or x.dtype != nnet.onehot_dtype):
            x = torch.nn.functional.pad(x.to(nnet.onehot_dtype), (0, in_pad - x.shape[1]))
        step = n if batch_size is None else batch_size
        # rows past `valid_rows` (set by BwasEngine.step for one call) are padding: only a model that calibrates on its
        # input cares (Fp8Resnet) — every row is evaluated either way
        valid = getattr(heuristic_fn_dev, "valid_rows", None)
        heuristic_fn_dev.valid_rows = None
        outs = []
        for s in range(0, n, max(step, 1)):
            xb = x[s:s + step]
            with torch.autocast("cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
                if is_onehot:
                    yb = nnet.forward_onehot(xb)
                elif takes_valid:
                    yb = nnet(xb, valid_rows=None if valid is None else max(0, min(valid - s, xb.shape[0])))
                else:
                    yb = nnet(xb)
            outs.append(yb[:, 0].float())
        y = torch.cat(outs) if len(outs) != 1 else outs[0]
        return torch.clamp_min(y, 0.0) if clip_zero else y

    heuristic_fn_dev.valid_rows = None
    return heuristic_fn_dev


def get_heuristic_fn(nnet: nn.Module, device: torch.device, env, clip_zero: bool = False,
                     batch_size: Optional[int] = None):
    """nnet_utils.py:156-198 (same signature and return type)."""
    dev_fn = get_heuristic_fn_dev(nnet, clip_zero=False, batch_size=batch_size)

    def heuristic_fn(states: List, is_nnet_format: bool = False) -> np.ndarray:
        if not is_nnet_format:
            num_states = len(states)
            states_nnet = env.state_to_nnet_input(states) if num_states else [np.zeros((0, env.state_dim), np.uint8)]
        else:
            states_nnet = states
            num_states = states[0].shape[0]
        if num_states == 0:
            return np.zeros(0)
        x = torch.from_numpy(np.ascontiguousarray(states_nnet[0], dtype=np.uint8)).to(device)
        cost_to_go = dev_fn(x).cpu().numpy().astype(np.float64)
        assert cost_to_go.shape[0] == num_states
        if clip_zero:
            cost_to_go = np.maximum(cost_to_go, 0.0)
        return cost_to_go

    return heuristic_fn


def load_heuristic_fn(nnet_dir: str, device: torch.device, on_gpu: bool, nnet: nn.Module, env,
                      clip_zero: bool = False, gpu_num: int = -1, batch_size: Optional[int] = None):
    """nnet_utils.py:206-221 (one replica per process; no DataParallel)."""
    if (gpu_num >= 0) and on_gpu:
        os.environ['CUDA_VISIBLE_DEVICES'] = str(gpu_num)
    nnet = load_nnet("%s/model_state_dict.pt" % nnet_dir, nnet, device=device)
    nnet.eval()
    nnet.to(device)
    return get_heuristic_fn(nnet, device, env, clip_zero=clip_zero, batch_size=batch_size)


# --------------------------------------------------------------------------------------------------
# training step: mirror of nnet_utils.py:31-118 (make_batches, train_nnet)
# --------------------------------------------------------------------------------------------------
def make_batches(num_examples: int, batch_size: int) -> List[np.ndarray]:
    """Index form of nnet_utils.py:31-50: one `np.random.choice(n, n, replace=False)` permutation cut into FULL
    batches (the tail is dropped).  Same numpy call as the reference, so a seeded run visits the same examples."""
    rand_idxs = np.random.choice(num_examples, num_examples, replace=False)
    return [rand_idxs[s:s + batch_size] for s in range(0, num_examples - batch_size + 1, batch_size)]


def train_nnet(nnet: nn.Module, states_nnet, outputs, device: torch.device, batch_size: int, num_itrs: int,
               train_itr: int, lr: float, lr_d: float, display: bool = True,
               batches_idx: Optional[List[np.ndarray]] = None) -> float:
    """nnet_utils.py:53-118, same signature and return value (the last loss): Adam on the MSE between
    `nnet(x)[:, 0]` and `outputs[:, 0]`, learning rate `lr * lr_d**train_itr` set every iteration, batches from
    `make_batches`, reshuffled with `random.shuffle` when exhausted, progress line every 100 iterations.

    MI355X differences: the training set (uint8 network inputs + f32 targets) is copied to the device ONCE and
    batches are gathered there by index — the reference re-uploads every batch; `states_nnet` / `outputs` may
    already be device tensors (what `Updater.update_dev` returns).  Under `torch.distributed` (world > 1) wrap `nnet`
    in DistributedDataParallel and give every rank its own shard with `batch_size // world` — the gradient
    all-reduce over RCCL/xGMI is the only collective of the whole framework and reproduces the reference's
    `nn.DataParallel` arithmetic (per-replica BatchNorm statistics, loss averaged over the global batch)."""
    from random import shuffle
    import time

    display_itrs = 100
    criterion = nn.MSELoss()
    optimizer = torch.optim.Adam(nnet.parameters(), lr=lr)
    start_time = time.time()

    x = states_nnet[0] if isinstance(states_nnet, (list, tuple)) else states_nnet
    x = (torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x).to(device)
    y = torch.from_numpy(np.asarray(outputs, dtype=np.float32)) if isinstance(outputs, np.ndarray) else outputs
    y = y.to(device=device, dtype=torch.float32)
    num_examples = int(y.shape[0])
    batches = batches_idx if batches_idx is not None else make_batches(num_examples, int(batch_size))
    if len(batches) == 0:
        raise ValueError("train_nnet: %d examples do not fill one batch of %d" % (num_examples, batch_size))
    batches = [torch.from_numpy(np.ascontiguousarray(b, dtype=np.int64)).to(device) for b in batches]

    nnet.train()
    max_itrs = train_itr + num_itrs
    last_loss = float("inf")
    batch_idx = 0
    while train_itr < max_itrs:
        optimizer.zero_grad()
        lr_itr = lr * (lr_d ** train_itr)
        for param_group in optimizer.param_groups:
            param_group['lr'] = lr_itr
        idx = batches[batch_idx]
        nnet_cost_to_go = nnet(x[idx])[:, 0]
        target_cost_to_go = y[idx][:, 0]
        loss = criterion(nnet_cost_to_go, target_cost_to_go)
        loss.backward()
        optimizer.step()
        if display and (train_itr % display_itrs == 0):
            last_loss = loss.item()
            print("Itr: %i, lr: %.2E, loss: %.2E, targ_ctg: %.2f, nnet_ctg: %.2f, "
                  "Time: %.2f" % (train_itr, lr_itr, last_loss, target_cost_to_go.mean().item(),
                                  nnet_cost_to_go.mean().item(), time.time() - start_time))
            start_time = time.time()
        train_itr = train_itr + 1
        batch_idx += 1
        if batch_idx >= len(batches):
            shuffle(batches)
            batch_idx = 0
    return float(loss.item())  # one host sync per call instead of one per iteration (nnet_utils.py:100)
