#!/usr/bin/env python3
"""Training step of the cube3 network (batch 10 000, BatchNorm in training mode) with the Linears' forward / input-gradient
GEMMs on dca_f16x3_gemm (`_lib.linear_train`, the default) against the same step on the library's fp32 GEMMs:
  * one step from identical weights: the loss and every parameter's gradient, relative to the gradient's largest element;
  * the loss trajectory of both over 30 Adam steps from the same seeds (rounding differences grow step by step in any
    two fp32 implementations; how fast is what this prints).
Run on the GPU box: python tools/train_grad_check.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcubea_amd import _lib  # noqa: E402
from deepcubea_amd.utils import env_utils  # noqa: E402
from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights  # noqa: E402


def make(seed=2024):
    env = env_utils.get_environment("cube3")
    m = env.get_nnet_model()
    load_synthetic_weights(m, seed)
    return env, m.cuda().train()


def grads(mode, x, y):
    _lib.TRAIN_F16X3 = mode != "library"
    _, net = make()
    if mode == "float64":  # the yardstick: the whole step in double precision (torch's own kernels)
        net = net.double()
        oh = torch.nn.functional.one_hot(x.long(), 6).double().view(x.shape[0], -1)
        out = net.trunk(oh)[:, 0]
        loss = torch.nn.functional.mse_loss(out, y.double())
        loss.backward()
        return float(loss), {k: p.grad.double().cpu() for k, p in net.named_parameters()}
    out = net(x)[:, 0]
    loss = torch.nn.functional.mse_loss(out, y)
    loss.backward()
    return float(loss), {k: p.grad.double().cpu() for k, p in net.named_parameters()}


def trajectory(mode, x, y, steps, B):
    _lib.TRAIN_F16X3 = mode == "f16x3"
    _, net = make()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    losses = []
    for i in range(steps):
        idx = torch.arange(i * B, (i + 1) * B, device="cuda") % x.shape[0]
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(net(x[idx])[:, 0], y[idx])
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses


def main():
    B = 10000
    env, _ = make()
    states, nb, _ = _lib.generate_states(env._env_id, env._dim, B * 4, 0, 30, 5, 0)
    x = _lib.nnet_input(env._env_id, env._dim, states)
    y = nb.float().contiguous()
    l_r, g_r = grads("float64", x[:B], y[:B])
    res = {"one_step_loss_float64": l_r}
    import re
    noise = re.compile(r"(fc1|fc2|blocks\.\d\.[02])\.bias")  # a Linear bias in front of BatchNorm: analytically zero gradient
    for mode in ("f16x3", "library"):  # (f16x3: forward + input-gradient GEMMs on dca_f16x3_gemm, weight gradient on the library)
        l_a, g_a = grads(mode, x[:B], y[:B])
        err = {k: float((g_a[k] - g_r[k]).abs().max()) / max(float(g_r[k].abs().max()), 1e-300) for k in g_a if not noise.fullmatch(k)}
        top = sorted(err.items(), key=lambda kv: -kv[1])[:3]
        res[mode] = {"loss": l_a, "grad_error_vs_float64_over_max_grad": {"worst": top, "median": float(np.median(list(err.values()))),
                                                                         "mean": float(np.mean(list(err.values())))}}
    print(json.dumps(res))
    ta, tb = trajectory("f16x3", x, y, 30, B), trajectory("library", x, y, 30, B)
    tb2 = trajectory("library", x, y, 30, B)
    print(json.dumps({"loss_f16x3": [round(v, 5) for v in ta], "loss_library": [round(v, 5) for v in tb],
                      "library_run_to_run_identical": ta is not None and tb == tb2}))


if __name__ == "__main__":
    main()
