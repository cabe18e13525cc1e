#!/usr/bin/env python3
"""Golden fixture for the training step, made by IMPORTING the reference (build container only):

    python tests/golden/make_golden_train.py      -> tests/golden/train_nnet.npz

Calls the reference's own `utils/nnet_utils.py:train_nnet` (Adam, MSE, lr*lr_d^itr, make_batches + shuffle) on
the tiny ResnetModel whose weights are in tiny_resnet.npz, CPU, fixed numpy / random seeds, and records the
inputs, the returned last loss and the final state dict (weights, BN running stats).  Data only.
"""
import os
import random
import sys

import numpy as np

np.float = float  # noqa: the reference targets numpy 1.22
np.int = int  # noqa

sys.path.insert(0, "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402
from utils import nnet_utils  # noqa: E402
from utils.pytorch_models import ResnetModel  # noqa: E402

torch.set_num_threads(1)
tiny = np.load(os.path.join(OUT, "tiny_resnet.npz"))
out = {}
for tag, bn, n, bs, itrs, itr0, lr, lr_d in (("bn", True, 50, 16, 5, 3, 0.01, 0.9), ("nobn", False, 40, 8, 7, 0, 0.005, 0.99)):
    torch.manual_seed(5)
    net = ResnetModel(54, 6, 64, 32, 2, 1, bn)
    if bn:
        net.load_state_dict({k[2:]: torch.tensor(tiny[k]) for k in tiny.files if k.startswith("w:")})
    rng = np.random.default_rng(11)
    x = rng.integers(0, 6, size=(n, 54)).astype(np.uint8)
    y = (rng.random((n, 1)) * 10).astype(np.float64)
    for k, v in net.state_dict().items():
        out["%s:init:%s" % (tag, k)] = v.numpy().copy()
    np.random.seed(7)
    random.seed(7)
    last = nnet_utils.train_nnet(net, [x], y, torch.device("cpu"), bs, itrs, itr0, lr, lr_d, display=False)
    out["%s:x" % tag], out["%s:y" % tag] = x, y
    out["%s:args" % tag] = np.array([bs, itrs, itr0, lr, lr_d], np.float64)
    out["%s:last_loss" % tag] = np.array(last, np.float64)
    for k, v in net.state_dict().items():
        out["%s:final:%s" % (tag, k)] = v.numpy().copy()
    print(tag, "last loss", last)
np.savez_compressed(os.path.join(OUT, "train_nnet.npz"), **out)
print("wrote train_nnet.npz:", len(out), "arrays")
