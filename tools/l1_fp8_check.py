#!/usr/bin/env python3
"""Diagnostic: dca_l1_onehot_gemm8 against float64 for one geometry, printing where bytes differ.
python tools/l1_fp8_check.py D depth n_pad m"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402
from deepcubea_amd.utils.pytorch_models import l1_weight_tiles8  # noqa: E402

E4M3 = torch.float8_e4m3fn
D, depth, n_pad, m = (int(v) for v in sys.argv[1:5])
g = torch.Generator().manual_seed(100 + D)
K = D * depth
w8 = (torch.randn(n_pad, K, generator=g) * 40.0 + torch.arange(K)[None, :] * 0.05 + torch.arange(n_pad)[:, None] * 0.01).to(E4M3)
scale = (torch.rand(n_pad, generator=g) * 3.0 + 0.5).float()
bias = (torch.randn(n_pad, generator=g) * 30.0).float()
tiles = l1_weight_tiles8(w8, _lib.l1_kpad8(D, depth)).cuda()
x = torch.randint(0, depth, (m, D), dtype=torch.uint8, generator=g)
y8 = _lib.l1_onehot_gemm8(x.cuda(), depth, tiles, scale.cuda(), bias.cuda(), True).cpu()
idx = (torch.arange(D)[None, :] * depth + x.long())
acc = w8.double().t()[idx].sum(dim=1)
v = torch.clamp(acc * scale.double()[None, :] + bias.double()[None, :], min=0.0)
want = v.float().clamp(-448.0, 448.0).to(E4M3)
bad = (y8.view(torch.uint8) != want.view(torch.uint8))
print("geometry", D, depth, n_pad, m, "mismatching bytes", int(bad.sum()), "of", bad.numel())
rows = bad.any(dim=1).nonzero().flatten()
cols = bad.any(dim=0).nonzero().flatten()
print("rows with a mismatch:", len(rows), rows[:40].tolist())
print("cols with a mismatch:", len(cols), cols[:40].tolist())
r, c = bad.nonzero()[:12].t() if bad.any() else (torch.tensor([], dtype=torch.long),) * 2
for i in range(len(r)):
    print("row %d col %d: got %s want %s v %.6f acc %.3f" % (int(r[i]), int(c[i]), float(y8[r[i], c[i]].float()), float(want[r[i], c[i]].float()),
                                                        float(v[r[i], c[i]]), float(acc[r[i], c[i]])))
# the implied accumulator of the kernel where ReLU did not clip and nothing saturated: (got - bias) / scale vs acc
