"""Cost-to-go network of the BWAS path: the residual MLP of the reference's
`utils/pytorch_models.py:5-86`, with the SAME state-dict layout (fc1/bn1/fc2/bn2/blocks.{b}.{0..3}/fc_out)
so reference checkpoints load unchanged.  Dense layers run on PyTorch-ROCm (rocBLAS/hipBLASLt → MFMA).

`ResnetModel.forward` accepts the uint8 network input (colour index / tiles) exactly like the
reference; `forward_onehot` accepts the one-hot rows the fused HIP expansion kernel already wrote, so
the int64 `F.one_hot` intermediate of pytorch_models.py:49-52 never exists on the device.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _dense_bn(in_dim: int, out_dim: int, batch_norm: bool):
    layers = [nn.Linear(in_dim, out_dim)]
    if batch_norm:
        layers.append(nn.BatchNorm1d(out_dim))
    return layers


class ResnetModel(nn.Module):
    def __init__(self, state_dim: int, one_hot_depth: int, h1_dim: int, resnet_dim: int, num_resnet_blocks: int,
                 out_dim: int, batch_norm: bool):
        super().__init__()
        self.one_hot_depth = one_hot_depth
        self.state_dim = state_dim
        self.num_resnet_blocks = num_resnet_blocks
        self.batch_norm = batch_norm
        # registration order fixes the state-dict key order: blocks first (pytorch_models.py:12)
        self.blocks = nn.ModuleList()
        in_dim = state_dim * one_hot_depth if one_hot_depth > 0 else state_dim
        stem1 = _dense_bn(in_dim, h1_dim, batch_norm)
        stem2 = _dense_bn(h1_dim, resnet_dim, batch_norm)
        self.fc1 = stem1[0]
        if batch_norm:
            self.bn1 = stem1[1]
        self.fc2 = stem2[0]
        if batch_norm:
            self.bn2 = stem2[1]
        for _ in range(num_resnet_blocks):
            self.blocks.append(nn.ModuleList(_dense_bn(resnet_dim, resnet_dim, batch_norm)
                                             + _dense_bn(resnet_dim, resnet_dim, batch_norm)))
        self.fc_out = nn.Linear(resnet_dim, out_dim)

    # -- pieces -------------------------------------------------------------------------------
    def encode(self, states_nnet: torch.Tensor) -> torch.Tensor:
        """uint8 [M, state_dim] -> float32 [M, state_dim*depth] (pytorch_models.py:49-52)."""
        if self.one_hot_depth <= 0:
            return states_nnet.float()
        if states_nnet.is_cuda and states_nnet.dtype == torch.uint8:
            from .. import _lib  # fused device encoder, no int64 intermediate
            return _lib.onehot(states_nnet, self.one_hot_depth, torch.float32)
        x = torch.nn.functional.one_hot(states_nnet.long(), self.one_hot_depth).float()
        return x.view(-1, self.state_dim * self.one_hot_depth)

    def trunk(self, x: torch.Tensor) -> torch.Tensor:
        bn = self.batch_norm
        x = self.fc1(x)
        x = torch.relu(self.bn1(x) if bn else x)
        x = self.fc2(x)
        x = torch.relu(self.bn2(x) if bn else x)
        for blk in self.blocks:
            skip = x
            if bn:
                x = torch.relu(blk[1](blk[0](x)))
                x = blk[3](blk[2](x))
            else:
                x = blk[1](torch.relu(blk[0](x)))
            x = torch.relu(x + skip)
        return self.fc_out(x)

    def forward(self, states_nnet: torch.Tensor) -> torch.Tensor:
        return self.trunk(self.encode(states_nnet))

    def forward_onehot(self, onehot_rows: torch.Tensor) -> torch.Tensor:
        return self.trunk(onehot_rows)


def fold_batchnorm(model: ResnetModel) -> ResnetModel:
    """Return an eval-only copy with every BatchNorm1d folded into the preceding Linear
    (y = (Wx+b-mean)/sqrt(var+eps)*gamma+beta).  Same outputs up to fp32 rounding; removes 10
    elementwise passes over [M,1000..5000] activations per forward."""
    import copy
    m = copy.deepcopy(model).eval()
    if not m.batch_norm:
        return m

    def fold(lin: nn.Linear, bn: nn.BatchNorm1d):
        with torch.no_grad():
            s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            lin.weight.copy_((lin.weight.double() * s[:, None]).float())
            lin.bias.copy_(((lin.bias.double() - bn.running_mean.double()) * s + bn.bias.double()).float())

    fold(m.fc1, m.bn1)
    fold(m.fc2, m.bn2)
    m.bn1 = nn.Identity()
    m.bn2 = nn.Identity()
    for blk in m.blocks:
        fold(blk[0], blk[1])
        fold(blk[2], blk[3])
        blk[1] = nn.Identity()
        blk[3] = nn.Identity()
    return m
