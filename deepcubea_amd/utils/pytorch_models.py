"""Cost-to-go network of the BWAS path: the residual MLP of the reference's
`utils/pytorch_models.py:5-86`, with the SAME state-dict layout (fc1/bn1/fc2/bn2/blocks.{b}.{0..3}/fc_out)
so reference checkpoints load unchanged.  Dense layers run on PyTorch-ROCm (rocBLAS/hipBLASLt → MFMA).

`ResnetModel.forward` accepts the uint8 network input (colour index / tiles) exactly like the
reference; `forward_onehot` accepts the one-hot rows the fused HIP expansion kernel already wrote, so
the int64 `F.one_hot` intermediate of pytorch_models.py:49-52 never exists on the device.
"""
from __future__ import annotations

from typing import List, Optional

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

    def _trunk_train_dev(self, x: torch.Tensor) -> torch.Tensor:
        """Training mode on the HIP device: the BatchNorms (+ residual add + ReLU) run through the library's column
        reduction kernels (csrc/dca_train.hip) — same arithmetic as `trunk`, the framework's 2-D BatchNorm kernels
        were 41 % of the training step — and the Linears' forward / input-gradient GEMMs through dca_f16x3_gemm
        (`_lib.linear_train`)."""
        from .. import _lib
        lin = _lib.linear_train  # forward + input gradient on dca_f16x3_gemm (fp32-accurate), weight gradient on the library
        x = _lib.bn_train(lin(x, self.fc1), self.bn1, relu=True)
        x = _lib.bn_train(lin(x, self.fc2), self.bn2, relu=True)
        for blk in self.blocks:
            h = _lib.bn_train(lin(x, blk[0]), blk[1], relu=True)
            x = _lib.bn_train(lin(h, blk[2]), blk[3], relu=True, skip=x)
        return self.fc_out(x)

    def trunk(self, x: torch.Tensor) -> torch.Tensor:
        bn = self.batch_norm
        if bn and self.training and x.is_cuda and x.dtype == torch.float32 and isinstance(self.bn1, nn.BatchNorm1d):
            return self._trunk_train_dev(x)
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


def _pad_dim(n: int, spare: int) -> int:
    """Width a layer is padded to: a multiple of 256 for the wide layers (64 for narrow ones) — the library's MFMA
    macro-tiles divide these evenly, 1000 -> 1024 nearly doubles the bf16 GEMM rate on gfx950 — with `spare` free units."""
    q = 256 if n >= 512 else 64
    return ((n + spare + q - 1) // q) * q


def l1_weight_tiles(w: torch.Tensor, planes: int, k_pad: int) -> torch.Tensor:
    """fp32 weight matrix [n_pad, K] (out, in) -> the bf16 plane tiles `dca_l1_onehot_gemm` stages into LDS:
    W = hi + mid + lo with every plane bf16 (8 + 8 + 8 mantissa bits: the sum is W to fp32 precision), zero padded along K
    to k_pad, laid out [n_pad/64][planes][k_pad/8][64][8]."""
    n_pad, k = w.shape
    assert n_pad % 64 == 0 and k_pad % 8 == 0 and k_pad >= k
    r = torch.zeros((n_pad, k_pad), dtype=torch.float32)
    r[:, :k] = w.detach().float().cpu()
    ps = []
    for _ in range(planes):
        p = r.to(torch.bfloat16)
        ps.append(p)
        r = r - p.float()
    t = torch.stack(ps)  # [P, n_pad, k_pad]
    t = t.view(planes, n_pad // 64, 64, k_pad // 8, 8).permute(1, 0, 3, 2, 4).contiguous()
    return t


def l1_weight_tiles8(w8: torch.Tensor, k_pad: int) -> torch.Tensor:
    """e4m3 weight matrix [n_pad, K] (out, in; already divided by its per-unit scales) -> the byte tiles `dca_l1_onehot_gemm8`
    stages into LDS: K zero padded to k_pad (a multiple of 64), laid out [n_pad/128][k_pad/16][128][16]."""
    n_pad, k = w8.shape
    assert n_pad % 128 == 0 and k_pad % 64 == 0 and k_pad >= k and w8.dtype == torch.float8_e4m3fn
    r = torch.zeros((n_pad, k_pad), dtype=torch.uint8)
    r[:, :k] = w8.detach().cpu().view(torch.uint8)
    return r.view(n_pad // 128, 128, k_pad // 16, 16).permute(0, 2, 1, 3).contiguous()


def _pow2_scale(w: torch.Tensor) -> torch.Tensor:
    """Per output unit (row of w) the power of two that brings max|row| to ~1024, so that the fp16 low halves of the
    scaled weights stay normal numbers whatever the spread of magnitudes ACROSS units (BatchNorm folding can make it large)."""
    amax = w.abs().amax(dim=1)
    sc = torch.exp2(torch.floor(torch.log2(1024.0 / amax.clamp_min(1e-30))))
    return torch.where(amax > 0, sc, torch.ones_like(sc)).clamp(2.0 ** -100, 2.0 ** 100)


def _split_f16(w: torch.Tensor, sc: torch.Tensor):
    ws = w * sc[:, None]
    wh = ws.to(torch.float16)
    return wh, (ws - wh.float()).to(torch.float16)


# One-hot depth from which layer 1 runs as the embedding sum (dca_l1_embed) instead of the one-hot MFMA kernel, by mode.  Measured,
# ms per 409 600 rows x 5120 units, MFMA kernel / embedding sum (profiles/r06_l1_embed_bench.txt):
#   fp32 (planes out): cube3 4.0 / 6.0, puzzle15 3.0 / 2.1, puzzle24 9.2 / 3.7, puzzle35 17.3 / 5.2, puzzle48 30.1 / 6.5
#   bf16:              cube3 1.7 / 5.5, puzzle15 1.3 / 1.8, puzzle24 4.8 / 3.1, puzzle35 9.4 / 4.7, puzzle48 16.0 / 6.4
L1_EMBED_MIN_DEPTH = {torch.float32: 16, torch.bfloat16: 25, torch.float16: 1 << 30}


class FastResnet(nn.Module):
    """Inference-only re-layout of a `ResnetModel` (pytorch_models.py:5-86 of the reference), same function:

      * BatchNorm folded into the Linears (eval statistics);
      * every width padded (`_pad_dim`): padded units have zero weights and zero bias, so they stay 0 through ReLU and
        contribute nothing — outputs equal the unpadded network's up to fp summation order;
      * bias + ReLU ride in the GEMM epilogue (`torch._addmm_activation`);
      * in a residual block the skip connection is the GEMM's C operand, and the second Linear's bias is folded into
        its weight matrix through a constant-one hidden unit (the first padded unit of the block's hidden layer has
        zero weights and bias 1), leaving one in-place ReLU pass per block as the only elementwise kernel;
      * layer 1 runs as the library's hand-written one-hot MFMA kernel (`csrc/dca_mlp.hip`, `forward` on uint8 rows) where
        its geometry is instantiated: no one-hot matrix, fp32-exact through three bf16 weight planes;
      * dtype float32 on the device (`split=True`, the default): every other dense layer is ONE f16 GEMM with fp32 output
        over split operands ("f16x3", `dca_act_split`) — fp32 accuracy at 2.4-2.9x the speed of the library's fp32 GEMM.
        `split=False` keeps the plain fp32 GEMMs (what the host path always uses).

    Input: uint8 network inputs `[M, state_dim]` through `forward` (`uses_l1_kernel`: feed the engine's packed
    network-input rows), or one-hot rows `[M, in_pad]` in `dtype` (row stride `in_pad` >= state_dim*depth, tail zero) as
    written by the engine's pack kernel through `forward_onehot`.  fp32 is the 1e-5 parity mode."""

    def __init__(self, model: ResnetModel, dtype: torch.dtype = torch.float32, split: bool = True, gemm: str = "hip",
                 gemm16: str = "hip", l1: str = "auto"):
        super().__init__()
        assert l1 in ("auto", "mfma", "embed")
        m = fold_batchnorm(model)
        self.state_dim, self.one_hot_depth = m.state_dim, m.one_hot_depth
        self.dtype = dtype
        in_dim = m.fc1.in_features
        self.in_dim = in_dim
        self.in_pad = ((in_dim + 63) // 64) * 64
        h1, r = m.fc1.out_features, m.fc2.out_features
        h1p, rp = _pad_dim(h1, 0), _pad_dim(r, 1)
        self.res_dim, self.res_pad = r, rp

        def padw(lin: nn.Linear, outp: int, inp: int):
            w = torch.zeros(outp, inp, dtype=torch.float32)
            b = torch.zeros(outp, dtype=torch.float32)
            w[:lin.out_features, :lin.in_features] = lin.weight.detach().float().cpu()
            b[:lin.out_features] = lin.bias.detach().float().cpu()
            return w, b

        ws, bs = [], []
        w, b = padw(m.fc1, h1p, self.in_pad)
        ws.append(w), bs.append(b)
        w, b = padw(m.fc2, rp, h1p)
        ws.append(w), bs.append(b)
        raw = [(w, b)]  # dense layers after the first, before the bias-folding trick (the f16x3 path adds biases itself)
        for blk in m.blocks:
            wa, ba = padw(blk[0], rp, rp)
            wb, bb = padw(blk[2] if len(blk) == 4 else blk[1], rp, rp)
            raw += [(wa.clone(), ba.clone()), (wb.clone(), bb.clone())]
            ba[r] = 1.0  # constant-one unit feeding the next Linear's folded bias
            wb[:, r] = bb
            ws += [wa, wb]
            bs += [ba, torch.zeros(0)]
        wo, bo = padw(m.fc_out, m.fc_out.out_features, rp)
        self.weights = nn.ParameterList([nn.Parameter(w.to(dtype), requires_grad=False) for w in ws])
        self.biases = nn.ParameterList([nn.Parameter(b.to(dtype), requires_grad=False) for b in bs])
        # fp32 copies of the biases for the 16-bit kernels' epilogues (the bias is added to the fp32 accumulator there)
        self.biases_f32 = nn.ParameterList([nn.Parameter(b.to(dtype).float(), requires_grad=False) for b in bs])
        self.w_out = nn.Parameter(wo.to(dtype), requires_grad=False)
        self.b_out = nn.Parameter(bo.float(), requires_grad=False)
        # the output layer runs as dca_head_gemv on the device (fixed summation order): fp32 copy of the weights in `dtype`
        self.w_out_f32 = nn.Parameter(wo.to(dtype).float().contiguous(), requires_grad=False)
        # fp32 mode on the device: every dense layer after the first as ONE f16 GEMM with fp32 output over the split
        # operands A3[3k..3k+2] = (xh, xl, xh), W3[3k..3k+2] = (wh, wh, wl) (csrc/dca_mlp.hip k_act_split): fp32-accurate, 2.4-2.9x faster
        # than the library's fp32 GEMM.  Weights are pre-scaled by a power of two so their low halves stay normal numbers.
        self.split = bool(split) and dtype == torch.float32
        # "hip": every dense layer after the first is ONE launch of the hand-written f16x3 kernel (csrc/dca_gemm.hip: operand
        # planes, the three products per K-step, layer tail in the epilogue); "library": round 1's arrangement — one library
        # f16 GEMM over the 3x-wide interleaved operand plus the dca_act_split glue kernel per layer (kept for comparison)
        self.gemm = gemm
        assert gemm in ("hip", "library")
        # bf16 / fp16 (non-parity) modes.  "hip" (default since round 4: the product's own kernels) = one dca_gemm16 launch per
        # layer, whole tail (bias, residual add, ReLU, rounding) in the epilogue (csrc/dca_gemm16.hip); "library" = hipBLASLt
        # GEMMs with fused bias+ReLU, plus one ReLU pass per residual block — selectable from the CLI (`--gemm16 library`).
        # Per layer at 204 800 rows, candidates taking turns (profiles/r05_gemm_bench.txt, ms, hip / library): 1024->1024
        # residual+ReLU 0.50 / 0.56 (ahead), 1024->1024 bias+ReLU 0.45 / 0.38, 5120->1024 bias+ReLU 1.93 / 1.67 (behind).
        # What bounds both is the operand stream from L2 into the LDS (64 KB per K-tile of a 256 x 256 tile, ~15-20 B/clk/CU
        # with every CU fetching) and, on random operands, the power limit; csrc/dca_gemm16.hip has the measurements.
        self.gemm16 = gemm16
        assert gemm16 in ("hip", "library")
        # set by the split kernels when a value does not fit fp16 (|v| > 60000): that batch is redone with fp32 GEMMs
        self.register_buffer("_overflow", torch.zeros(1, dtype=torch.int32), persistent=False)
        self.split_fallbacks = 0
        self.split_w = nn.ParameterList()
        self.split_wh = nn.ParameterList()
        self.split_wl = nn.ParameterList()
        self.split_b = nn.ParameterList()
        self.split_alpha = nn.ParameterList()  # per-output-unit 1/scale vectors
        self.l1_split_w = nn.ParameterList()
        self.l1_split_alpha = None
        if self.split:
            # layer 1 on materialised one-hot rows (geometries without the MFMA kernel): the rows are exact in fp16, so two
            # fp16 weight planes (22 bits) and two f16 GEMMs with fp32 output give the fp32 layer
            sc = _pow2_scale(ws[0])
            w1h, w1l = _split_f16(ws[0], sc)
            self.l1_split_w.append(nn.Parameter(w1h, requires_grad=False))
            self.l1_split_w.append(nn.Parameter(w1l, requires_grad=False))
            self.l1_split_alpha = nn.Parameter(1.0 / sc, requires_grad=False)
            for w, b in raw:
                sc = _pow2_scale(w)
                wh, wl = _split_f16(w, sc)
                # only the operand layout the selected mode reads is kept (each is a full fp16 copy or three of the weights)
                if gemm == "library":
                    self.split_w.append(nn.Parameter(torch.stack([wh, wh, wl], dim=2).reshape(w.shape[0], -1).contiguous(),
                                                     requires_grad=False))  # W3[:, 3k..3k+2] = (wh, wh, wl)
                else:
                    self.split_wh.append(nn.Parameter(wh.contiguous(), requires_grad=False))  # planes for dca_f16x3_gemm
                    self.split_wl.append(nn.Parameter(wl.contiguous(), requires_grad=False))
                self.split_b.append(nn.Parameter(b.clone(), requires_grad=False))
                self.split_alpha.append(nn.Parameter(1.0 / sc, requires_grad=False))
        # layer 1 straight from the uint8 rows (csrc/dca_mlp.hip) where the geometry is instantiated: fp32 weights as
        # three bf16 planes (exact), fp16 as two, bf16 as one
        self.l1_planes = {torch.float32: 3, torch.float16: 2, torch.bfloat16: 1}[dtype]
        self.l1_tiles = None
        self.l1_bias = None
        self.l1_embed_w = None
        # measured at 204 800 rows x 5120 units: fp32 2.30 ms vs 5.89 ms for the library's fp32 GEMM; bf16 on par with the
        # library (and no one-hot rows to write/read); two-plane fp16 is slower than the library's f16 GEMM -> not used
        if self.one_hot_depth > 0 and dtype != torch.float16:
            try:
                from .. import _lib
                ok = _lib.l1_supported(self.state_dim, self.one_hot_depth)
                kpad = _lib.l1_kpad(self.state_dim, self.one_hot_depth) if ok else 0
                emb_ok = _lib.l1_embed_supported(self.state_dim, self.one_hot_depth)
            except Exception:  # library not built (host-only use of the module)
                ok = emb_ok = False
            # the same layer as an embedding sum on the vector pipes (csrc/dca_embed.hip: one gathered fp32 weight per position
            # instead of `depth` multiply-adds, exact fp32 arithmetic): ahead of the MFMA kernel where the one-hot depth is
            # large — the sliding puzzles (see L1_EMBED_MIN_DEPTH); cube3 (depth 6) stays on the matrix pipes
            if l1 == "embed" and not emb_ok:
                raise ValueError("FastResnet(l1='embed'): dca_l1_embed is not available for geometry (%d, %d)"
                                 % (self.state_dim, self.one_hot_depth))
            # (lightsout7 has no one-hot MFMA instantiation in these modes: its fp32 forward is faster with the embedding sum than
            # with layer 1 on materialised one-hot rows — 31.7 vs 38.6 ms per 409 600 rows —, its bf16 forward is not: 15.2 vs 12.0)
            use_emb = emb_ok and (l1 == "embed" or (l1 == "auto" and (
                (ok and self.one_hot_depth >= L1_EMBED_MIN_DEPTH[dtype]) or (not ok and dtype == torch.float32))))
            if ok or use_emb:
                w1 = ws[0][:, :in_dim] if dtype == torch.float32 else ws[0][:, :in_dim].to(dtype).float()
                self.l1_bias = nn.Parameter(bs[0].to(dtype).float(), requires_grad=False)
            if ok:
                self.l1_tiles = nn.Parameter(l1_weight_tiles(w1, self.l1_planes, kpad), requires_grad=False)
            if use_emb:
                self.l1_embed_w = nn.Parameter(w1.t().contiguous(), requires_grad=False)  # [K, h1_pad] fp32 (bf16 mode: bf16-rounded values)

    @property
    def uses_l1_kernel(self) -> bool:
        return self.l1_tiles is not None or self.l1_embed_w is not None

    def _head(self, x: torch.Tensor) -> torch.Tensor:
        """fc_out (pytorch_models.py:83-86 of the reference): [M, res_pad] -> [M, out_dim] float32.  On the device: the
        library's own streaming kernel (dca_head_gemv) — a row's value then has the same bits whatever batch it sits in."""
        if x.is_cuda and self.w_out_f32.shape[0] <= 8:
            from .. import _lib
            return _lib.head_gemv(x, self.w_out_f32, self.b_out)
        return (x @ self.w_out.t()).float() + self.b_out

    @property
    def onehot_dtype(self) -> torch.dtype:
        """Element type of the one-hot rows `forward_onehot` wants on the device (0/1 are exact in every type)."""
        return torch.float16 if self.split else self.dtype

    @torch.no_grad()
    def forward_onehot(self, x: torch.Tensor) -> torch.Tensor:
        """[M, in_pad] one-hot rows (`onehot_dtype` on the device, self.dtype on the host) -> [M, out_dim] float32."""
        W, B = self.weights, self.biases
        if self.split and x.is_cuda:
            from .. import _lib
            x = x if x.dtype == torch.float16 else x.to(torch.float16)
            self._overflow.zero_()
            y = torch.mm(x, self.l1_split_w[0].t(), out_dtype=torch.float32)
            y.add_(torch.mm(x, self.l1_split_w[1].t(), out_dtype=torch.float32))
            a3, _ = _lib.act_split(y, B[0], None, self.l1_split_alpha, True, False,
                                   want_a3="planes" if self.gemm == "hip" else True, overflow=self._overflow)
            out = self._after_l1_planes(a3) if self.gemm == "hip" else self._after_l1_split(a3)
            if int(self._overflow.item()) == 0:
                return out
            self.split_fallbacks += 1
            x = x.float()
        return self._after_l1(torch._addmm_activation(B[0], x, W[0].t()))

    def _after_l1(self, x: torch.Tensor) -> torch.Tensor:
        """bf16 / fp16 on the device with `gemm16="hip"`: one dca_gemm16 launch per dense layer — bias, residual add, ReLU
        and the rounding to 16 bits in its epilogue (csrc/dca_gemm16.hip).  Otherwise (the default for the 16-bit modes, fp32
        without the split, the host): library GEMMs with fused epilogues."""
        W, B = self.weights, self.biases
        if x.is_cuda and self.gemm16 == "hip" and self.dtype in (torch.bfloat16, torch.float16) and x.dtype == self.dtype:
            from .. import _lib
            Bf = self.biases_f32
            x = _lib.gemm16(x.contiguous(), W[1], Bf[1], None, True)
            for k in range(2, len(W), 2):
                h = _lib.gemm16(x, W[k], Bf[k], None, True)
                x = _lib.gemm16(h, W[k + 1], None, x, True, out=x)  # (the block's second bias rides in W through h's constant-one unit)
            return self._head(x)
        x = torch._addmm_activation(B[1], x, W[1].t())
        for k in range(2, len(W), 2):
            h = torch._addmm_activation(B[k], x, W[k].t())
            x = x.addmm_(h, W[k + 1].t()).relu_()  # in place: the skip is the GEMM's C operand, no copy of it
        return self._head(x)  # fc_out: [M,rp] x [rp,out_dim], fp32 bias add

    @torch.no_grad()
    def encode(self, states_nnet: torch.Tensor) -> torch.Tensor:
        from .. import _lib
        dt = self.onehot_dtype if states_nnet.is_cuda else self.dtype
        oh = _lib.onehot(states_nnet, self.one_hot_depth, dt) if self.one_hot_depth > 0 else states_nnet.to(dt)
        return torch.nn.functional.pad(oh, (0, self.in_pad - oh.shape[1]))

    @torch.no_grad()
    def forward(self, states_nnet: torch.Tensor) -> torch.Tensor:
        """uint8 network inputs [M, state_dim] -> [M, out_dim] float32."""
        emb = self.l1_embed_w if (self.gemm == "hip" or not self.split) else None  # (the library-GEMM f16x3 operand: MFMA kernel only)
        if (self.l1_tiles is None and emb is None) or not states_nnet.is_cuda:
            return self.forward_onehot(self.encode(states_nnet))
        from .. import _lib
        if self.split:  # the layer-1 kernel's epilogue writes the next layer's split operand directly
            self._overflow.zero_()
            if emb is not None:
                a3 = _lib.l1_embed(states_nnet, self.one_hot_depth, emb, self.l1_bias, True, split="planes", overflow=self._overflow)
            else:
                a3 = _lib.l1_onehot_gemm(states_nnet, self.one_hot_depth, self.l1_tiles, self.l1_planes, self.l1_bias, True,
                                         self.dtype, split="planes" if self.gemm == "hip" else True, overflow=self._overflow)
            out = self._after_l1_planes(a3) if self.gemm == "hip" else self._after_l1_split(a3)
            if int(self._overflow.item()) == 0:
                return out
            self.split_fallbacks += 1  # some activation beyond fp16 range: same batch again with fp32 GEMMs
        if emb is not None:
            x = _lib.l1_embed(states_nnet, self.one_hot_depth, emb, self.l1_bias, True, self.dtype)
        else:
            x = _lib.l1_onehot_gemm(states_nnet, self.one_hot_depth, self.l1_tiles, self.l1_planes, self.l1_bias, True,
                                    self.dtype)
        return self._after_l1(x)

    def _after_l1_split(self, a3: torch.Tensor) -> torch.Tensor:
        """split operand of relu(layer 1) [M, 3*h1_pad] fp16 -> [M, out_dim]; the f16x3 path (see __init__)."""
        from .. import _lib
        W, B, A, ovf = self.split_w, self.split_b, self.split_alpha, self._overflow
        f32 = torch.float32
        y = torch.mm(a3, W[0].t(), out_dtype=f32)
        nblk = (len(W) - 1) // 2
        a3, x = _lib.act_split(y, B[0], None, A[0], True, True, want_a3=nblk > 0, overflow=ovf)
        for blk in range(nblk):
            ka, kb = 1 + 2 * blk, 2 + 2 * blk
            y = torch.mm(a3, W[ka].t(), out_dtype=f32)
            ah, _ = _lib.act_split(y, B[ka], None, A[ka], True, False, overflow=ovf)
            y = torch.mm(ah, W[kb].t(), out_dtype=f32)
            a3, x = _lib.act_split(y, B[kb], x, A[kb], True, True, want_a3=blk + 1 < nblk, overflow=ovf)
        return self._head(x)

    def _after_l1_planes(self, planes: torch.Tensor) -> torch.Tensor:
        """fp16 planes of relu(layer 1) [2, M, h1_pad] -> [M, out_dim]: one dca_f16x3_gemm launch per dense layer (scale,
        bias, residual add, ReLU and the split of the result into the next layer's planes ride in its epilogue)."""
        from .. import _lib
        Wh, Wl, B, A, ovf = self.split_wh, self.split_wl, self.split_b, self.split_alpha, self._overflow
        nblk = (len(Wh) - 1) // 2
        planes, x = _lib.f16x3_gemm(planes, Wh[0], Wl[0], A[0], 1.0, B[0], None, True, nblk > 0, True, ovf)
        for blk in range(nblk):
            ka, kb = 1 + 2 * blk, 2 + 2 * blk
            ph, _ = _lib.f16x3_gemm(planes, Wh[ka], Wl[ka], A[ka], 1.0, B[ka], None, True, True, False, ovf)
            planes, x = _lib.f16x3_gemm(ph, Wh[kb], Wl[kb], A[kb], 1.0, B[kb], x, True, blk + 1 < nblk, True, ovf)
        return self._head(x)


class _Fp8BlockMixin:
    """The block-scaled forward of Fp8Resnet (kept apart for readability)."""

    @torch.no_grad()
    def _forward_block_scaled(self, states_nnet: torch.Tensor) -> torch.Tensor:
        from .. import _lib
        b = self.base
        h8, hs = _lib.l1_onehot_gemm_mx(states_nnet, self.one_hot_depth, b.l1_tiles, b.l1_bias, True)
        x16, x8, xs = _lib.gemm8_mx(h8, hs, self.w8[0], self.w_scale[0], self.bias[0], None, True, True, True)
        nblk = (len(self.w8) - 1) // 2
        for i in range(nblk):
            ja, jb = 1 + 2 * i, 2 + 2 * i
            _, h8, hs = _lib.gemm8_mx(x8, xs, self.w8[ja], self.w_scale[ja], self.bias[ja], None, True, False, True)
            last = i == nblk - 1  # the last block's output only feeds the output layer (bf16 stream)
            x16, x8, xs = _lib.gemm8_mx(h8, hs, self.w8[jb], self.w_scale[jb], self.bias[jb], x16, True, True, not last, out16=x16)
        return b._head(x16)


class Fp8Resnet(_Fp8BlockMixin, nn.Module):
    """The same network (pytorch_models.py:5-86 of the reference, BatchNorm folded, widths padded as in `FastResnet`) evaluated
    at fp8 operand precision on the device: a NON-parity speed mode (`--nnet_dtype fp8`), twice the matrix rate of bf16.

      * layer 1: the one-hot MFMA kernel (`dca_l1_onehot_gemm`, one bf16 weight plane) writing its output already QUANTISED
        to OCP e4m3 (the activation scale is folded into the layer's weights and bias);
      * every other dense layer: ONE `dca_gemm8` launch (csrc/dca_gemm8.hip) — e4m3 operands, fp32 accumulation on
        `v_mfma_f32_32x32x64_f8f6f4`, and dequantisation + bias + residual add + ReLU + quantisation of the result for the
        next layer in the epilogue.  The residual stream stays bf16; weights carry one scale per output unit, activations one
        scale per tensor;
      * the activation scales are calibrated once, on the first batch of at least 1024 rows this module sees (amax of every
        intermediate tensor in a bf16 evaluation of at most 4096 of its rows, 25 % headroom, saturating conversion), or
        explicitly (`calibrate`); thinner batches before that (a search's root) are evaluated in bf16.

    Same call interface as `FastResnet` (uint8 network inputs through `forward`).  Needs the layer-1 kernel's geometry
    (`dca_l1_supported`) and a GPU; there is no host path and no one-hot-row path."""

    E4M3_MAX = 448.0
    HEADROOM = 1.25
    MIN_CALIB_ROWS = 1024

    def __init__(self, model: ResnetModel, scaling: str = "tensor", l1: str = "auto"):
        """l1: "fp8" = layer 1 on the f8f6f4 pipe with e4m3 weights (dca_l1_onehot_gemm8; per-tensor scaling only), "bf16" = the
        round-5 arrangement (one bf16 weight plane on the bf16 pipe, output rounded to e4m3), "auto" = fp8 where the geometry is
        instantiated (dca_l1_supported8) and the scaling is per tensor.
        scaling="tensor" (default, `--nnet_dtype fp8`): one static scale per activation tensor, calibrated on the first
        batch of >= 1024 REAL rows (see `forward`).  scaling="block" (`--nnet_dtype fp8mx`): activations carry one E8M0
        scale per row and 64 elements, computed in the epilogue that produces them and applied by the scaled MFMA
        (dca_gemm8_mx / dca_l1_onehot_gemm_mx) — nothing is calibrated, nothing is frozen, nothing saturates, whatever depth
        of the search the states come from.  Measured: both sit at e4m3's own precision floor on this
        network (max 8-10 % of max|h|, rms 2.3-2.6 %: the 3-bit mantissa, not the scaling, sets it — tools/
        fp8_precision_floor.py), and the block-scaled layers cost 9-22 % more time (the scale traffic and the scaled
        MFMA's operand), 3.5e6 against 4.1e6 nodes/s end to end: hence the default."""
        super().__init__()
        assert scaling in ("block", "tensor")
        self.scaling = scaling
        from .. import _lib
        self.base = FastResnet(model, torch.bfloat16, gemm16="hip")  # calibration path / thin batches; also owns the folded bf16 weights
        if not self.base.uses_l1_kernel:
            raise ValueError("Fp8Resnet needs the layer-1 one-hot kernel (dca_l1_supported) for this geometry")
        b = self.base
        self.state_dim, self.one_hot_depth, self.in_pad, self.in_dim = b.state_dim, b.one_hot_depth, b.in_pad, b.in_dim
        self.dtype = _lib.E4M3
        W = [w.detach().float().cpu() for w in b.weights]
        B = [x.detach().float().cpu() for x in b.biases_f32]
        r = b.res_dim
        # undo FastResnet's bias folding (the block's second bias rides through a constant-one hidden unit there; quantised to
        # e4m3 that unit would carry a 6 % error into every bias): explicit biases, added to the fp32 accumulator
        for k in range(2, len(W), 2):
            B[k + 1] = W[k + 1][:, r].clone()
            W[k + 1][:, r] = 0.0
            B[k][r] = 0.0
        self.w8, self.w_scale, self.bias = nn.ParameterList(), nn.ParameterList(), nn.ParameterList()
        for k in range(1, len(W)):
            sw = (W[k].abs().amax(dim=1) / self.E4M3_MAX).clamp_min(1e-30)
            self.w8.append(nn.Parameter((W[k] / sw[:, None]).to(_lib.E4M3), requires_grad=False))
            self.w_scale.append(nn.Parameter(sw.contiguous(), requires_grad=False))
            self.bias.append(nn.Parameter(B[k].contiguous(), requires_grad=False))
        self._w1 = W[0][:, :self.in_dim].contiguous()  # layer 1 is rebuilt with the activation scale folded in (calibrate)
        self._b1 = B[0].contiguous()
        self.l1_tiles8 = None
        self.l1_embed_w8 = None
        self.l1_bias8 = None
        assert l1 in ("auto", "fp8", "bf16")
        can8 = scaling == "tensor" and _lib.l1_supported8(self.state_dim, self.one_hot_depth) and self._w1.shape[0] % 128 == 0
        if l1 == "fp8" and not can8:
            raise ValueError("Fp8Resnet(l1='fp8') needs per-tensor scaling and a geometry dca_l1_supported8 instantiates")
        self.l1_fp8 = can8 and l1 != "bf16"
        self.l1_w8_tiles = None   # l1_fp8: e4m3 weight tiles, and scale[n] = w_scale[n] / act_scale[0] for the kernel's epilogue
        self.l1_scale8 = None
        self.act_scale: List[float] = []  # [h1, x after fc2, then (h, x) per residual block]
        self.layer_scale = None           # per dense layer: activation scale of its input x w_scale, on the device

    @property
    def uses_l1_kernel(self) -> bool:
        return True

    @property
    def onehot_dtype(self) -> torch.dtype:
        return torch.bfloat16

    @property
    def l1_planes(self) -> int:
        return 1

    def forward_onehot(self, x: torch.Tensor) -> torch.Tensor:
        raise RuntimeError("Fp8Resnet takes the engine's packed uint8 network-input rows (forward), not one-hot rows")

    @torch.no_grad()
    def calibrate(self, states_nnet: torch.Tensor) -> None:
        """Per-tensor activation scales from a bf16 evaluation of (at most 4096 of) these rows."""
        from .. import _lib
        b = self.base
        W, Bf = b.weights, b.biases_f32  # (the bf16 layers on the library's own kernel, like FastResnet(gemm16="hip"))
        x = _lib.l1_onehot_gemm(states_nnet[:4096].contiguous(), self.one_hot_depth, b.l1_tiles, b.l1_planes, b.l1_bias, True, b.dtype)
        amax = [float(x.float().abs().max())]
        x = _lib.gemm16(x.contiguous(), W[1], Bf[1], None, True)
        amax.append(float(x.float().abs().max()))
        for k in range(2, len(W), 2):
            h = _lib.gemm16(x, W[k], Bf[k], None, True)
            amax.append(float(h.float().abs().max()))
            x = _lib.gemm16(h, W[k + 1], None, x, True, out=x)
            amax.append(float(x.float().abs().max()))
        self.act_scale = [max(a, 1e-6) * self.HEADROOM / self.E4M3_MAX for a in amax]
        dev = states_nnet.device
        kpad = _lib.l1_kpad(self.state_dim, self.one_hot_depth)
        inv1 = 1.0 / self.act_scale[0]
        if self.l1_fp8:
            # one scale per output unit like every other e4m3 layer; the activation scale of h1 rides in the epilogue's factors
            sw = (self._w1.abs().amax(dim=1) / self.E4M3_MAX).clamp_min(1e-30)
            w8 = (self._w1 / sw[:, None]).to(_lib.E4M3)
            self.l1_w8_tiles = l1_weight_tiles8(w8, _lib.l1_kpad8(self.state_dim, self.one_hot_depth)).to(dev)
            self.l1_scale8 = (sw * inv1).float().contiguous().to(dev)
        else:
            w1 = (self._w1 * inv1).to(torch.bfloat16).float()
            self.l1_tiles8 = l1_weight_tiles(w1, 1, kpad).to(dev)
            # deep one-hot geometries (the larger sliding puzzles): the same weights through the embedding sum, like the bf16 mode
            self.l1_embed_w8 = w1.t().contiguous().to(dev) if b.l1_embed_w is not None else None
        self.l1_bias8 = (self._b1 * inv1).float().contiguous().to(dev)
        # input scale of dense layer j (0 = fc2): h1, then inside block i: x_i for the first Linear, h_i for the second
        s_in = [self.act_scale[0]]
        for i in range((len(self.w8) - 1) // 2):
            s_in += [self.act_scale[1 + 2 * i], self.act_scale[2 + 2 * i]]
        self.layer_scale = [(self.w_scale[j].to(dev) * s_in[j]).contiguous() for j in range(len(self.w8))]

    takes_valid_rows = True  # get_heuristic_fn_dev passes the engine's live row count (the batch is padded to 1024 rows)

    @torch.no_grad()
    def forward(self, states_nnet: torch.Tensor, valid_rows: Optional[int] = None) -> torch.Tensor:
        """uint8 network inputs [M, state_dim] (device) -> [M, out_dim] float32.  valid_rows: the first rows that hold real
        states (the engine pads its batches to 1024 rows with zero / stale rows): only those may calibrate the scales."""
        from .. import _lib
        if not states_nnet.is_cuda:
            raise RuntimeError("Fp8Resnet runs on the GPU only")
        if self.scaling == "block":
            return self._forward_block_scaled(states_nnet)
        if self.layer_scale is None:
            real = states_nnet.shape[0] if valid_rows is None else min(int(valid_rows), states_nnet.shape[0])
            if real < self.MIN_CALIB_ROWS:  # a search's root / first thin batches: bf16 until there is a sample of REAL rows
                return self.base(states_nnet)
            self.calibrate(states_nnet[:real])
        s = self.act_scale
        if self.l1_fp8:
            h8 = _lib.l1_onehot_gemm8(states_nnet, self.one_hot_depth, self.l1_w8_tiles, self.l1_scale8, self.l1_bias8, True)
        else:
            if self.l1_embed_w8 is not None:
                h8 = _lib.l1_embed(states_nnet, self.one_hot_depth, self.l1_embed_w8, self.l1_bias8, True, _lib.E4M3)
            else:
                h8 = _lib.l1_onehot_gemm(states_nnet, self.one_hot_depth, self.l1_tiles8, 1, self.l1_bias8, True, _lib.E4M3)
        x16, x8 = _lib.gemm8(h8, self.w8[0], self.layer_scale[0], self.bias[0], None, True, True, 1.0 / s[1])
        nblk = (len(self.w8) - 1) // 2
        for i in range(nblk):
            ja, jb = 1 + 2 * i, 2 + 2 * i
            _, h8 = _lib.gemm8(x8, self.w8[ja], self.layer_scale[ja], self.bias[ja], None, True, False, 1.0 / s[2 + 2 * i])
            nxt = None if i == nblk - 1 else 1.0 / s[3 + 2 * i]  # the last block's output only feeds the bf16 output layer
            x16, x8 = _lib.gemm8(h8, self.w8[jb], self.layer_scale[jb], self.bias[jb], x16, True, True, nxt, out16=x16)
        return self.base._head(x16)
