"""GPU tests of the fp8 (OCP e4m3) layer kernel (csrc/dca_gemm8.hip, `dca_gemm8`), the e4m3 output of the layer-1 kernel and
the fp8 evaluation of the whole network (`Fp8Resnet`, `--nnet_dtype fp8`).  fp8 is never a parity mode: the operands handed to
the kernel are exact e4m3 numbers, so the KERNEL is checked against float64 arithmetic on those same numbers (only fp32
accumulation and the output rounding may differ); the NETWORK's deviation from the fp32 network is measured and stated."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

E4M3 = torch.float8_e4m3fn


def _q(x):
    """fp32 -> e4m3 with saturation (what the kernels do), on the host."""
    return x.clamp(-448.0, 448.0).to(E4M3)


def test_gemm8_identity_catches_transposition():
    from deepcubea_amd import _lib
    _lib.require_gpu()
    k = n = 256
    w = ((torch.arange(n * k, dtype=torch.float32).view(n, k) % 13) - 6.0)  # small integers: exact in e4m3 and bf16
    w[:, 3] += 2.0
    x = torch.eye(k, dtype=torch.float32)
    one = torch.ones(n).cuda()
    y, y8 = _lib.gemm8(_q(x).cuda(), _q(w).cuda(), one, None, None, False, True, 1.0)
    assert torch.equal(y.float().cpu(), w.t().contiguous())
    assert torch.equal(y8.float().cpu(), w.t().contiguous())
    perm = torch.tensor([5, 0, 255, 17, 128, 64, 200], dtype=torch.long)
    y, _ = _lib.gemm8(_q(x[perm]).cuda().contiguous(), _q(w).cuda(), one, None, None, False, True, None)
    assert torch.equal(y.float().cpu(), w.t()[perm].contiguous())


@pytest.mark.parametrize("m,n,k", [(1, 4, 128), (300, 200, 256), (257, 1024, 1024), (1000, 1024, 5120), (513, 260, 384)])
def test_gemm8_layer_tail_against_float64(m, n, k):
    """relu((a.w^T) * scale + bias + skip): bf16 output within one unit in the last place of the float64 value (+ the
    instruction's accumulation bound); e4m3 output within one e4m3 step of the float64 value times out8_scale, saturating at 448; ragged m / n,
    K-tile counts 1, 2, 3, 8, 40; in-place residual form; no-bias / no-skip / no-ReLU forms."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    a = _q(torch.randn(m, k, generator=g) * 2.0)
    w = _q(torch.randn(n, k, generator=g) * 3.0)
    scale = (torch.rand(n, generator=g) + 0.5) * (0.05 / k ** 0.5)
    bias = torch.randn(n, generator=g)
    skip = torch.randn(m, n, generator=g).to(torch.bfloat16)
    dot = a.double() @ w.double().t()
    absdot = (a.double().abs() @ w.double().abs().t()) * scale.double()
    out8_scale = 37.0
    for use_b, use_s, relu in ((True, True, True), (True, False, True), (False, True, False), (False, False, False)):
        want = dot * scale.double() + (bias.double() if use_b else 0.0) + (skip.double() if use_s else 0.0)
        if relu:
            want = want.clamp_min(0.0)
        sk = skip.cuda().clone() if use_s else None
        y, y8 = _lib.gemm8(a.cuda(), w.cuda(), scale.cuda(), bias.cuda() if use_b else None, sk, relu, True, out8_scale,
                           out16=sk if use_s else None)
        # the e4m3 MFMA sums the 64 products of an instruction with LESS than fp32 precision (measured here: errors up to
        # 2^-17.5 of sum |a_i w_i|, against 2^-21 for the bf16 instruction) — immaterial next to e4m3's own 2^-4 steps, but it
        # is what bounds the distance to float64 when the terms cancel
        acc_tol = 2.0 ** -15 * absdot + 1e-30
        err = (y.double().cpu() - want).abs()
        tol = torch.maximum(want.abs(), torch.tensor(1e-3, dtype=torch.float64)) * 2.0 ** -7 * 1.01 + acc_tol
        assert bool((err <= tol).all()), ("bf16", use_b, use_s, relu, float((err / tol).max()))
        w8 = (want * out8_scale).clamp(-448.0, 448.0)
        err8 = (y8.double().cpu() - w8).abs()
        # one e4m3 step: 2^-3 of the value's binade (normal numbers), 2^-9 below 2^-6 (subnormals)
        step = torch.maximum(w8.abs(), torch.tensor(2.0 ** -6, dtype=torch.float64)) * 2.0 ** -3
        assert bool((err8 <= step * 1.01 + acc_tol * out8_scale).all()), ("e4m3", use_b, use_s, relu, float((err8 / step).max()))
        assert not bool(torch.isnan(y8.float()).any())
        if relu:
            assert float(y.float().min()) >= 0.0 and float(y8.float().min()) >= 0.0


def test_gemm8_repeatable_and_close_to_fp32_under_load():
    """Race screen of the ping-pong schedule at fp8: full-chip problems, K-tile counts 1..5, 8, 9 and 40, repeated launches —
    every launch bit-identical to the first, and the first within accumulation distance of an fp32 GEMM on the same
    (exact) operands: a fragment read that ever met a half-tile still in flight would show up as a wrong tile."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    g = torch.Generator().manual_seed(5)
    for m, n, k, reps in ((70000, 1024, 1024, 5), (70000, 1024, 1152, 3), (33000, 1024, 5120, 3), (66000, 768, 128, 2),
                          (66000, 512, 256, 2), (66000, 512, 384, 2), (66000, 512, 512, 2), (66000, 260, 640, 2)):
        a = _q(torch.randn(m, k, generator=g)).cuda()
        w = _q(torch.randn(n, k, generator=g)).cuda()
        scale = torch.full((n,), 1.0 / k ** 0.5).cuda()
        first, _ = _lib.gemm8(a, w, scale, None, None, False, True, None)
        ref = (a.float() @ w.float().t()) * scale
        err = (first.float() - ref).abs()
        tol = ref.abs() * 2.0 ** -7 + 4e-3  # bf16 rounding + the instruction's accumulation error (see the test above)
        assert bool((err <= tol).all()), (m, n, k, float((err / tol).max()))
        for _ in range(reps):
            again, _ = _lib.gemm8(a, w, scale, None, None, False, True, None)
            assert torch.equal(again, first), (m, n, k, int((again != first).sum()))
        del a, w, first, again, ref, err, tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_quant_e4m3_matches_the_saturating_host_conversion(dt):
    from deepcubea_amd import _lib
    _lib.require_gpu()
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(777, 1024, generator=g) * 40.0).to(dt)
    x[0, :8] = torch.tensor([0.0, -0.0, 448.0, -448.0, 1000.0, -1e9, 2.0 ** -9, 2.0 ** -10]).to(dt)
    got = _lib.quant_e4m3(x.cuda().contiguous(), 3.0).cpu()
    want = _q(x.float() * 3.0)
    assert torch.equal(got.view(torch.uint8), want.view(torch.uint8))


def test_layer1_kernel_e4m3_output_is_the_rounded_accumulator():
    """dca_l1_onehot_gemm with DCA_DT_E4M3: the same accumulator as its fp32 output, converted to e4m3 with saturation."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import l1_weight_tiles
    _lib.require_gpu()
    g = torch.Generator().manual_seed(11)
    n_pad, D, depth = 256, 54, 6
    w1 = (torch.randn(n_pad, D * depth, generator=g) * 30.0).to(torch.bfloat16).float()
    b1 = torch.randn(n_pad, generator=g) * 100.0
    tiles = l1_weight_tiles(w1, 1, _lib.l1_kpad(D, depth)).cuda()
    x = torch.randint(0, depth, (1500, D), dtype=torch.uint8, generator=g).cuda()
    y32 = _lib.l1_onehot_gemm(x, depth, tiles, 1, b1.cuda(), True, torch.float32)
    y8 = _lib.l1_onehot_gemm(x, depth, tiles, 1, b1.cuda(), True, _lib.E4M3)
    assert float(y32.max()) > 448.0  # the saturation branch is exercised
    assert torch.equal(y8.cpu().view(torch.uint8), _q(y32.cpu()).view(torch.uint8))


@pytest.mark.parametrize("D,depth,n_pad,m", [(54, 6, 256, 1500), (54, 6, 5120, 777), (16, 16, 128, 513), (25, 25, 256, 64),
                                             (49, 6, 128, 1000)])
def test_layer1_fp8_pipe_kernel_against_float64(D, depth, n_pad, m):
    """dca_l1_onehot_gemm8 (layer 1 on the f8f6f4 pipe): the one-hot row is exact in e4m3 and the weights ARE e4m3 numbers, so
    the accumulator must equal the float64 sum of the D selected weights up to fp32 accumulation — the output is that sum times
    scale[n] plus bias[n], ReLU, saturated, rounded to e4m3: compared byte for byte with the host's rounding of the float64
    value wherever that value is not within an fp32 rounding error of a tie between two e4m3 neighbours.  Rows are ragged
    (m not a multiple of 64 / 512), weights asymmetric in K and n (a transposed or permuted operand cannot pass), and the
    saturation branch is exercised."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import l1_weight_tiles8
    _lib.require_gpu()
    assert _lib.l1_supported8(D, depth) and _lib.l1_kpad8(D, depth) % 64 == 0
    g = torch.Generator().manual_seed(100 + D)
    K = D * depth
    w8 = (torch.randn(n_pad, K, generator=g) * 40.0 + torch.arange(K)[None, :] * 0.05 + torch.arange(n_pad)[:, None] * 0.01).to(E4M3)
    scale = (torch.rand(n_pad, generator=g) * 3.0 + 0.5).float()   # sums of D weights of ~40 each, times up to 3.5: both signs saturate
    bias = (torch.randn(n_pad, generator=g) * 30.0).float()
    tiles = l1_weight_tiles8(w8, _lib.l1_kpad8(D, depth)).cuda()
    x = torch.randint(0, depth, (m, D), dtype=torch.uint8, generator=g)
    y8 = _lib.l1_onehot_gemm8(x.cuda(), depth, tiles, scale.cuda(), bias.cuda(), True).cpu()
    idx = (torch.arange(D)[None, :] * depth + x.long())                        # [m, D] one-hot columns
    acc = w8.double().t()[idx].sum(dim=1)                                       # [m, n_pad] exact in float64
    v = torch.clamp(acc * scale.double()[None, :] + bias.double()[None, :], min=0.0)
    assert float(v.max()) > 448.0 and float((v > 0).float().mean()) > 0.2      # saturation and both ReLU branches are exercised
    want = _q(v.float())
    got, wnt = y8.view(torch.uint8), want.view(torch.uint8)
    same = got == wnt
    # Where the bytes differ the float64 value must sit on the midpoint between two neighbouring e4m3 numbers to within the
    # accumulator's error: the e4m3 MFMA sums the 64 products of an instruction with LESS than fp32 precision (up to
    # 2^-17.5 * sum|a_i w_i|, measured in round 3 — DESIGN §4.5), so acc is exact only to ~2^-16 * (sum of the |weights| picked),
    # times scale[n] in v.  |got - v| and |want - v| then differ by at most twice that.
    if not bool(same.all()):
        absum = w8.double().abs().t()[idx].sum(dim=1)
        eps = (absum * 2.0 ** -16 + acc.abs() * 1e-6) * scale.double()[None, :] + 1e-6
        vc = v.clamp(max=448.0)
        dg, dw = (y8.double() - vc).abs()[~same], (want.double() - vc).abs()[~same]
        assert bool(((dg - dw).abs() <= 2.0 * eps[~same]).all()), "a byte differs where the value is not at a rounding tie"
        assert int((~same).sum()) < 2e-2 * same.numel()
    # no ReLU: negative values saturate at -448
    y8n = _lib.l1_onehot_gemm8(x.cuda(), depth, tiles, scale.cuda(), bias.cuda(), False).cpu()
    vn = (acc * scale.double()[None, :] + bias.double()[None, :]).float()
    bad = y8n.view(torch.uint8) != _q(vn).view(torch.uint8)
    assert float(vn.min()) < -448.0 and int(bad.sum()) < 2e-2 * bad.numel()
    # repeated launches are bit-identical (the race screen: wave-private LDS slices, no workgroup barrier in the row loop)
    for _ in range(3):
        assert torch.equal(_lib.l1_onehot_gemm8(x.cuda(), depth, tiles, scale.cuda(), bias.cuda(), True).cpu().view(torch.uint8), got)


@torch.no_grad()
def test_fp8_layer1_on_the_fp8_pipe_costs_no_accuracy_that_matters():
    """Fp8Resnet with layer 1 on the f8f6f4 pipe (e4m3 weights; the default since round 6) against the round-5 arrangement
    (bf16 weights on the bf16 pipe, output rounded to e4m3): deviation of both from the fp32 parity-mode network on the same
    states, printed; the new arrangement may not be worse by more than a tenth of the format's own deviation."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import FastResnet, Fp8Resnet, ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    _lib.require_gpu()
    net = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(net, 2024)
    net.eval()
    x = torch.randint(0, 6, (6000, 54), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).cuda()
    y32 = FastResnet(net).cuda()(x)[:, 0]
    f_new, f_old = Fp8Resnet(net, l1="fp8").cuda(), Fp8Resnet(net, l1="bf16").cuda()
    assert f_new.l1_fp8 and not f_old.l1_fp8 and Fp8Resnet(net).l1_fp8
    scale = float(y32.abs().max())
    out = {}
    for name, f in (("l1 fp8 pipe", f_new), ("l1 bf16 pipe", f_old)):
        y = f(x)[:, 0]
        out[name] = (float((y - y32).abs().max()) / scale, float((y - y32).pow(2).mean().sqrt()) / scale)
    print("fp8 network vs fp32 network, deviation / max|h| (max, rms):", out)
    assert out["l1 fp8 pipe"][1] <= out["l1 bf16 pipe"][1] * 1.10 + 1e-3
    assert out["l1 fp8 pipe"][0] < 0.25


@torch.no_grad()
def test_fp8_network_deviation_from_the_fp32_network_is_stated():
    """Whole cube3 network at fp8 operand precision (layer 1 -> e4m3, nine dca_gemm8 layers, bf16 residual stream) against the
    fp32 parity-mode network on the same states.  NOT a parity mode (north star tolerance 1e-5 applies to fp32): the deviation is
    measured, printed and bounded loosely — e4m3 carries 3 mantissa bits (6 % per element before averaging over K)."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import FastResnet, Fp8Resnet, ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    _lib.require_gpu()
    net = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(net, 2024)
    net.eval()
    x = torch.randint(0, 6, (6000, 54), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).cuda()
    y32 = FastResnet(net).cuda()(x)[:, 0]
    y16 = FastResnet(net, torch.bfloat16).cuda()(x)[:, 0]
    f8 = Fp8Resnet(net, scaling="tensor").cuda()  # round 3's arrangement: one static scale per activation tensor
    y8 = f8(x)[:, 0]
    assert len(f8.act_scale) == 10 and all(s > 0 for s in f8.act_scale)
    y8b = f8(x[:100])[:, 0]  # calibrated once: later batches reuse the scales
    assert torch.equal(y8b, y8[:100])
    scale = float(y32.abs().max())
    dev8 = float((y8 - y32).abs().max()) / scale
    dev16 = float((y16 - y32).abs().max()) / scale
    rms8 = float((y8 - y32).pow(2).mean().sqrt()) / scale
    corr = float(torch.corrcoef(torch.stack([y8, y32]))[0, 1])
    print("network vs fp32 network, deviation / max|h|: fp8 max %.3e rms %.3e (bf16 max %.3e); correlation fp8~fp32 %.5f"
          % (dev8, rms8, dev16, corr))
    assert not bool(torch.isnan(y8).any())
    assert dev8 < 0.25 and rms8 < 0.05 and corr > 0.97  # measured: 0.103, 0.026, 0.987


# ---------------------------------------------------------------------------------------------------------------------
# block-scaled fp8 (dca_gemm8_mx / dca_l1_onehot_gemm_mx): one E8M0 scale per row and 64 elements, applied by the scaled MFMA
# ---------------------------------------------------------------------------------------------------------------------
def _deq(a8, sc):
    """e4m3 [m, k] with E8M0 block scales [m, k/64] -> float64 values."""
    m, k = a8.shape
    f = torch.pow(torch.tensor(2.0, dtype=torch.float64), sc.to(torch.float64) - 127.0)
    return a8.double() * f.repeat_interleave(64, dim=1)


@pytest.mark.parametrize("m,n,k", [(1, 64, 256), (300, 192, 256), (257, 1024, 1024), (1000, 1024, 5120), (513, 320, 768)])
def test_gemm8_mx_against_float64(m, n, k):
    """Operands = exact e4m3 numbers times exact powers of two: the kernel against float64 arithmetic on the same numbers.  The
    scales differ from row to row and between the two 64-deep halves of every K-tile (an op_sel / byte-order slip cannot pass);
    outputs: bf16 within one ulp (+ the instruction's accumulation bound), the e4m3 blocks within one e4m3 step of the
    float64 value divided by the block scale the kernel chose, which must be the smallest power of two that fits the block."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    g = torch.Generator().manual_seed(m * 11 + n * 5 + k)
    a = _q(torch.randn(m, k, generator=g) * 40.0)
    asc = torch.randint(119, 131, (m, k // 64), generator=g, dtype=torch.int64).to(torch.uint8)
    w = _q(torch.randn(n, k, generator=g) * 3.0)
    wsc = (torch.rand(n, generator=g) + 0.5) * (0.02 / k ** 0.5)
    bias = torch.randn(n, generator=g)
    skip = torch.randn(m, n, generator=g).to(torch.bfloat16)
    av = _deq(a, asc)
    dot = av @ w.double().t()
    absdot = (av.abs() @ w.double().abs().t()) * wsc.double()
    for use_b, use_s, relu in ((True, True, True), (False, False, False), (True, False, True)):
        want = dot * wsc.double() + (bias.double() if use_b else 0.0) + (skip.double() if use_s else 0.0)
        if relu:
            want = want.clamp_min(0.0)
        sk = skip.cuda().clone() if use_s else None
        y16, y8, ysc = _lib.gemm8_mx(a.cuda(), asc.cuda(), w.cuda(), wsc.cuda(), bias.cuda() if use_b else None, sk, relu, True, True,
                                     out16=sk if use_s else None)
        y = y16.double().cpu()
        ulp = torch.pow(torch.tensor(2.0, dtype=torch.float64), torch.floor(torch.log2(want.abs().clamp_min(1e-30))) - 7)
        # (the scaled instruction adds its 64 products after scaling them: with block scales 2^11 apart inside one sum its
        # accumulation error reaches 2^-16.5 of sum |a_i w_i| — measured; 2^-17.5 for the unscaled form above)
        tol = ulp * 1.01 + 2.0 ** -16 * absdot + 1e-30
        assert bool(((y - want).abs() <= tol).all()), (use_b, use_s, relu, float(((y - want).abs() / tol).max()))
        # block scales: smallest power of two with amax * 2^-e <= 448 (computed from the kernel's own fp32 values: allow the
        # neighbouring exponent where the block maximum sits within rounding of a power-of-two boundary)
        e = ysc.cpu().to(torch.float64) - 127.0
        am = want.abs().view(m, n // 64, 64).amax(dim=2)
        e_want = torch.ceil(torch.log2((am / 448.0).clamp_min(2.0 ** -126)))
        # (the kernel takes the maximum of ITS fp32 values: where a block's maximum sits next to a power-of-two boundary — or is
        # itself mostly accumulation noise — it may land one exponent off float64's; everywhere else it must agree)
        assert bool(((e - e_want).abs() <= 1).all()) and float((e == e_want).double().mean()) > 0.98
        deq = y8.double().cpu() * torch.pow(torch.tensor(2.0, dtype=torch.float64), e).repeat_interleave(64, dim=1)
        step = torch.pow(torch.tensor(2.0, dtype=torch.float64), e).repeat_interleave(64, dim=1) * 32.0  # e4m3 step at the top binade
        fine = torch.pow(torch.tensor(2.0, dtype=torch.float64), torch.floor(torch.log2(want.abs().clamp_min(1e-30))) - 3)
        assert bool(((deq - want).abs() <= torch.minimum(step, torch.maximum(fine, step / 2 ** 12)) * 0.51 + tol).all())


@torch.no_grad()
def test_l1_mx_output_is_the_fp32_layer_quantised_block_by_block():
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import l1_weight_tiles
    _lib.require_gpu()
    g = torch.Generator().manual_seed(5)
    D, depth, n_pad = 54, 6, 320
    w1 = (torch.randn(n_pad, D * depth, generator=g) * 4.0).to(torch.bfloat16).float()
    b1 = torch.randn(n_pad, generator=g)
    tiles = l1_weight_tiles(w1, 1, _lib.l1_kpad(D, depth)).cuda()
    x = torch.randint(0, depth, (1500, D), dtype=torch.uint8, generator=g).cuda()
    y32 = _lib.l1_onehot_gemm(x, depth, tiles, 1, b1.cuda(), True, torch.float32).double().cpu()
    y8, ysc = _lib.l1_onehot_gemm_mx(x, depth, tiles, b1.cuda(), True)
    e = ysc.cpu().to(torch.float64) - 127.0
    am = y32.abs().view(1500, n_pad // 64, 64).amax(dim=2)
    assert torch.equal(e, torch.ceil(torch.log2((am / 448.0).clamp_min(2.0 ** -126))).clamp(-126, 126))
    f = torch.pow(torch.tensor(2.0, dtype=torch.float64), e).repeat_interleave(64, dim=1)
    want8 = _q((y32 / f).float())  # exact power-of-two division, then the same e4m3 rounding
    assert torch.equal(y8.cpu().view(torch.uint8), want8.view(torch.uint8))


@torch.no_grad()
def test_fp8_block_scaled_network_deviation_is_the_formats_floor_at_any_depth():
    """VERDICT r03 item 5 asked for E8M0 block scales in place of one frozen per-tensor scale, expecting <= 3 % of max|h|.
    Built (dca_gemm8_mx) and measured: the deviation from the fp32 network does NOT come from the scaling — max 8.9-10.1 %,
    rms 2.7-3.3 %, the same as the per-tensor arrangement (10.2 % / 2.6 %) — it is e4m3's three mantissa bits.  The yardstick:
    the same network with activations and weights rounded to 3 mantissa bits and an UNBOUNDED exponent (ideal per-element
    scaling: nothing saturates, nothing underflows), evaluated in float32 on the host, deviates max 7.9 %, rms 2.3 %
    (weights alone 4.1 % / 1.3 %, activations alone 6.4 % / 1.9 %; synthetic weights, 3000 random states) — no scaling scheme
    can do better than that with e4m3 operands on both sides.  What block scaling buys is robustness: nothing is calibrated
    or frozen, so shallow states, deep states and random ones behave alike, nothing saturates, and a batch evaluated after a
    very different one gives the same bits as on its own.  Asserted: within 15 % / 4 % rms at every depth (2x the floor)."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import FastResnet, Fp8Resnet, ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    from oracle import c_oracle as co
    _lib.require_gpu()
    net = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(net, 2024)
    net.eval()
    f32, f8 = FastResnet(net).cuda(), Fp8Resnet(net, scaling="block").cuda()
    assert Fp8Resnet(net).scaling == "tensor"  # the default (faster, same accuracy): block scaling is `--nnet_dtype fp8mx`
    rng = np.random.default_rng(1)

    def scrambled(n, depth):
        s = np.tile(np.arange(54, dtype=np.uint8), (n, 1))
        for _ in range(depth):
            mv = rng.integers(0, 12, size=n)
            for a in range(12):
                idx = np.flatnonzero(mv == a)
                if idx.size:
                    s[idx] = co.next_state("cube3", s[idx], a)
        return torch.from_numpy(s // 9).cuda()

    shallow, deep = scrambled(3000, 2), scrambled(3000, 40)
    rand = torch.randint(0, 6, (3000, 54), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).cuda()
    out = {}
    for name, x in (("shallow (2 moves)", shallow), ("deep (40 moves)", deep), ("random stickers", rand)):
        y32, y8 = f32(x)[:, 0], f8(x)[:, 0]
        scale = float(y32.abs().max())
        out[name] = (float((y8 - y32).abs().max()) / scale, float((y8 - y32).pow(2).mean().sqrt()) / scale,
                     float(torch.corrcoef(torch.stack([y8, y32]))[0, 1]))
        assert not bool(torch.isnan(y8).any())
    print("fp8 (block-scaled) vs fp32 network — max / rms deviation over max|h|, correlation:", out)
    for name, (mx, rms, corr) in out.items():
        assert mx <= 0.15 and rms <= 0.04 and corr > 0.97, (name, mx, rms, corr)
    # order independence: the deep batch after the shallow one equals the deep batch on a fresh module
    assert torch.equal(f8(deep), Fp8Resnet(net, scaling="block").cuda()(deep))
    # the per-tensor arrangement next to it, calibrated on the shallow batch and evaluated on the deep one (what a search does)
    st = Fp8Resnet(net, scaling="tensor").cuda()
    st(shallow)
    yd32, yd = f32(deep)[:, 0], st(deep)[:, 0]
    print("per-tensor scales calibrated on shallow states, evaluated on deep ones: max deviation / max|h| = %.3f"
          % (float((yd - yd32).abs().max()) / float(yd32.abs().max())))
