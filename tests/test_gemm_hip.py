"""GPU tests of the hand-written fp32-accurate dense-layer kernel (csrc/dca_gemm.hip, `dca_f16x3_gemm`): three f16 MFMA
products per K-step over fp16 operand planes, layer tail (scale, bias, residual add, ReLU, split of the result into the
next layer's planes) in the epilogue — against float64 evaluations of the same expression."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[3, 2], ids=["ping_pong_256", "two_stage_256"])
def variant(request):
    from deepcubea_amd import _lib
    _lib.f16x3_gemm_variant(request.param)
    yield request.param
    _lib.f16x3_gemm_variant(3)


def _split_w(w):
    from deepcubea_amd.utils.pytorch_models import _pow2_scale, _split_f16
    sc = _pow2_scale(w)
    wh, wl = _split_f16(w, sc)
    return wh.contiguous(), wl.contiguous(), (1.0 / sc).contiguous()


def test_identity_activations_with_asymmetric_weights_catch_transposition(variant):
    """A = I (exactly representable), asymmetric W: the output must be W^T tile for tile — a swapped row/column map of the
    MFMA accumulator or of the LDS images cannot pass."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    k = n = 256
    w = (torch.arange(n * k, dtype=torch.float32).view(n, k) % 251) - 125.0   # integers: exact in fp16
    w[:, 3] += 7.0
    wh, wl = w.to(torch.float16).cuda(), torch.zeros(n, k, dtype=torch.float16).cuda()
    x = torch.eye(k, dtype=torch.float32).cuda()
    planes = _lib.split_planes(x)
    assert torch.equal(planes[0].float(), x) and not planes[1].any()
    _, y = _lib.f16x3_gemm(planes, wh, wl, None, 1.0, None, None, False, False, True)
    assert torch.equal(y.cpu(), w.t().contiguous())
    # ragged m: a row subset in a different order
    idx = torch.tensor([5, 0, 200, 131, 77, 255, 128])
    _, y2 = _lib.f16x3_gemm(_lib.split_planes(x[idx.cuda()].contiguous()), wh, wl, None, 1.0, None, None, False, False, True)
    assert torch.equal(y2.cpu(), w.t()[idx])


@pytest.mark.parametrize("m", [1, 130, 1000])
@pytest.mark.parametrize("n,k", [(64, 64), (192, 128), (1024, 1024), (1024, 5120)])
def test_f16x3_gemm_is_fp32_accurate(m, n, k, variant):
    from deepcubea_amd import _lib
    g = torch.Generator().manual_seed(1000 * m + n + k)
    x = torch.randn(m, k, generator=g) * 3.0
    x[:, ::7] *= 40.0                       # a spread of magnitudes inside a row
    w = torch.randn(n, k, generator=g) / np.sqrt(k)
    w[5] *= 300.0                           # units of very different scale: per-row power-of-two scaling
    w[7] *= 1e-3
    b = torch.randn(n, generator=g)
    skip = torch.randn(m, n, generator=g)
    wh, wl, inv = _split_w(w)
    planes = _lib.split_planes(x.cuda())
    ref = x.double() @ w.double().t()
    mag = (x.double().abs() @ w.double().abs().t())   # sum |x||w| per output: the natural error scale
    for bias, sk, relu in ((None, None, False), (b, skip, True), (b, None, True)):
        pl, y = _lib.f16x3_gemm(planes, wh.cuda(), wl.cuda(), inv.cuda(), 1.0, None if bias is None else bias.cuda(),
                                None if sk is None else sk.cuda(), relu, True, True)
        want = ref + (bias.double() if bias is not None else 0) + (sk.double() if sk is not None else 0)
        if relu:
            want = torch.relu(want)
        err = (y.double().cpu() - want).abs()
        # fp32-GEMM class: a few 2^-22 of sum|x||w| (measured ~2e-7), plus one fp32 rounding of the result
        assert float((err / (mag * 2.0 ** -19 + want.abs() * 2.0 ** -22 + 1e-30)).max()) <= 1.0, float(err.max())
        # the planes are exactly the split of the fp32 result
        hi = y.to(torch.float16)
        lo = (y - hi.float()).to(torch.float16)
        assert torch.equal(pl[0], hi) and torch.equal(pl[1], lo)
    # planes-only and fp32-only calls agree with the combined one
    pl2, none = _lib.f16x3_gemm(planes, wh.cuda(), wl.cuda(), inv.cuda(), 1.0, b.cuda(), None, True, True, False)
    assert none is None and torch.equal(pl2, pl)


def test_overflow_flag_and_split_planes():
    from deepcubea_amd import _lib
    x = torch.randn(300, 128) * 1000.0
    pl = _lib.split_planes(x.cuda())
    hi = x.to(torch.float16)
    assert torch.equal(pl[0].cpu(), hi) and torch.equal(pl[1].cpu(), (x - hi.float()).to(torch.float16))
    assert float((pl[0].float() + pl[1].float() - x.cuda()).abs().max()) <= float(x.abs().max()) * 2.0 ** -21
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    w = torch.ones(64, 128)
    wh, wl, inv = _split_w(w)
    _lib.f16x3_gemm(pl, wh.cuda(), wl.cuda(), inv.cuda(), 1.0, None, None, False, True, True, flag)
    assert int(flag.item()) == 0
    big = torch.full((4, 128), 600.0).cuda()   # 128 * 600 = 76800 > fp16 range
    _lib.f16x3_gemm(_lib.split_planes(big), wh.cuda(), wl.cuda(), inv.cuda(), 1.0, None, None, False, True, True, flag)
    assert int(flag.item()) == 1
    flag.zero_()
    _lib.split_planes(torch.full((4, 8), 7e4).cuda(), flag)
    assert int(flag.item()) == 1


@torch.no_grad()
def test_hand_written_layers_equal_the_library_arrangement(golden):
    """FastResnet(gemm="hip") (one dca_f16x3_gemm launch per layer) vs round 1's library f16 GEMM + glue kernel on the same
    split weights: both fp32-accurate, so they agree to a few fp32 ulps of the activations; both within 1e-5 of the reference."""
    from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    full = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(full, 2024)
    x = torch.tensor(golden["cube3_resnet_seed2024_x"]).cuda()
    ref = golden["cube3_resnet_seed2024_y"]
    hip, lib_ = FastResnet(full, gemm="hip").cuda(), FastResnet(full, gemm="library").cuda()
    yh, yl = hip(x)[:, 0].cpu().numpy(), lib_(x)[:, 0].cpu().numpy()
    assert np.max(np.abs(yh - ref)) < 1e-5 and np.max(np.abs(yl - ref)) < 1e-5
    xb = torch.randint(0, 6, (20000, 54), dtype=torch.uint8, device="cuda")
    assert float((hip(xb) - lib_(xb)).abs().max()) < 1e-5
    # the one-hot entry (geometries fed with one-hot rows) takes the same layers
    assert float((hip.forward_onehot(hip.encode(xb[:700])) - hip(xb[:700])).abs().max()) < 1e-5
    assert hip.split_fallbacks == 0 and lib_.split_fallbacks == 0


def test_f16x3_schedules_agree_bit_for_bit_under_load():
    """Race screen for the ping-pong schedule (variant 3): it adds the same products in the same order as the two-stage
    kernel (variant 2), so outputs and result planes are bit-identical — unless a fragment read ever meets a half-tile that has
    not landed (or has been restaged).  Full-chip problems, K-step counts 1..5, 32, 34 and 160, repeated launches."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    g = torch.Generator().manual_seed(77)
    for m, n, k, reps in ((70000, 1024, 1024, 5), (70000, 1024, 1088, 3), (33000, 1024, 5120, 3), (66000, 768, 64, 2),
                          (66000, 512, 128, 2), (66000, 512, 192, 2), (66000, 260, 320, 2)):
        x = (torch.randn(m, k, generator=g) * 2.0).cuda()
        w = torch.randn(n, k, generator=g) / np.sqrt(k)
        wh, wl, inv = _split_w(w)
        wh, wl, inv = wh.cuda(), wl.cuda(), inv.cuda()
        b = torch.randn(n, generator=g).cuda()
        planes = _lib.split_planes(x)
        _lib.f16x3_gemm_variant(2)
        (ph, pl), y = _lib.f16x3_gemm(planes, wh, wl, inv, 1.0, b, None, True, True, True)
        try:
            for v in (3,):
                _lib.f16x3_gemm_variant(v)
                for _ in range(reps):
                    (qh, ql), z = _lib.f16x3_gemm(planes, wh, wl, inv, 1.0, b, None, True, True, True)
                    assert torch.equal(z, y), (v, m, n, k, int((z != y).sum()))
                    assert torch.equal(qh, ph) and torch.equal(ql, pl)
            # the residual form (fp32 skip rows through the tail), ragged m inside the last tile
            skip = torch.randn(m, n, generator=g).cuda()
            _lib.f16x3_gemm_variant(2)
            (ph, pl), y = _lib.f16x3_gemm(planes, wh, wl, inv, 1.0, b, skip, True, True, True)
            _lib.f16x3_gemm_variant(3)
            (qh, ql), z = _lib.f16x3_gemm(planes, wh, wl, inv, 1.0, b, skip, True, True, True)
            assert torch.equal(z, y) and torch.equal(qh, ph) and torch.equal(ql, pl)
            del skip
        finally:
            _lib.f16x3_gemm_variant(3)
        del x, planes, y, z, ph, pl, qh, ql


# ---------------------------------------------------------------------------------------------------------------------
# dca_gemm16 (csrc/dca_gemm16.hip): the same layer in the non-parity 16-bit modes, tail in the epilogue
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(params=[1, 2, 3], ids=["two_stage", "eight_phase", "eight_phase_lean_tail"])
def variant16(request):
    from deepcubea_amd import _lib
    _lib.gemm16_variant(request.param)
    yield request.param
    _lib.gemm16_variant(3)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_gemm16_identity_catches_transposition(dt, variant16):
    from deepcubea_amd import _lib
    _lib.require_gpu()
    k = n = 256
    w = ((torch.arange(n * k, dtype=torch.float32).view(n, k) % 251) - 125.0)  # |v| <= 125: exact in bf16 and fp16
    w[:, 3] += 2.0
    x = torch.eye(k, dtype=torch.float32)
    y = _lib.gemm16(x.to(dt).cuda(), w.to(dt).cuda(), None, None, False)
    assert torch.equal(y.float().cpu(), w.t().contiguous())
    perm = torch.tensor([5, 0, 255, 17, 128, 64, 200], dtype=torch.long)
    y = _lib.gemm16(x[perm].to(dt).cuda().contiguous(), w.to(dt).cuda(), None, None, False)
    assert torch.equal(y.float().cpu(), w.t()[perm].contiguous())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("m,n,k", [(1, 4, 64), (300, 200, 128), (257, 1024, 1024), (1000, 1024, 5120), (513, 260, 192)])
def test_gemm16_layer_tail_against_float64(dt, m, n, k, variant16):
    """relu(a.w^T + bias + skip) rounded to the 16-bit type: the kernel accumulates in fp32, the yardstick in float64 — they
    may land on opposite sides of a rounding boundary, so one unit in the last place of the output type is allowed (and
    nothing more); ragged m / n, the in-place residual form (out == skip), no-bias / no-skip / no-ReLU forms."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    a = (torch.randn(m, k, generator=g) * 0.5).to(dt)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dt)
    bias = torch.randn(n, generator=g)
    skip = torch.randn(m, n, generator=g).to(dt)
    ref64 = a.double() @ w.double().t()
    absdot = a.double().abs() @ w.double().abs().t()
    # one unit in the last place is at most |v| * 2^-(p-1) for a p-bit significand (8 bits bf16, 11 bits fp16)
    ulp = lambda v: torch.maximum(v.abs(), torch.tensor(1e-3, dtype=torch.float64)) * (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10)
    for use_b, use_s, relu in ((True, True, True), (True, False, True), (False, True, False), (False, False, False)):
        want = ref64 + (bias.double() if use_b else 0.0) + (skip.double() if use_s else 0.0)
        if relu:
            want = want.clamp_min(0.0)
        sk = skip.cuda().clone() if use_s else None
        y = _lib.gemm16(a.cuda(), w.cuda(), bias.cuda() if use_b else None, sk, relu, out=sk if use_s else None)
        err = (y.double().cpu() - want).abs()
        # ... plus the fp32 accumulation error of the K-long dot product itself (MFMA partial sums, order unspecified):
        # 16 fp32 epsilons of sum |a_i w_i| — loose against the sqrt(k)-like growth seen in practice, tight against k eps
        tol = ulp(want) * 1.01 + 16.0 * 2.0 ** -24 * absdot
        assert bool((err <= tol).all()), (use_b, use_s, relu, float((err / tol).max()))
        if relu:
            assert float(y.float().min()) >= 0.0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_gemm16_schedules_agree_bit_for_bit_under_load(dt):
    """Race screen for the 8-phase schedule: both schedules add the same products in the same order, so their outputs are
    bit-identical — unless a fragment read ever meets a half-tile that has not landed (or has been restaged).  Full-chip
    problems (every CU busy, DMA latency at its worst), K-tile counts 1..5, 16, 17 and 80, repeated launches."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    g = torch.Generator().manual_seed(99)
    for m, n, k, reps in ((70000, 1024, 1024, 6), (70000, 1024, 1088, 3), (33000, 1024, 5120, 3), (66000, 768, 64, 2),
                          (66000, 512, 128, 2), (66000, 512, 192, 2), (66000, 512, 256, 2), (66000, 260, 320, 2)):
        a = (torch.randn(m, k, generator=g) * 0.5).to(dt).cuda()
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dt).cuda()
        bias = torch.randn(n, generator=g).cuda()
        _lib.gemm16_variant(1)
        want = _lib.gemm16(a, w, bias, None, True)
        try:
            for v in (2, 3):  # 3: swapped operand roles + lean tail per layer form (the default)
                _lib.gemm16_variant(v)
                for _ in range(reps):
                    got = _lib.gemm16(a, w, bias, None, True)
                    assert torch.equal(got, want), (v, m, n, k, int((got != want).sum()))
            skip = torch.randn(m, n, generator=g).to(dt).cuda()
            _lib.gemm16_variant(1)
            want = _lib.gemm16(a, w, None, skip, True)
            _lib.gemm16_variant(1)
            want_b = _lib.gemm16(a, w, bias, skip, True)
            for v in (2, 3):
                _lib.gemm16_variant(v)
                got = _lib.gemm16(a, w, None, skip, True)
                assert torch.equal(got, want), v
                got = _lib.gemm16(a, w, bias, skip, True)  # (the lean tails serve the forms WITH bias)
                assert torch.equal(got, want_b), v
            sk2 = skip.clone()
            got = _lib.gemm16(a, w, bias, sk2, True, out=sk2)  # in place: the residual stream
            assert torch.equal(got, want_b)
            del sk2, want_b
            del skip
        finally:
            _lib.gemm16_variant(3)
        del a, w, want, got


@torch.no_grad()
def test_fastresnet_bf16_on_the_hand_written_kernels_matches_the_library_path():
    """Whole network, bf16 mode: every dense layer on dca_gemm16 (layer 1 on dca_l1_onehot_gemm) vs the same FastResnet on
    the library GEMMs — both are bf16 evaluations of the same weights; they differ by bf16 rounding of intermediate sums
    only.  The deviation from the fp32 network is stated (not a parity mode: north star tolerance 1e-5 applies to fp32)."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    _lib.require_gpu()
    net = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(net, 2024)
    net.eval()
    x = torch.randint(0, 6, (3000, 54), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).cuda()
    y32 = FastResnet(net).cuda()(x)[:, 0]
    hip = FastResnet(net, torch.bfloat16, gemm16="hip").cuda()
    libm = FastResnet(net, torch.bfloat16).cuda()
    yh, yl = hip(x)[:, 0], libm(x)[:, 0]
    scale = float(y32.abs().max())
    dev_h, dev_l = float((yh - y32).abs().max()) / scale, float((yl - y32).abs().max()) / scale
    print("bf16 network vs fp32 network, max deviation / max|h|: hand-written %.3e, library %.3e" % (dev_h, dev_l))
    assert dev_h < 5e-2 and dev_l < 5e-2          # bf16: 8 mantissa bits through 10 layers
    assert dev_h < 2.0 * dev_l + 1e-3             # no worse than the library's bf16 evaluation
