"""GPU tests of the hand-written layer-1 MFMA kernel (csrc/dca_mlp.hip): relu(onehot(s) . W1^T + b1) straight from the
uint8 network-input rows, against a float64 evaluation of the same expression."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D,depth", [(54, 6), (16, 16), (25, 25), (36, 36), (49, 49)])
@pytest.mark.parametrize("planes,out_dtype,tol", [(3, torch.float32, 2e-6), (2, torch.float16, 2e-3), (1, torch.bfloat16, 1.5e-2),
                                                  (3, torch.bfloat16, 1e-2), (1, torch.float32, 1e-2)])
@pytest.mark.parametrize("m", [1, 257, 1500])
def test_l1_onehot_gemm_matches_float64(D, depth, planes, out_dtype, tol, m):
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import l1_weight_tiles
    _lib.require_gpu()
    assert _lib.l1_supported(D, depth) and not _lib.l1_supported(64, 64)
    g = torch.Generator().manual_seed(D * 1000 + planes * 10 + m)
    K, n_pad = D * depth, 192
    w = torch.randn(n_pad, K, generator=g) * 0.2      # asymmetric, every column different
    w[5] *= 37.0                                     # a few large-magnitude rows: exercise the plane split
    b = torch.randn(n_pad, generator=g)
    x = torch.randint(0, depth, (m, D), generator=g, dtype=torch.uint8)
    kpad = _lib.l1_kpad(D, depth)
    assert kpad % 16 == 0 and kpad >= K
    tiles = l1_weight_tiles(w, planes, kpad).cuda()
    for relu in (True, False):
        y = _lib.l1_onehot_gemm(x.cuda(), depth, tiles, planes, b.cuda(), relu, out_dtype).float().cpu().double()
        oh = torch.nn.functional.one_hot(x.long(), depth).double().view(m, K)
        ref = oh @ w.double().t() + b.double()
        if relu:
            ref = torch.relu(ref)
        # error scale: the largest output, or — reduced-precision weight planes — the largest sum of |weights| a row picks
        scale = max(float(ref.abs().max()), float((oh @ w.double().abs().t()).max()) if planes < 3 else 0.0)
        assert float((y - ref).abs().max()) <= tol * max(1.0, scale), (float((y - ref).abs().max()), scale)


def test_fastresnet_uint8_path_uses_the_kernel_and_matches(golden, tiny_resnet):
    from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    full = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(full, 2024)
    x = torch.tensor(golden["cube3_resnet_seed2024_x"]).cuda()
    ref = golden["cube3_resnet_seed2024_y"]
    tol = 1e-5 * max(1.0, float(np.abs(ref).max()))
    fast = FastResnet(full).cuda()
    assert fast.l1_tiles is not None and fast.l1_planes == 3
    y_kernel = fast(x)[:, 0].cpu().numpy()                       # uint8 rows -> dca_l1_onehot_gemm -> library GEMMs
    y_onehot = fast.forward_onehot(fast.encode(x))[:, 0].cpu().numpy()  # all-library path
    assert np.max(np.abs(y_kernel - ref)) < tol and np.max(np.abs(y_kernel - y_onehot)) < tol
    # reduced-precision modes stay close to fp32 (not parity modes)
    for dt, lim in ((torch.bfloat16, 5e-2), (torch.float16, 1e-2)):
        yl = FastResnet(full, dt).cuda()(x)[:, 0].float().cpu().numpy()
        assert np.max(np.abs(yl - ref)) < lim


def test_act_split_kernel_matches_torch():
    from deepcubea_amd import _lib
    torch.manual_seed(3)
    m, n = 777, 1024
    y = torch.randn(m, n, device="cuda") * 3000.0
    b = torch.randn(n, device="cuda")
    sk = torch.randn(m, n, device="cuda")
    cs = torch.exp2(-torch.randint(6, 14, (n,), device="cuda").float())  # per-column powers of two
    for bias, skip, relu, alpha in ((b, sk, True, 2.0 ** -10), (None, None, False, 2.0 ** -10), (b, None, True, cs)):
        a3, x = _lib.act_split(y, bias, skip, alpha, relu, True)
        v = y * alpha + (bias if bias is not None else 0) + (skip if skip is not None else 0)
        if relu:
            v = torch.relu(v)
        assert torch.equal(x, v)  # same fp32 operations in the same order
        hi = v.to(torch.float16)
        lo = (v - hi.float()).to(torch.float16)
        a3v = a3.view(m, n, 3)  # a3[3k..3k+2] = (vh, vl, vh)
        assert torch.equal(a3v[:, :, 0], hi) and torch.equal(a3v[:, :, 1], lo) and torch.equal(a3v[:, :, 2], hi)
        # the split reproduces v to 2^-21
        assert float((hi.float() + lo.float() - v).abs().max()) <= float(v.abs().max()) * 2.0 ** -21
    a3, x = _lib.act_split(y, b, None, 1.0, True, True, want_a3=False)
    assert a3 is None and torch.equal(x, torch.relu(y + b))


def test_f16x3_split_network_is_fp32_accurate(golden, tiny_resnet):
    """fp32 parity mode = every dense layer after the first as one f16 GEMM over split operands (FastResnet.split): must stay
    within the north star's 1e-5 of the reference's own fp32 forward, like the native fp32 GEMM path."""
    from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    full = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    load_synthetic_weights(full, 2024)
    x = torch.tensor(golden["cube3_resnet_seed2024_x"]).cuda()
    ref = golden["cube3_resnet_seed2024_y"]
    tol = 1e-5 * max(1.0, float(np.abs(ref).max()))
    split, native = FastResnet(full).cuda(), FastResnet(full, split=False).cuda()
    assert split.split and not native.split
    ys, yn = split(x)[:, 0].cpu().numpy(), native(x)[:, 0].cpu().numpy()
    assert np.max(np.abs(ys - ref)) < tol and np.max(np.abs(yn - ref)) < tol
    # a larger batch of random states: split vs native fp32 GEMMs
    xb = torch.randint(0, 6, (20000, 54), dtype=torch.uint8, device="cuda")
    d = (split(xb) - native(xb)).abs().max().item()
    assert d < tol, d
    # the one-hot input path (geometries without the layer-1 kernel) takes the same split layers
    assert split.onehot_dtype == torch.float16 and native.onehot_dtype == torch.float32
    d = (split.forward_onehot(split.encode(xb[:512])) - native.forward_onehot(native.encode(xb[:512]))).abs().max()
    assert float(d) < tol
    m = ResnetModel(54, 6, 64, 32, 2, 1, True)
    m.load_state_dict({k[2:]: torch.tensor(tiny_resnet[k]) for k in tiny_resnet.files if k.startswith("w:")})
    yt = FastResnet(m).cuda()(torch.tensor(tiny_resnet["x"]).cuda())[:, 0].cpu().numpy()
    assert np.max(np.abs(yt - tiny_resnet["y"])) < 1e-5


def test_l1_kernel_split_epilogue_equals_act_split_of_its_fp32_output():
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import l1_weight_tiles
    torch.manual_seed(9)
    m, n_pad = 1111, 128
    w = torch.randn(n_pad, 324) * 0.3
    b = torch.randn(n_pad).cuda()
    x = torch.randint(0, 6, (m, 54), dtype=torch.uint8).cuda()
    tiles = l1_weight_tiles(w, 3, _lib.l1_kpad(54, 6)).cuda()
    y = _lib.l1_onehot_gemm(x, 6, tiles, 3, b, True, torch.float32)
    a3 = _lib.l1_onehot_gemm(x, 6, tiles, 3, b, True, torch.float32, split=True)
    want, _ = _lib.act_split(y, None, None, 1.0, False, False)
    assert a3.shape == (m, 3 * n_pad) and torch.equal(a3, want)


def test_f16x3_survives_a_wide_spread_of_unit_scales():
    """BatchNorm folding can leave output units with very different weight magnitudes; the per-unit power-of-two scales keep
    the f16x3 layers as accurate as the fp32 GEMMs (both compared with a float64 evaluation)."""
    from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel, fold_batchnorm
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    net = ResnetModel(54, 6, 600, 200, 2, 1, True)
    load_synthetic_weights(net, 5)
    with torch.no_grad():
        for bn in [net.bn1, net.bn2] + [b for blk in net.blocks for b in (blk[1], blk[3])]:
            c = bn.weight.numel()
            bn.weight[: c // 4] *= 8.0        # units with large folded weights ...
            bn.weight[c // 4: c // 2] *= 1.0e-3  # ... next to tiny ones (spread 2e4 across units, activations up to 6e3)
    x = torch.randint(0, 6, (4096, 54), dtype=torch.uint8, device="cuda")
    oh = torch.nn.functional.one_hot(x.long(), 6).double().view(-1, 324)
    with torch.no_grad():
        y64 = fold_batchnorm(net).double().cuda().forward_onehot(oh)[:, 0]
    scale = float(y64.abs().max())
    fs = FastResnet(net).cuda()
    es = float((fs(x)[:, 0].double() - y64).abs().max()) / scale
    en = float((FastResnet(net, split=False).cuda()(x)[:, 0].double() - y64).abs().max()) / scale
    assert fs.split_fallbacks == 0 and es < 1e-5 and en < 1e-5 and es < 4 * en + 1e-6, (es, en)
    # activations beyond the fp16 range: the kernels raise the overflow flag and the batch is redone with fp32 GEMMs
    with torch.no_grad():
        net.bn1.weight *= 1.0e4
    fo, fnat = FastResnet(net).cuda(), FastResnet(net, split=False).cuda()
    assert torch.equal(fo(x), fnat(x)) and fo.split_fallbacks == 1
    assert torch.equal(fo.forward_onehot(fo.encode(x)), fnat.forward_onehot(fnat.encode(x))) and fo.split_fallbacks == 2


@pytest.mark.parametrize("D,depth", [(54, 6), (49, 49)])
def test_l1_kernel_planes_epilogue_is_the_split_of_its_fp32_output(D, depth):
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import l1_weight_tiles
    torch.manual_seed(11)
    m, n_pad = 777, 128
    w = torch.randn(n_pad, D * depth) * 0.3
    b = torch.randn(n_pad).cuda()
    x = torch.stack([torch.randperm(D) for _ in range(m)]).to(torch.uint8).cuda() if depth == D else \
        torch.randint(0, depth, (m, D), dtype=torch.uint8).cuda()
    tiles = l1_weight_tiles(w, 3, _lib.l1_kpad(D, depth)).cuda()
    y = _lib.l1_onehot_gemm(x, depth, tiles, 3, b, True, torch.float32)
    pl = _lib.l1_onehot_gemm(x, depth, tiles, 3, b, True, torch.float32, split="planes")
    hi = y.to(torch.float16)
    assert pl.shape == (2, m, n_pad) and torch.equal(pl[0], hi) and torch.equal(pl[1], (y - hi.float()).to(torch.float16))
