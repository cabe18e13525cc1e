"""GPU tests of the hand-written layer-1 MFMA kernel (csrc/dca_mlp.hip): relu(onehot(s) . W1^T + b1) straight from the
uint8 network-input rows, against a float64 evaluation of the same expression."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D,depth", [(54, 6), (16, 16)])
@pytest.mark.parametrize("planes,out_dtype,tol", [(3, torch.float32, 2e-6), (2, torch.float16, 2e-3), (1, torch.bfloat16, 1.5e-2),
                                                  (3, torch.bfloat16, 1e-2), (1, torch.float32, 1e-2)])
@pytest.mark.parametrize("m", [1, 257, 1500])
def test_l1_onehot_gemm_matches_float64(D, depth, planes, out_dtype, tol, m):
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import l1_weight_tiles
    _lib.require_gpu()
    assert _lib.l1_supported(D, depth) and not _lib.l1_supported(49, 49)
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
        scale = float(ref.abs().max())
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
