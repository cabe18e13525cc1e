"""FastResnet (padded / epilogue-fused re-layout) vs the folded ResnetModel on the cube3 network, synthetic weights."""
import sys, torch
sys.path.insert(0, ".")
from deepcubea_amd.utils import env_utils
from deepcubea_amd.utils.pytorch_models import FastResnet, fold_batchnorm
from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
M = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
env = env_utils.get_environment("cube3")
model = env.get_nnet_model()
load_synthetic_weights(model, 2024)
model = model.cuda().eval()
folded = fold_batchnorm(model).cuda().eval()
x = torch.randint(0, 6, (M, 54), dtype=torch.uint8, device="cuda")
oh = torch.nn.functional.one_hot(x.long(), 6).float().view(M, 324)
def t(fn, n=5):
    for _ in range(2): y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y
FL = 2 * 14621000 * M
with torch.no_grad():
    ms, y32 = t(lambda: folded.forward_onehot(oh)[:, 0])
    print("folded fp32        %.3f ms %.1f TF" % (ms, FL / ms / 1e9))
    f32n = FastResnet(model, torch.float32, split=False).cuda()
    ohp = torch.nn.functional.pad(oh, (0, f32n.in_pad - 324)).contiguous()
    ms, yf = t(lambda: f32n.forward_onehot(ohp)[:, 0])
    print("fast   fp32 native %.3f ms %.1f TF  maxdiff %.3g (|y|max %.3g)" % (ms, FL / ms / 1e9, (yf - y32).abs().max().item(), y32.abs().max().item()))
    f32 = FastResnet(model, torch.float32).cuda()
    ms, yf = t(lambda: f32(x)[:, 0])
    print("fast   fp32 f16x3 + layer-1 kernel %.3f ms %.1f TF  maxdiff %.3g" % (ms, FL / ms / 1e9, (yf - y32).abs().max().item()))
    y64 = fold_batchnorm(model).double().cuda().forward_onehot(oh[:4096].double())[:, 0]
    print("   vs float64: folded fp32 %.3g | native %.3g | f16x3 %.3g" % ((y32[:4096].double() - y64).abs().max().item(),
          (f32n.forward_onehot(ohp[:4096])[:, 0].double() - y64).abs().max().item(), (f32(x[:4096])[:, 0].double() - y64).abs().max().item()))
    def ac():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return folded.forward_onehot(oh)[:, 0].float()
    ms, yb = t(ac)
    print("folded autocast bf16 %.3f ms %.1f TF  maxdiff_vs_fp32 %.3g" % (ms, FL / ms / 1e9, (yb - y32).abs().max().item()))
    for dt in (torch.bfloat16, torch.float16):
        fb = FastResnet(model, dt).cuda()
        ohb = ohp.to(dt)
        ms, yfb = t(lambda: fb.forward_onehot(ohb)[:, 0])
        print("fast   %s      %.3f ms %.1f TF  maxdiff_vs_fp32 %.3g" % (str(dt)[6:], ms, FL / ms / 1e9, (yfb - y32).abs().max().item()))

# layer 1 alone: hand-written one-hot MFMA kernel (csrc/dca_mlp.hip) vs the library GEMM on materialised one-hot rows
from deepcubea_amd import _lib
with torch.no_grad():
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        f = FastResnet(model, dt).cuda()
        ohd = torch.nn.functional.pad(oh, (0, f.in_pad - 324)).to(dt).contiguous()
        ms_lib, _ = t(lambda: torch._addmm_activation(f.biases[0], ohd, f.weights[0].t()))
        if f.l1_tiles is None:
            continue
        ms_k, _ = t(lambda: _lib.l1_onehot_gemm(x, 6, f.l1_tiles, f.l1_planes, f.l1_bias, True, dt))
        ms_all, yk = t(lambda: f(x)[:, 0])
        fl = 2 * 324 * 5000 * M
        print("layer1 %s: library %.3f ms (%.0f TF)  kernel(P=%d) %.3f ms (%.0f TF useful, %.0f TF on the MFMA pipe) | whole net via kernel %.3f ms"
              % (str(dt)[6:], ms_lib, fl / ms_lib / 1e9, f.l1_planes, ms_k, fl / ms_k / 1e9,
                 2 * 336 * 5120 * M * f.l1_planes / ms_k / 1e9, ms_all))
