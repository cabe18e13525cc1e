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
    f32 = FastResnet(model, torch.float32).cuda()
    ohp = torch.nn.functional.pad(oh, (0, f32.in_pad - 324)).contiguous()
    ms, yf = t(lambda: f32.forward_onehot(ohp)[:, 0])
    print("fast   fp32        %.3f ms %.1f TF  maxdiff %.3g (|y|max %.3g)" % (ms, FL / ms / 1e9, (yf - y32).abs().max().item(), y32.abs().max().item()))
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
