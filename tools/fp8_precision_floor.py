"""The precision floor of an fp8 (e4m3) evaluation of the cost-to-go network, on the host: activations and / or weights rounded to
3 mantissa bits with an UNBOUNDED exponent (ideal per-element scaling) against the fp32 network (synthetic weights).  What no
scaling scheme can beat; quoted in tests/test_gemm8_hip.py and DESIGN.md.   python tools/fp8_precision_floor.py"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcubea_amd.utils.pytorch_models import ResnetModel, fold_batchnorm
from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
torch.set_num_threads(8)
net = ResnetModel(54, 6, 5000, 1000, 4, 1, True); load_synthetic_weights(net, 2024); net.eval()
m = fold_batchnorm(net)
def q3(x):  # round to 3 explicit mantissa bits, unbounded exponent ("ideal scaling": no saturation, no underflow)
    mant, ex = torch.frexp(x)            # x = mant * 2^ex, mant in [0.5,1)
    return torch.ldexp(torch.round(mant * 16) / 16, ex)
def bf(x): return x.to(torch.bfloat16).float()
x = torch.randint(0, 6, (3000, 54), generator=torch.Generator().manual_seed(3))
oh = torch.nn.functional.one_hot(x, 6).float().view(-1, 324)
def fwd(qa, qw, stream):
    h = torch.relu(oh @ bf(m.fc1.weight).t() + m.fc1.bias)   # layer 1: bf16 weights, exact one-hot
    xx = torch.relu(qa(h) @ qw(m.fc2.weight).t() + m.fc2.bias)
    xx = stream(xx)
    for blk in m.blocks:
        la, lb = blk[0], blk[2]
        hh = torch.relu(qa(xx) @ qw(la.weight).t() + la.bias)
        xx = stream(torch.relu(qa(hh) @ qw(lb.weight).t() + lb.bias + xx))
    return (xx @ m.fc_out.weight.t() + m.fc_out.bias)[:, 0]
ident = lambda t: t
with torch.no_grad():
    y32 = fwd(ident, ident, ident)
    for name, qa, qw in (("acts+weights 3-bit mantissa (ideal per-element scaling)", q3, q3), ("acts only", q3, ident), ("weights only", ident, q3)):
        y = fwd(qa, qw, bf)
        s = float(y32.abs().max())
        print(name, "max %.4f rms %.4f corr %.5f" % (float((y-y32).abs().max())/s, float((y-y32).pow(2).mean().sqrt())/s, float(torch.corrcoef(torch.stack([y,y32]))[0,1])))
