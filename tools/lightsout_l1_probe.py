"""lightsout7 (49 lights, one-hot depth 6) has no one-hot MFMA layer-1 kernel in the fp32 / bf16 modes: whole-network forward with the
default arrangement (materialised one-hot rows + library GEMMs for layer 1) against `l1="embed"` (dca_l1_embed on the uint8 rows)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel
from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
model = ResnetModel(49, 6, 5000, 1000, 4, 1, True)
load_synthetic_weights(model, 2024)
model = model.eval()
M = 409600
x = torch.randint(0, 2, (M, 49), dtype=torch.uint8, device="cuda")
for dt in (torch.float32, torch.bfloat16):
    for l1 in ("auto", "embed"):
        f = FastResnet(model, dt, l1=l1).cuda()
        for _ in range(2):
            f(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            f(x)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        print("lightsout7", dt, "l1", l1, "uses_l1_kernel", f.uses_l1_kernel, "%.2f ms per %d rows" % (ms, M), flush=True)
