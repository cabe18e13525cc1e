"""Workload for rocprofv3: the cost-to-go network (FastResnet layout) of cube3 (default) or a sliding puzzle on one dedup-first
batch of 204 800 rows, fp32 (parity mode), bf16 or fp8 (Fp8Resnet).  `python tools/profile_nnet.py fp32|bf16|fp8 [reps] [env]`"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcubea_amd.utils import env_utils
from deepcubea_amd.utils.pytorch_models import FastResnet, Fp8Resnet
from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
name = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp8": None}[name]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
M = 204800
envn = sys.argv[3] if len(sys.argv) > 3 else "cube3"
model = env_utils.get_environment(envn).get_nnet_model()
load_synthetic_weights(model, 2024)
fast = (Fp8Resnet(model) if name == "fp8" else FastResnet(model, dt)).cuda()
if envn == "cube3":
    x = torch.randint(0, 6, (M, 54), dtype=torch.uint8, device="cuda")
else:  # sliding puzzle: rows are permutations of the tiles
    D = model.state_dim
    x = torch.stack([torch.randperm(D) for _ in range(2048)]).to(torch.uint8).cuda().repeat(M // 2048, 1)
for _ in range(reps):
    y = fast(x)  # uint8 rows: layer-1 MFMA kernel where instantiated, then the f16x3 (fp32) / library (bf16) layers
torch.cuda.synchronize()
print("done", float(y.abs().max()))
