"""Workload for rocprofv3: the cube3 cost-to-go network (FastResnet layout) on one dedup-first batch of 204 800 rows,
fp32 (parity mode), bf16 or fp8 (Fp8Resnet).  `python tools/profile_nnet.py fp32|bf16|fp8 [reps]`"""
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
model = env_utils.get_environment("cube3").get_nnet_model()
load_synthetic_weights(model, 2024)
fast = (Fp8Resnet(model) if name == "fp8" else FastResnet(model, dt)).cuda()
x = torch.randint(0, 6, (M, 54), dtype=torch.uint8, device="cuda")
for _ in range(reps):
    y = fast(x)  # uint8 rows: layer-1 MFMA kernel where instantiated, then the f16x3 (fp32) / library (bf16) layers
torch.cuda.synchronize()
print("done", float(y.abs().max()))
