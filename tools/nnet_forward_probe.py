"""Forward time of the full heuristic network (FastResnet) at 409 600 rows by mode and layer-1 kernel (l1 = mfma | embed)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from deepcubea_amd.utils import env_utils
from deepcubea_amd.utils.pytorch_models import FastResnet
from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
for envn in ("puzzle48", "puzzle15", "cube3"):
    env = env_utils.get_environment(envn)
    model = env.get_nnet_model(); load_synthetic_weights(model, 2024)
    D = {"puzzle48": 49, "puzzle15": 16, "cube3": 54}[envn]
    M = 409600
    x = torch.stack([torch.randperm(D) for _ in range(2048)]).to(torch.uint8).cuda().repeat(M // 2048, 1) if envn != "cube3" else torch.randint(0, 6, (M, 54), dtype=torch.uint8, device="cuda")
    for dt, g16, l1 in ((torch.bfloat16, "hip", "mfma"), (torch.bfloat16, "hip", "embed"), (torch.bfloat16, "library", "mfma"),
                        (torch.float32, "hip", "mfma"), (torch.float32, "hip", "embed")):
        f = FastResnet(model, dt, gemm16=g16, l1=l1).cuda()
        for _ in range(2): f(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f(x)
        torch.cuda.synchronize(); dt_ms = (time.perf_counter() - t0) / 5 * 1e3
        print(envn, dt, g16, "l1", l1, "uses_l1", f.uses_l1_kernel, "%.2f ms per %d rows" % (dt_ms, M), "%.3e rows/s" % (M / dt_ms * 1e3))
