"""Worker of tests/test_world8_hip.py: one PROCESS pinned to a GPU the way the reference pins its per-GPU workers — the
visibility mask is written INSIDE the spawned process before the first device call, and the process then addresses `cuda:0`
and nothing else (utils/nnet_utils.py:208-209 `os.environ['CUDA_VISIBLE_DEVICES'] = str(gpu_num)`, 292-301 one process per
GPU).  On ROCm the mask is HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is honoured too).  Writes what it saw as JSON.
Run as a script; not collected by pytest."""
import json
import os
import sys


def main():
    out_path, var, gpu_num = sys.argv[1], sys.argv[2], sys.argv[3]
    os.environ[var] = gpu_num  # before torch / HIP initialise
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import numpy as np
    import torch
    rec = {"var": var, "gpu_num": gpu_num, "device_count": int(torch.cuda.device_count())}
    if rec["device_count"] == 0:
        # no device behind the mask: the product must refuse loudly, not fall back to anything
        from deepcubea_amd import _lib
        try:
            _lib.require_gpu()
            rec["require_gpu"] = "passed"
        except Exception as e:  # noqa: BLE001
            rec["require_gpu"] = type(e).__name__
        json.dump(rec, open(out_path, "w"))
        return
    from deepcubea_amd import _lib
    from deepcubea_amd.search_methods import sharding
    from deepcubea_amd.search_methods.engine import BwasEngine
    from oracle import c_oracle as co
    rec["current_device"] = int(torch.cuda.current_device())
    rec["identity"] = [str(x) for x in sharding._device_identity()[1:]]
    s = np.arange(54, dtype=np.uint8)[None]
    for a in (0, 5, 7, 2):
        s = co.next_state("cube3", s, a)
    eng = BwasEngine("cube3", 0.8, 50, max_nodes=1 << 16)
    res = eng.solve_builtin(s[0], _lib.HEUR_MOD97)
    ref = co.astar("cube3", s[0], 0.8, 50, co.SEM_PY, heur_builtin_id=0)
    rec["solved"] = bool(res["solved"]) and res["moves"] == ref["moves"] and res["nodes_generated"] == ref["nodes_generated"]
    # every allocation of the engine sits on the one visible device: cuda:0
    rec["alloc_device"] = int(torch.empty(1, device="cuda").device.index)
    eng.close()
    json.dump(rec, open(out_path, "w"))


if __name__ == "__main__":
    main()
