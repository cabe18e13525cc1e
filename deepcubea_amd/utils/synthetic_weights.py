"""Deterministic synthetic weights for the cost-to-go network.

The reference's trained checkpoints (`saved_models/*/model_state_dict.pt`) are absent from the mount
(.MISSING_LARGE_BLOBS), so throughput runs and parity fixtures use weights regenerated from a seed with
NumPy's PCG64 — both sides can rebuild the same 58 MB without shipping it.  Same recipe as
tests/golden/make_golden.py:det_weights (state-dict order)."""
import numpy as np
import torch


def load_synthetic_weights(model: torch.nn.Module, seed: int) -> None:
    rng = np.random.default_rng(seed)
    new = {}
    for k, v in model.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            new[k] = torch.tensor(7, dtype=torch.long)
        elif k.endswith("running_var"):
            new[k] = torch.tensor(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif k.endswith("running_mean"):
            new[k] = torch.tensor(rng.normal(0, 0.1, shp).astype(np.float32))
        elif k.endswith("weight") and len(shp) == 2:
            new[k] = torch.tensor((rng.normal(0, 1.0, shp) / np.sqrt(shp[1])).astype(np.float32))
        elif k.endswith("weight"):
            new[k] = torch.tensor(rng.uniform(0.8, 1.2, shp).astype(np.float32))
        else:
            new[k] = torch.tensor(rng.normal(0, 0.1, shp).astype(np.float32))
    model.load_state_dict(new)
