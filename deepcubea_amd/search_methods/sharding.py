"""Per-instance sharding of test scrambles across ranks: replicas only, no data-path
collective.  The reference's only hook is `--start_idx` (astar.py:354,376); here state i goes to rank
i mod world and rank 0 merges the per-state results in state order.

Under `torch.distributed.run` the ranks rendezvous with gloo (CPU objects are all that is exchanged:
move lists, times, node counts); a single process needs no process group at all.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch


def world_info() -> Tuple[int, int]:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def init_from_env(backend: str = "gloo") -> Tuple[int, int]:
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).  One GPU per rank.
    The search only exchanges CPU objects (gloo); the training step passes backend="nccl" (= RCCL over xGMI) for
    its gradient all-reduce."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        if not dist.is_initialized():
            dist.init_process_group(backend, rank=rank, world_size=world)
        _count_sharers(world)
        if ranks_on_my_device() > 1:
            _share_gpu(rank)
    return world, rank


def ranks_on_my_device() -> int:
    """How many ranks of this node are mapped onto the GPU this rank uses (LOCAL_RANK % device_count).  The deployment
    model is one rank per GPU (nnet_utils.py:292-301 of the reference); more local ranks than GPUs — a test
    box, a partitioned node — share devices, and the engine has to know: its node pool may only take its share of the
    HBM, and the grid-wide refinement of giant tie bins assumes it has the GPU to itself."""
    if not torch.cuda.is_available():
        return 1
    if _sharers is not None:  # counted over the process group: (host, device) pairs equal to mine
        return _sharers
    if "LOCAL_WORLD_SIZE" not in os.environ:
        # srun / mpirun set RANK, WORLD_SIZE and LOCAL_RANK only: WORLD_SIZE counts every NODE's ranks, so guessing from
        # it would cut each pool to 1/nodes of its GPU.  Without a process group to ask, assume the deployment model.
        return 1
    local_world = int(os.environ["LOCAL_WORLD_SIZE"])
    ndev = max(1, torch.cuda.device_count())
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    mine = local_rank % ndev
    return max(1, sum(1 for r in range(local_world) if r % ndev == mine))


_sharers: Optional[int] = None


def _device_identity() -> tuple:
    """(host, PHYSICAL identity of the current device).  Slurm / srun on AMD bind GPUs through ROCR_VISIBLE_DEVICES: every
    task then sees its GPU as device 0 with HIP_ / CUDA_VISIBLE_DEVICES unset, so an index plus those two variables would make
    all ranks of a node compare equal.  The device's UUID — or its PCI address — says which GPU it is whatever
    the masking; the index and all three masks are only the last resort when the runtime reports neither."""
    import socket
    dev = int(torch.cuda.current_device())
    props = torch.cuda.get_device_properties(dev)
    uuid = getattr(props, "uuid", None)
    if uuid is not None and str(uuid).strip("0-") != "":
        return (socket.gethostname(), "uuid", str(uuid))
    pci = tuple(getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    if pci[1] is not None:
        return (socket.gethostname(), "pci", pci)
    return (socket.gethostname(), "index", dev, os.environ.get("ROCR_VISIBLE_DEVICES", ""),
            os.environ.get("HIP_VISIBLE_DEVICES", ""), os.environ.get("CUDA_VISIBLE_DEVICES", ""))


def count_sharers(world: int) -> int:
    """Public form of the count for drivers that set up their own process group (bench.py)."""
    _count_sharers(world)
    return ranks_on_my_device()


def _count_sharers(world: int) -> None:
    """After the rendezvous: every rank says which device of which host it uses; the ranks whose answer equals mine share
    my GPU.  Independent of the launcher's environment variables (torch.distributed.run, srun, mpirun)."""
    global _sharers
    if world <= 1 or not torch.cuda.is_available():
        return
    import torch.distributed as dist
    mine = _device_identity()
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    _sharers = max(1, sum(1 for e in everyone if e == mine))


_shared_noted = False


def _share_gpu(rank: int) -> None:
    """Several ranks on one GPU: select the engine's single-workgroup path for giant tie bins before any engine exists
    (`dca_debug_tune(5, 1)`; the grid-wide path needs every workgroup of a launch resident, which two processes on one
    device cannot promise each other — it would notice by itself at its first barrier and fall back, 0.25 s later)."""
    global _shared_noted
    from .. import _lib
    _lib.check(_lib.lib().dca_debug_tune(5, 1), "dca_debug_tune")
    if not _shared_noted:
        _shared_noted = True
        print("rank %d: %d ranks share this GPU — node pool sized to 1/%d of its memory, grid-wide tie refinement off"
              % (rank, ranks_on_my_device(), ranks_on_my_device()))


def shard_indices(n: int, world: int, rank: int) -> List[int]:
    """Instance i -> rank i mod world (round robin keeps the expensive scrambles spread out)."""
    return list(range(rank, n, world))


class WorkQueue:
    """Dynamic per-instance sharding: every rank draws the next unsolved state index from a shared counter kept in the
    process group's key-value store (`store.add` is atomic) — no collective, and a rank that drew a 40x harder scramble
    (results/cube3/output.txt spans 1.6e6..6.1e7 nodes) simply draws fewer of them.  world == 1: plain iteration."""

    _instances = 0  # queues are created in the same order on every rank: the counter gives each one its own store key

    def __init__(self, n: int, world: int, rank: int, key: str = "dca_next_state"):
        WorkQueue._instances += 1
        self.n, self.world, self.rank, self.key = int(n), world, rank, "%s/%d" % (key, WorkQueue._instances)
        self._local = 0
        self._store = None
        if world > 1:
            import torch.distributed as dist
            self._store = dist.distributed_c10d._get_default_store()

    def next(self, k: int = 1) -> List[int]:
        """Up to k fresh indices (empty list = queue drained)."""
        if self._store is None:
            first = self._local
            self._local += k
        else:
            first = int(self._store.add(self.key, k)) - k
        return list(range(first, min(first + k, self.n)))


def gather_results(local: Dict[int, tuple], n: int, world: int, rank: int) -> Optional[Dict[int, tuple]]:
    """Merge per-rank {state index: result} dicts on rank 0 (None elsewhere)."""
    if world == 1:
        assert sorted(local) == list(range(n))
        return local
    import torch.distributed as dist
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(local, bucket, dst=0)
    if rank != 0:
        return None
    merged: Dict[int, tuple] = {}
    for part in bucket:
        for k, v in part.items():
            assert k not in merged, "state %d solved twice" % k
            merged[k] = v
    assert sorted(merged) == list(range(n)), "missing states after merge"
    return merged


def finalize() -> None:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
