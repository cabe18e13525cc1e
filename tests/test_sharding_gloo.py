"""world_size-2 gloo test (CPU) of the N>1 path: per-instance sharding + merge on rank 0.  The solver is the
CPU oracle here (tests may use it); on the GPU box the same code path is driven by bwas_hip."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from deepcubea_amd.search_methods import sharding
    from oracle import c_oracle as co
    w, r = sharding.init_from_env()
    assert (w, r) == (world, rank)
    roots = []
    scr = [[0, 5, 7, 2], [1, 3, 8], [4, 9, 1], [6], [], [11, 2, 6], [3, 3]]
    for mv in scr:
        s = np.arange(54, dtype=np.uint8)[None]
        for a in mv:
            s = co.next_state("cube3", s, a)
        roots.append(s[0])
    mine = sharding.shard_indices(len(roots), w, r)
    assert mine == list(range(rank, len(roots), world))
    # the default of the CLI: a shared work queue (atomic counter in the process group's store)
    queue = sharding.WorkQueue(len(roots), w, r)
    mine = []
    while True:
        got = queue.next(2)
        if not got:
            break
        mine += got
    local = {}
    for i in mine:
        res = co.astar("cube3", roots[i], 0.8, 50, co.SEM_PY, heur_builtin_id=1)
        local[i] = (res["moves"], None, 0.0, res["nodes_generated"])
    merged = sharding.gather_results(local, len(roots), w, r)
    if r == 0:
        assert sorted(merged) == list(range(len(roots)))
        np.save(out_path, np.array([merged[i][3] for i in range(len(roots))], np.int64))
        for i, mv in enumerate(scr):
            s = roots[i][None].copy()
            for a in merged[i][0]:
                s = co.next_state("cube3", s, a)
            assert co.is_solved("cube3", s)[0]
    else:
        assert merged is None
    sharding.finalize()


def test_two_rank_sharding_and_merge(tmp_path):
    from oracle import c_oracle as co
    co.lib()  # build once before forking workers
    out = str(tmp_path / "nodes.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    nodes2 = np.load(out)
    # single-process path gives the same merged answer
    from deepcubea_amd.search_methods import sharding
    assert sharding.shard_indices(7, 1, 0) == list(range(7))
    assert sharding.gather_results({i: (None,) for i in range(3)}, 3, 1, 0) == {i: (None,) for i in range(3)}
    assert nodes2.shape == (7,) and nodes2[4] == 12  # solved root: one expansion
    q = sharding.WorkQueue(5, 1, 0)
    assert [q.next(2), q.next(2), q.next(2), q.next(2)] == [[0, 1], [2, 3], [4], []]


def test_cli_rejects_other_languages(tmp_path):
    import pickle
    import pytest
    from deepcubea_amd.search_methods import astar
    p = tmp_path / "s.pkl"
    pickle.dump({"states": []}, open(p, "wb"))
    # argparse prefix matching like the reference's train.sh (`--model` for `--model_dir`)
    args = astar.build_parser().parse_args(["--states", str(p), "--model", "synthetic:1", "--env", "cube3",
                                            "--results_dir", str(tmp_path / "r")])
    assert args.model_dir == "synthetic:1" and args.language == "hip" and args.batch_size == 1 and args.weight == 1.0
    with pytest.raises(ValueError, match="Unknown language python"):
        astar.main(["--states", str(p), "--model_dir", "x", "--env", "cube3", "--results_dir", str(tmp_path / "r"),
                    "--language", "python", "--debug"])


# ---- training step under DistributedDataParallel (the only collective of the framework) --------------------------
def _train_setup():
    import torch
    from deepcubea_amd.utils.pytorch_models import ResnetModel
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_nnet.npz"))
    net = ResnetModel(54, 6, 64, 32, 2, 1, False)
    net.load_state_dict({k.split(":", 2)[2]: torch.tensor(g[k]) for k in g.files if k.startswith("nobn:init:")})
    perm = np.random.default_rng(3).permutation(40)
    batches = [perm[s:s + 8] for s in range(0, 40, 8)]  # 5 global batches of 8
    return net, g["nobn:x"], g["nobn:y"], batches


def _train_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import random
    import torch
    from deepcubea_amd.search_methods import sharding
    from deepcubea_amd.utils import nnet_utils
    torch.set_num_threads(1)
    w, r = sharding.init_from_env("gloo")
    net, x, y, batches = _train_setup()
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    random.seed(3)
    half = [b[r * 4:(r + 1) * 4] for b in batches]  # this rank's share of every global batch
    nnet_utils.train_nnet(ddp, [x], y, torch.device("cpu"), 4, 7, 2, 0.01, 0.95, display=False, batches_idx=half)
    if r == 0:
        np.savez(out_path, **{k: v.numpy() for k, v in net.state_dict().items()})
    sharding.finalize()


def test_train_nnet_ddp_two_ranks_equals_full_batch(tmp_path):
    """2 ranks x half batches under DDP (gradient all-reduce; gloo here, RCCL on the GPUs) == one process on the full
    batches: the arithmetic of the reference's nn.DataParallel training (nnet_utils.py:53-118, avi.py:207-208)."""
    import random
    import torch
    from deepcubea_amd.utils import nnet_utils
    torch.set_num_threads(1)
    out = str(tmp_path / "ddp.npz")
    mp.spawn(_train_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    net, x, y, batches = _train_setup()
    random.seed(3)
    nnet_utils.train_nnet(net, [x], y, torch.device("cpu"), 8, 7, 2, 0.01, 0.95, display=False, batches_idx=batches)
    for k, v in net.state_dict().items():
        assert np.allclose(got[k], v.numpy(), rtol=1e-5, atol=1e-6), k


def _queue_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import time
    from deepcubea_amd.search_methods import sharding
    w, r = sharding.init_from_env()
    n = 100
    # per-state cost varies 40x (results/cube3/output.txt: 1.6e6 .. 6.1e7 nodes): state i "costs" 1 or 40 ticks
    cost = [40 if i % 9 == 0 else 1 for i in range(n)]
    queue = sharding.WorkQueue(n, w, r)
    mine, busy = [], 0.0
    while True:
        got = queue.next(1)
        if not got:
            break
        mine += got
        time.sleep(cost[got[0]] * 0.002)
        busy += cost[got[0]] * 0.002
    local = {i: ([i], None, 0.0, i * i) for i in mine}
    merged = sharding.gather_results(local, n, w, r)
    np.save(os.path.join(out_dir, "rank%d.npy" % r), np.array([len(mine), int(busy * 1000)], np.int64))
    if r == 0:
        assert sorted(merged) == list(range(n)) and all(merged[i][3] == i * i for i in range(n))
    sharding.finalize()


def test_work_queue_world_8_balances_skewed_costs(tmp_path):
    """The shape of the 8-GPU run (configs[3]: 1000 scrambles over 8 ranks), on 8 gloo ranks: every state is drawn exactly
    once, rank 0 merges all of them in order, and the queue — unlike `i mod N` — evens out a 40x spread in per-state cost."""
    world = 8
    mp.spawn(_queue_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    stats = np.array([np.load(str(tmp_path / ("rank%d.npy" % r))) for r in range(world)])
    assert stats[:, 0].sum() == 100 and stats[:, 0].min() >= 1
    busy = stats[:, 1].astype(np.float64)
    # static round robin would give rank 0 (states 0, 8, 16, ...: five of the expensive ones) ~2.4x the mean; the queue
    # keeps every rank within one expensive state of the mean
    assert busy.max() <= busy.mean() + 40 * 2 + 20, busy
