"""GPU tests of the training step and the AVI driver: `nnet_utils.train_nnet` on the device against the
run recorded from the reference (tests/golden/train_nnet.npz), and one tiny end-to-end `ctg_approx/avi.py` loop."""
import os
import pickle
import random
import re
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _free_port() -> str:
    """A port nobody listens on right now (fixed ports collide when an earlier rendezvous still lingers)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])



@pytest.mark.parametrize("tag,bn", [("bn", True), ("nobn", False)])
def test_train_nnet_on_device_matches_reference_run(tag, bn):
    from deepcubea_amd import _lib
    from deepcubea_amd.utils import nnet_utils
    from deepcubea_amd.utils.pytorch_models import ResnetModel
    _lib.require_gpu()
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_nnet.npz"))
    net = ResnetModel(54, 6, 64, 32, 2, 1, bn)
    net.load_state_dict({k.split(":", 2)[2]: torch.tensor(g[k]) for k in g.files if k.startswith(tag + ":init:")})
    net = net.cuda()
    bs, itrs, itr0, lr, lr_d = g[tag + ":args"]
    np.random.seed(7)
    random.seed(7)
    x = torch.from_numpy(g[tag + ":x"]).cuda()  # device-resident training set, as Updater.update_dev returns it
    y = torch.from_numpy(g[tag + ":y"].astype(np.float32)).cuda()
    last = nnet_utils.train_nnet(net, x, y, torch.device("cuda"), int(bs), int(itrs), int(itr0), float(lr), float(lr_d),
                                 display=False)
    # tolerance: fp32 GEMMs on MFMA vs the reference's CPU run, 5-7 Adam steps
    assert abs(last - float(g[tag + ":last_loss"])) < 1e-3 * max(1.0, abs(last))
    for k, v in net.state_dict().items():
        if "num_batches_tracked" in k:
            assert int(v) == int(g["%s:final:%s" % (tag, k)])
            continue
        if bn and (re.fullmatch(r"(fc1|fc2|blocks\.\d\.[02])\.bias", k) or "running_mean" in k):
            # a Linear bias in front of BatchNorm has an analytically zero gradient: Adam turns its rounding noise into
            # +-lr steps (in the reference too), so these entries are noise by construction and do not affect the loss
            continue
        assert np.allclose(v.cpu().numpy(), g["%s:final:%s" % (tag, k)], rtol=2e-3, atol=2e-4), k


@pytest.mark.parametrize("m,k,n", [(1000, 324, 5000), (2048, 5000, 1000), (777, 1000, 1000)])
def test_linear_train_is_fp32_accurate_forward_and_backward(m, k, n):
    """_lib.linear_train (nn.Linear inside the training step: forward and input gradient on dca_f16x3_gemm, operands scaled
    by powers of two and split on the spot; weight gradient on the library) against float64.  fp32 accuracy means: as close
    to the float64 result as torch's own fp32 nn.Linear is (factor 4 allowed), forward and all three gradients — with
    gradients at 1e-7 scale and a 2^20 spread between rows, which an unscaled fp16 split would flush."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    g = torch.Generator().manual_seed(m + k + n)
    lin = torch.nn.Linear(k, n)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(n, k, generator=g) / k ** 0.5)
        lin.weight[::7] *= 1e-3  # rows of very different magnitude: the per-row scales
        lin.bias.copy_(torch.randn(n, generator=g))
    x = torch.relu(torch.randn(m, k, generator=g)) * 2.0
    dy = torch.randn(m, n, generator=g) * 1e-7
    dy *= torch.exp2(-20.0 * torch.rand(m, 1, generator=g))  # per-sample gradient magnitudes over 20 binades
    ref = torch.nn.Linear(k, n).double()
    ref.load_state_dict({kk: v.double() for kk, v in lin.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double())
    lin = lin.cuda()

    def run(fn):
        lin.zero_grad()
        xd = x.cuda().requires_grad_(True)
        yd = fn(xd)
        yd.backward(dy.cuda())
        return [t.double().cpu() for t in (yd.detach(), xd.grad, lin.weight.grad, lin.bias.grad)]

    ours = run(lambda t: _lib.linear_train(t, lin))
    lib32 = run(lambda t: torch.nn.functional.linear(t, lin.weight, lin.bias))
    want = [yr.detach(), xr.grad, ref.weight.grad, ref.bias.grad]
    for name, o, l, w_ in zip(("y", "dx", "dw", "db"), ours, lib32, want):
        e_ours, e_lib = float((o - w_).abs().max()), float((l - w_).abs().max())
        scale = float(w_.abs().max())
        assert e_ours <= max(4.0 * e_lib, 2e-7 * scale), (name, e_ours, e_lib, scale)
    # per-ROW accuracy of the input gradient: a sample whose gradient is 2^-20 of the largest keeps its own digits
    rows = (ours[1] - want[1]).abs().amax(dim=1) / want[1].abs().amax(dim=1).clamp_min(1e-300)
    assert float(rows.max()) < 1e-3 and float(rows.median()) < 1e-5, (float(rows.max()), float(rows.median()))


@pytest.mark.gpu
def test_linear_train_survives_activations_beyond_the_fp16_range():
    """the forward used to split activations into fp16 planes unscaled — a value past 65504 became inf and the
    loss NaN without a diagnostic.  Activations now carry a power-of-two scale from their own magnitude (like the
    gradients): outputs and all gradients stay finite and fp32-accurate with inputs of 3e5."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    g = torch.Generator().manual_seed(5)
    m, k, n = 512, 256, 128
    lin = torch.nn.Linear(k, n)
    x = torch.randn(m, k, generator=g)
    x[3, 5], x[100, 17] = 3.0e5, -2.0e5
    dy = torch.randn(m, n, generator=g) * 1e-3
    ref = torch.nn.Linear(k, n).double()
    ref.load_state_dict({kk: v.double() for kk, v in lin.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double())
    lin = lin.cuda()
    xd = x.cuda().requires_grad_(True)
    yd = _lib.linear_train(xd, lin)
    yd.backward(dy.cuda())
    assert torch.isfinite(yd).all() and torch.isfinite(xd.grad).all() and torch.isfinite(lin.weight.grad).all()
    y32 = torch.nn.functional.linear(x.cuda(), lin.weight, lin.bias).double().cpu()
    e_ours, e_lib = float((yd.detach().double().cpu() - yr.detach()).abs().max()), float((y32 - yr.detach()).abs().max())
    assert e_ours <= max(4.0 * e_lib, 2e-7 * float(yr.abs().max())), (e_ours, e_lib)


def test_avi_loop_end_to_end(tmp_path):
    """update (device) -> train -> save -> GBFS test -> target update, twice, with the reference's file layout and
    log lines (ctg_approx/avi.py:176-270)."""
    from deepcubea_amd.ctg_approx import avi
    save = str(tmp_path / "saved_models")
    argv = ["--env", "cube3", "--states_per_update", "3000", "--batch_size", "1000", "--nnet_name", "t", "--max_itrs", "6",
            "--loss_thresh", "1e9", "--back_max", "4", "--num_test", "60", "--save_dir", save, "--update_nnet_batch_size",
            "2000", "--max_update_steps", "2"]
    try:
        avi.main(argv)
    finally:
        sys.stdout = sys.__stdout__
    cur, targ = os.path.join(save, "t", "current"), os.path.join(save, "t", "target")
    assert pickle.load(open(os.path.join(cur, "train_itr.pkl"), "rb")) == 6
    assert pickle.load(open(os.path.join(cur, "update_num.pkl"), "rb")) == 2
    for d in (cur, targ):
        sd = torch.load(os.path.join(d, "model_state_dict.pt"), map_location="cpu")
        assert len(sd) == 72 and all(torch.isfinite(v.float()).all() for v in sd.values())
    assert os.path.isfile(os.path.join(save, "t", "args.pkl"))
    log = open(os.path.join(save, "t", "output.txt")).read()
    assert log.count("Updating cost-to-go with value iteration") == 2 and log.count("Updating target network") == 2
    assert "Using GBFS with 2 step(s) to add extra states to training set" in log  # second update: min(update_num+1, 2)
    assert len(re.findall(r"Itr: \d+, lr: [\d.E+-]+, loss: [\d.E+-]+, targ_ctg: [\d.-]+, nnet_ctg: [\d.-]+, Time: [\d.]+", log)) >= 1
    assert len(re.findall(r"Back Steps: \d+, %Solved: [\d.]+, avgSolveSteps: [\d.]+, CTG Mean\(Std/Min/Max\): ", log)) >= 5
    m = re.findall(r"Cost-to-go \(mean/min/max\): ([\d.]+)/([\d.]+)/([\d.]+)", log)
    # first update: all-zeros target -> every unsolved state backs up to exactly 1, solved ones to 0 (avi.py:219)
    assert len(m) == 2 and float(m[0][1]) == 0.0 and float(m[0][2]) == 1.0
    # resuming picks up the counters (avi.py:162-173)
    try:
        avi.main(argv[:9] + ["9"] + argv[10:])
    finally:
        sys.stdout = sys.__stdout__
    assert pickle.load(open(os.path.join(cur, "train_itr.pkl"), "rb")) == 9


@pytest.mark.parametrize("n,c,relu,use_skip", [(1003, 1000, True, False), (517, 72, True, True), (64, 37, False, False),
                                               (2, 5000, True, True), (10000, 1024, False, True)])
def test_bn_train_kernels_match_float64_reference(n, c, relu, use_skip):
    """csrc/dca_train.hip vs nn.BatchNorm1d(train) [+ skip] [+ ReLU] evaluated in float64: outputs, running statistics,
    and all gradients (dx, dskip, dgamma, dbeta)."""
    from deepcubea_amd import _lib
    torch.manual_seed(n + c)
    dev = "cuda"
    x = (torch.randn(n, c, device=dev) * 1.7 + 0.4).requires_grad_(True)
    skip = torch.randn(n, c, device=dev).requires_grad_(True) if use_skip else None
    bn = torch.nn.BatchNorm1d(c).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.2)
        bn.running_var.uniform_(0.5, 2.0)
    ref = torch.nn.BatchNorm1d(c).to(dev).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v.clone() for k, v in bn.state_dict().items()})
    xd = x.detach().double().requires_grad_(True)
    sd = skip.detach().double().requires_grad_(True) if use_skip else None
    yr = ref(xd)
    if use_skip:
        yr = yr + sd
    if relu:
        yr = torch.relu(yr)
    w = torch.randn(n, c, device=dev)
    (yr * w.double()).sum().backward()
    y = _lib.bn_train(x, bn, relu=relu, skip=skip)
    (y * w).sum().backward()
    tol = dict(rtol=2e-4, atol=2e-4)
    assert torch.allclose(y.double(), yr, **tol)
    assert torch.allclose(bn.running_mean.double(), ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.double(), ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    scale = max(1.0, float(xd.grad.abs().max()))
    assert torch.allclose(x.grad.double(), xd.grad, rtol=1e-3, atol=2e-4 * scale)
    assert torch.allclose(bn.weight.grad.double(), ref.weight.grad, rtol=1e-3, atol=1e-3 * max(1.0, float(ref.weight.grad.abs().max())))
    assert torch.allclose(bn.bias.grad.double(), ref.bias.grad, rtol=1e-3, atol=1e-3 * max(1.0, float(ref.bias.grad.abs().max())))
    if use_skip:
        assert torch.allclose(skip.grad.double(), sd.grad, **tol)


def test_avi_two_ranks_ddp(tmp_path):
    """N > 1 wiring of the AVI driver on the GPU box: two ranks under torch.distributed.run (sharing the one GPU here, so
    the process group is gloo; nccl = RCCL on a multi-GPU node), DistributedDataParallel training on per-rank shards,
    rank 0 writes the checkpoints."""
    import subprocess
    save = str(tmp_path / "saved_models")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), DCA_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", _free_port(), "-m", "deepcubea_amd.ctg_approx.avi", "--env", "puzzle15",
           "--states_per_update", "4000", "--batch_size", "1000", "--nnet_name", "p", "--max_itrs", "4", "--loss_thresh",
           "1e9", "--back_max", "6", "--num_test", "60", "--save_dir", save, "--update_nnet_batch_size", "2000"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    cur = os.path.join(save, "p", "current")
    assert pickle.load(open(os.path.join(cur, "train_itr.pkl"), "rb")) == 4  # 2000 states per rank / 500 per rank-batch
    assert pickle.load(open(os.path.join(cur, "update_num.pkl"), "rb")) == 1
    sd = torch.load(os.path.join(cur, "model_state_dict.pt"), map_location="cpu")
    assert all(not k.startswith("module.") for k in sd) and all(torch.isfinite(v.float()).all() for v in sd.values())
    log = open(os.path.join(save, "p", "output.txt")).read()
    assert "Training model for update number 0 for 4 iterations" in log and "Updating target network" in log


def test_avi_two_ranks_ddp_uneven_shards(tmp_path):
    """with --max_update_steps 3 the ranks' GBFS shards differ in size (trajectories end where an instance
    solves; puzzle15 at back_max 6 solves many) and 4001 states do not split evenly — the ranks must still agree on the
    number of DDP steps (all_reduce MIN of the shard sizes) instead of hanging in the gradient all-reduce."""
    import subprocess
    save = str(tmp_path / "saved_models")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), DCA_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", _free_port(), "-m", "deepcubea_amd.ctg_approx.avi", "--env", "puzzle15",
           "--states_per_update", "4001", "--batch_size", "1000", "--nnet_name", "q", "--max_itrs", "6", "--loss_thresh",
           "1e9", "--back_max", "6", "--num_test", "60", "--save_dir", save, "--update_nnet_batch_size", "2000",
           "--max_update_steps", "3", "--update_num", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    cur = os.path.join(save, "q", "current")
    itr = pickle.load(open(os.path.join(cur, "train_itr.pkl"), "rb"))
    assert itr >= 6
    log = open(os.path.join(save, "q", "output.txt")).read()
    assert "Using GBFS with 3 step(s)" in log and "Done" in log
