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
