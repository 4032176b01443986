import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _dev_switches():
    """Bisecting aid (dev only): UDAPOSE_TEST_ARENA_TEACHER=1 runs the whole session with the bump-allocated arena for no-grad forwards
    instead of the forward-only plans (round 5's default)."""
    if os.environ.get("UDAPOSE_TEST_ARENA_TEACHER") == "1":
        from uda_poseestimation_amd.lib.models.pose_resnet import PoseResNet
        PoseResNet.fwd_only_plans = False
    yield


def train_keypoint_net(num_keypoints=16, steps=400, seed=0, lr=2e-4):
    """A TRAINED-LIKE PoseResNet-101: the reference initialisation trained on the device (fp16 student precision, device-side GradScaler,
    Adam, `steps` steps of 8 fresh images each from synthetic.keypoint_batch - images whose content determines the labels), so that the
    whole-network parity tests do not run on the worst case for 16-bit storage (a randomly initialised train-mode-BN ResNet).
    Returns (state_dict on the CPU, loss history, held-out eval-mode PCK@0.05).  With the deterministic weight-gradient accumulation
    (round 6, policy wgrad_det) every run of this function yields the SAME network, bit for bit."""
    import torch
    import uda_poseestimation_amd.lib.models as models
    from uda_poseestimation_amd import synthetic
    from uda_poseestimation_amd.lib import keypoint_detection as kd
    from uda_poseestimation_amd.lib.models.loss import JointsMSELoss
    from uda_poseestimation_amd.optim import FusedAdam
    torch.manual_seed(seed)
    net = models.pose_resnet101(num_keypoints=num_keypoints, pretrained_backbone=False).cuda().train()
    net.precision = "fp16"
    opt = FusedAdam(net.parameters(), lr=lr, dynamic_loss_scale=True, init_scale=1024.0)
    crit = JointsMSELoss()
    hist = []
    for it in range(steps):
        x, lab, wt = (t.cuda() for t in synthetic.keypoint_batch(8, num_keypoints=num_keypoints, seed=1000 + it))
        opt.zero_grad()
        loss = crit(net(x), lab, wt)
        opt.scale_loss(loss).backward()
        opt.step()
        if it % 80 == 0 or it == steps - 1:
            hist.append(float(loss.detach()))
    x, lab, wt = (t.cuda() for t in synthetic.keypoint_batch(8, num_keypoints=num_keypoints, seed=5))
    net.eval()
    with torch.no_grad():
        pck = float(kd.accuracy(net(x), lab)[1])
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    del net, opt
    torch.cuda.empty_cache()
    return sd, hist, pck


@pytest.fixture(scope="session")
def trained_r101_k16():
    """(state_dict, loss history, held-out PCK) of the trained-like PoseResNet-101, K = 16: shared by tests/test_gpu_trained.py and
    tests/test_gpu_fullsize.py (one training run per session, ~20 s)."""
    sd, hist, pck = train_keypoint_net(16)
    print("trained-like PoseResNet-101 (K=16): JointsMSE over the 400 steps " + " ".join(f"{h:.3e}" for h in hist) + f"; held-out PCK@0.05 (eval mode) {pck:.3f}")
    assert hist[-1] < 0.5 * hist[0] and pck > 0.5, (hist, pck)           # it learned the task, and generalises
    return sd, hist, pck
