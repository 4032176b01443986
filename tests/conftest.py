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
    """uda_poseestimation_amd.synthetic.trained_like_state_dict: (state_dict on the CPU, loss history, held-out eval-mode PCK@0.05)."""
    from uda_poseestimation_amd import synthetic
    return synthetic.trained_like_state_dict(num_keypoints, steps, seed, lr)


@pytest.fixture(scope="session")
def trained_r101_k16():
    """(state_dict, loss history, held-out PCK) of the trained-like PoseResNet-101, K = 16: shared by tests/test_gpu_trained.py and
    tests/test_gpu_fullsize.py (one training run per session, ~20 s)."""
    sd, hist, pck = train_keypoint_net(16)
    print("trained-like PoseResNet-101 (K=16): JointsMSE over the 400 steps " + " ".join(f"{h:.3e}" for h in hist) + f"; held-out PCK@0.05 (eval mode) {pck:.3f}")
    assert hist[-1] < 0.5 * hist[0] and pck > 0.5, (hist, pck)           # it learned the task, and generalises
    return sd, hist, pck
