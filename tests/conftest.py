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
