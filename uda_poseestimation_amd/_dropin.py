"""Drop-in binding of this package under the reference's own module names.

The reference's scripts reach the hot path with (train_human.py:19-29)

    import lib.models as models
    from lib.models.loss import JointsMSELoss, ConsLoss
    from lib.keypoint_detection import accuracy
    from lib.models import Style_net
    from utils import *

With THIS directory first on sys.path those statements find `lib/__init__.py` and `utils.py` of this package as
top-level modules.  Both hand over to here: the real modules are imported ONCE under their package names
(`uda_poseestimation_amd.lib.models`, ...) and registered in sys.modules under the reference's names as well, so the two
spellings are the same module objects (one set of classes, one pair of Style_net singletons, one loaded .so).

    import sys; sys.path.insert(0, "/path/to/repo/uda_poseestimation_amd")
    import _dropin; _dropin.install("/path/to/UDA_PoseEstimation")     # optional 2nd line: lib.datasets / lib.transforms /
                                                                        # lib.data / lib.meter / lib.logger stay the reference's
"""
import importlib
import os
import sys

_PKG = "uda_poseestimation_amd"
_HERE = os.path.dirname(os.path.abspath(__file__))
# reference name -> module of this package
ALIASES = {
    "lib": _PKG + ".lib",
    "lib.models": _PKG + ".lib.models",
    "lib.models.pose_resnet": _PKG + ".lib.models.pose_resnet",
    "lib.models.resnet": _PKG + ".lib.models.resnet",
    "lib.models.loss": _PKG + ".lib.models.loss",
    "lib.models.Style_net": _PKG + ".lib.models.Style_net",
    "lib.keypoint_detection": _PKG + ".lib.keypoint_detection",
    "utils": _PKG + ".utils",
}


def _package():
    """Import the package under its own name (its parent directory goes to the END of sys.path if needed)."""
    root = os.path.dirname(_HERE)
    if _PKG not in sys.modules and root not in sys.path:
        sys.path.append(root)
    return importlib.import_module(_PKG)


def alias(names=None):
    """Register the reference's module names for this package's modules; returns {reference name: module}."""
    _package()
    out = {}
    for ref_name in (names or ALIASES):
        mod = importlib.import_module(ALIASES[ref_name])
        sys.modules[ref_name] = mod
        out[ref_name] = mod
    return out


def install(reference_root=None):
    """alias() + let `lib.datasets`, `lib.transforms`, `lib.data`, `lib.meter`, `lib.logger` resolve from the reference
    checkout (its `lib/` has no __init__.py: it is merged into this package's `lib.__path__`)."""
    mods = alias()
    if reference_root is not None:
        ref_lib = os.path.join(os.path.abspath(reference_root), "lib")
        if not os.path.isdir(ref_lib):
            raise FileNotFoundError(f"{ref_lib}: not a UDA_PoseEstimation checkout")
        if ref_lib not in mods["lib"].__path__:
            mods["lib"].__path__.append(ref_lib)
    return mods
