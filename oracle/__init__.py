"""CPU parity oracle for the mean-teacher UDA pose-estimation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product package (``uda_poseestimation_amd``) never
imports this package and raises when its HIP library is missing.

Every function here is a plain torch/numpy CPU restatement of one row of SURVEY.md §8(a), with
the reference file:line it follows in its docstring.

Pinning status:
  * losses, decode, PCK, rectify, EMA, AdaIN statistics, Style net, label generator:
    pinned against outputs of the reference's own Python, captured in this repo's
    ``tests/golden/*.npz`` by ``tests/golden/make_golden.py`` (run where /root/reference exists).
  * ResNet trunk (torchvision ``models.ResNet``/``Bottleneck``) and ``tF.affine``:
    torchvision is a third-party dependency that is absent from the reference tree and from this
    image, version unpinned (API usage brackets it to 0.8-0.12).  These two are restated from the
    published algorithm and validated on analytic cases and on the reference's own
    ``Upsampling``/``PoseResNet`` wrapper code -> **parity unpinned** at that boundary.  For ``tF.affine`` the direction
    conventions are additionally pinned against the reference's own key-point algebra and against PIL (forward warp by PIL,
    inverse by the restated three-warp chain): tests/test_oracle_affine.py.
  * data-pipeline transforms (``F.affine`` on PIL images, ``ColorJitter``): the oracle IS PIL (Pillow is the reference's own
    dependency for them and is present in this image): oracle/transforms_ref.py.
"""
