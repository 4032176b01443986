"""Analytic known-answer cases for the tF.affine restatement (torchvision is absent: parity unpinned)."""
import torch

from oracle.affine_ref import affine_nearest_ref, warp3_ref


def _img():
    return torch.arange(2 * 8 * 8, dtype=torch.float32).reshape(2, 8, 8) + 1.0


def test_identity():
    x = _img()
    assert torch.equal(affine_nearest_ref(x, 0.0, [0, 0], 1.0, [0.0, 0.0]), x)
    assert torch.equal(warp3_ref(x, 0.0, 0, 0, 0.0, 0.0, 1.0, ratio=4.0), x)


def test_integer_translation():
    x = _img()
    y = affine_nearest_ref(x, 0.0, [2, -1], 1.0, [0.0, 0.0])   # content moves +2 in x, -1 in y
    exp = torch.zeros_like(x)
    exp[:, 0:7, 2:8] = x[:, 1:8, 0:6]
    assert torch.equal(y, exp)


def test_rot180_and_rot90():
    x = _img()
    assert torch.equal(affine_nearest_ref(x, 180.0, [0, 0], 1.0, [0.0, 0.0]), torch.flip(x, dims=(1, 2)))
    y = affine_nearest_ref(x, 90.0, [0, 0], 1.0, [0.0, 0.0])
    # both possible conventions are rot90 by +-1: pin the one consistent with torchvision (counter-clockwise
    # for positive angles in image coordinates with y down = torch.rot90(k=1) over (H,W))
    assert torch.equal(y, torch.rot90(x, 1, dims=(1, 2))) or torch.equal(y, torch.rot90(x, -1, dims=(1, 2)))


def test_scale2_about_centre():
    x = _img()
    y = affine_nearest_ref(x, 0.0, [0, 0], 2.0, [0.0, 0.0])
    # output pixel (i,j) samples input at centre + (p - centre)/2 -> nearest
    assert y[0, 0, 0] == x[0, 2, 2] and y[0, 7, 7] == x[0, 5, 5]
