"""torchvision.transforms.functional.affine on tensors (nearest, zero fill), CPU oracle (test-only).

torchvision is NOT in the reference tree nor in this image: restated from the published algorithm of
torchvision 0.8-0.12.  It cannot be run against torchvision itself (-> still "parity unpinned" at
that boundary); what the reference's OWN code pins is checked in tests/test_oracle_affine.py: identity / translation /
180-degree / scale cases, the rotation, shear and translation DIRECTIONS against the reference's key-point algebra
(lib/transforms/keypoint_detection.py:141-165: content and key points move together), and the loop's three-warp inverse
(train_human.py:366-368) against a forward warp by PIL, the reference's image backend.
Call sites it stands in for: train_human.py:366-368, 388-390, 412, 421-423.
"""
import math
import torch
import torch.nn.functional as F


def inverse_affine_matrix(angle, translate, scale, shear):
    rot = math.radians(angle)
    sx = math.radians(shear[0])
    sy = math.radians(shear[1])
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0]
    m[2] += m[0] * (-tx) + m[1] * (-ty)
    m[5] += m[3] * (-tx) + m[4] * (-ty)
    return m


def affine_nearest_ref(img, angle, translate, scale, shear):
    """img [C,H,W] float -> [C,H,W]."""
    C, H, W = img.shape
    m = inverse_affine_matrix(angle, translate, scale, shear)
    theta = torch.tensor(m, dtype=torch.float32).reshape(1, 2, 3)
    xs = torch.linspace(-W * 0.5 + 0.5, W * 0.5 - 0.5, W)
    ys = torch.linspace(-H * 0.5 + 0.5, H * 0.5 - 0.5, H)
    base = torch.empty(1, H, W, 3)
    base[..., 0] = xs[None, None, :]
    base[..., 1] = ys[None, :, None]
    base[..., 2] = 1
    resc = theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H])
    grid = base.reshape(1, H * W, 3).bmm(resc).reshape(1, H, W, 2)
    out = F.grid_sample(img[None].float(), grid, mode="nearest", padding_mode="zeros", align_corners=False)
    return out[0].to(img.dtype)


def warp3_ref(img, angle, tx, ty, shear_x, shear_y, scale, ratio=1.0):
    """The three sequential nearest warps of train_human.py:366-368."""
    t = affine_nearest_ref(img, 0.0, [tx / ratio, ty / ratio], 1.0, [0.0, 0.0])
    t = affine_nearest_ref(t, angle, [0.0, 0.0], scale, [0.0, 0.0])
    return affine_nearest_ref(t, 0.0, [0.0, 0.0], 1.0, [shear_x, shear_y])
