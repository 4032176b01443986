"""Target-view data pipeline, CPU oracle (test-only): what a DataLoader worker of the reference does per sample.

  * affine_view_ref     <- lib/transforms/keypoint_detection.py:137-167 (`affine`): torchvision F.affine on a PIL image =
                           PIL Image.transform(size, AFFINE, inverse matrix, NEAREST) + the key-point algebra + aug_param
  * color_jitter_ref    <- T.ColorJitter (train_human.py:68,75) on a PIL image = PIL.ImageEnhance Brightness / Contrast / Color
                           applied in the given order
  * resized_crop_ref    <- T.RandomResizedCrop's transform (lib/transforms/keypoint_detection.py:59-88,507-521): PIL crop + PIL
                           Image.resize(BILINEAR) + the key-point shift / scale
  * gaussian_blur_ref   <- T.GaussianBlur (lib/transforms/keypoint_detection.py:216-225) = PIL.ImageFilter.GaussianBlur(radius)
  * to_tensor_normalize_ref <- T.ToTensor + T.Normalize (train_human.py:52,70-71)
  * generate_target_ref lives in mean_teacher_ref.py (lib/datasets/util.py:12-70)

PIL (Pillow) is the reference's own dependency for these operations and is present in this image, so the image arithmetic
here IS PIL's (pinned by construction); torchvision is absent: the inverse-matrix formula (`_get_inverse_affine_matrix` with
centre (w/2, h/2) for PIL inputs) is restated from the published algorithm - the same a, b, c, d algebra as the reference's own
key-point transform (keypoint_detection.py:141-155), against which tests/test_oracle_transforms.py cross-checks it.
"""
import math

import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageFilter


def inverse_matrix_ref(width, height, angle, translate, scale, shear):
    rot, sx, sy = math.radians(angle), math.radians(shear[0]), math.radians(shear[1])
    cx, cy = width * 0.5, height * 0.5
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    M = [d, -b, 0.0, -c, a, 0.0]
    M = [v / scale for v in M]
    M[2] += M[0] * (-cx - tx) + M[1] * (-cy - ty)
    M[5] += M[3] * (-cx - tx) + M[4] * (-cy - ty)
    M[2] += cx
    M[5] += cy
    return M


def keypoints_affine_ref(kp, angle, shear_x, shear_y, trans_x, trans_y, scale, width, height):
    """keypoint_detection.py:141-165"""
    ang, sx, sy = np.deg2rad(angle), np.deg2rad(shear_x), np.deg2rad(shear_y)
    kp = np.copy(kp)
    a = np.cos(ang - sy) / np.cos(sy)
    b = (-np.cos(ang - sy) * np.tan(sx) / np.cos(sy) - np.sin(ang))
    c = np.sin(ang - sy) / np.cos(sy)
    d = (-np.sin(ang - sy) * np.tan(sx) / np.cos(sy) + np.cos(ang))
    R = np.array([[scale * a, scale * b], [scale * c, scale * d]])
    kp[:, 0] = kp[:, 0] - width / 2
    kp[:, 1] = kp[:, 1] - height / 2
    kp = np.matmul(R, kp.T).T
    kp[:, 0] = kp[:, 0] + width / 2
    kp[:, 1] = kp[:, 1] + height / 2
    kp[:, 0] = kp[:, 0] + trans_x
    kp[:, 1] = kp[:, 1] + trans_y
    return kp


def affine_view_ref(img_u8, kp, angle, shear_x, shear_y, trans_x, trans_y, scale):
    """img_u8 [H,W,3] uint8 -> (warped uint8 image, key points, aug_param)"""
    H, W, _ = img_u8.shape
    M = inverse_matrix_ref(W, H, angle, (trans_x, trans_y), scale, (shear_x, shear_y))
    out = np.array(Image.fromarray(img_u8).transform((W, H), Image.AFFINE, M, Image.NEAREST))
    aug_param = [-angle, [-trans_x, -trans_y], [-shear_x, -shear_y], 1.0 / scale]
    return out, keypoints_affine_ref(kp.astype(np.float64), angle, shear_x, shear_y, trans_x, trans_y, scale, W, H), aug_param


def color_jitter_ref(img_u8, ops, factors):
    """ops: 1 brightness, 2 contrast, 3 saturation (PIL.ImageEnhance.Color), applied in order"""
    im = Image.fromarray(img_u8)
    enh = {1: ImageEnhance.Brightness, 2: ImageEnhance.Contrast, 3: ImageEnhance.Color}
    for o, f in zip(ops, factors):
        if o:
            im = enh[o](im).enhance(f)
    return np.array(im)


def resized_crop_ref(img_u8, kp, top, left, h, w, size):
    """lib/transforms/keypoint_detection.py:66-88: crop (:59-64: F.crop = Image.crop((left, top, left + w, top + h)), key points shifted)
    then resize (:39-57: asserts a square image, F.resize(image, size, BILINEAR) = Image.resize((size, size), BILINEAR), key points
    scaled by size / width)."""
    img = Image.fromarray(np.ascontiguousarray(img_u8)).crop((left, top, left + w, top + h))
    kp = np.copy(kp).astype(np.float64)
    kp[:, 0] -= left
    kp[:, 1] -= top
    width, height = img.size
    assert width == height
    factor = float(size) / float(width)
    if width != size:
        img = img.resize((size, size), Image.BILINEAR)
    kp *= factor
    return np.asarray(img), kp


def gaussian_blur_ref(img_u8, radius):
    return np.array(Image.fromarray(img_u8).filter(ImageFilter.GaussianBlur(radius)))


def to_tensor_normalize_ref(img_u8, mean, std):
    t = torch.from_numpy(img_u8).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    m, s = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)
    return t.sub_(m[:, None, None]).div_(s[:, None, None])
