"""CPU checks of the target-view data pipeline: the oracle (oracle/transforms_ref.py, PIL's own arithmetic)
against analytic cases and against the reference's own key-point algebra, and the product's HOST-side helpers
(uda_poseestimation_amd/data_gpu.py: inverse matrix, PIL's fixed-point coefficients, key-point transform, parameter draws)
against the oracle - so the GPU kernels (tests/test_gpu_data.py) only have to reproduce integer index / blend arithmetic."""
import math
import random

import numpy as np
import pytest

from oracle import transforms_ref as R
from uda_poseestimation_amd import data_gpu as D


def _emulate_affine(img, coef):
    """numpy statement of what csrc/augment.hip aug_affine_u8_k computes from the fixed-point coefficients"""
    H, W, _ = img.shape
    a0, a1, a2, a3, a4, a5 = coef
    ys, xs = np.mgrid[0:H, 0:W].astype(np.int64)
    xin, yin = (a2 + xs * a0 + ys * a1) >> 16, (a5 + xs * a3 + ys * a4) >> 16
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def test_affine_identity_translation_and_rotation_sense():
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    kp = np.array([[10.0, 20.0], [40.0, 5.0]])
    out, k2, aug = R.affine_view_ref(img, kp, 0.0, 0.0, 0.0, 0, 0, 1.0)
    assert np.array_equal(out, img) and np.allclose(k2, kp) and aug == [-0.0, [0, 0], [-0.0, -0.0], 1.0]
    out, k2, _ = R.affine_view_ref(img, kp, 0.0, 0.0, 0.0, 5, -3, 1.0)      # translate right 5, up 3
    assert np.array_equal(out[0:61, 5:64], img[3:64, 0:59]) and (out[:, :5] == 0).all() and (out[61:, :] == 0).all()
    assert np.allclose(k2, kp + [5, -3])
    # rotation sense: the image content and the key points must move TOGETHER (a bright pixel at a key point lands at the
    # transformed key point), which pins the sign conventions of the restated inverse matrix against the reference's own algebra
    for angle, shx, sc in ((90.0, 0.0, 1.0), (-37.0, 12.0, 0.8), (141.0, -25.0, 1.25)):
        blank = np.zeros((128, 128, 3), np.uint8)
        pts = np.array([[80.0, 52.0], [40.0, 70.0], [64.0, 90.0]])
        for (x, y) in pts.astype(int):
            blank[y - 1:y + 2, x - 1:x + 2] = 255
        out, k2, _ = R.affine_view_ref(blank, pts, angle, shx, 0.0, 4, -6, sc)
        for (x, y) in k2:
            xi, yi = int(round(x)), int(round(y))
            assert out[yi - 2:yi + 3, xi - 2:xi + 3].max() == 255, (angle, x, y)


def test_host_helpers_match_oracle_and_pil_fixed_point():
    rs = np.random.RandomState(1)
    for it in range(60):
        H = W = int(rs.choice([64, 96, 256]))
        img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
        kp = rs.uniform(0, W, (16, 2))
        angle, shx = rs.uniform(-180, 180), rs.uniform(-30, 30)
        tx, ty, sc = int(round(rs.uniform(-12.8, 12.8))), int(round(rs.uniform(-12.8, 12.8))), rs.uniform(0.6, 1.3)
        ref, kref, aug = R.affine_view_ref(img, kp, angle, shx, 0.0, tx, ty, sc)
        m = D.inverse_affine_matrix_pil((W * 0.5, H * 0.5), angle, (tx, ty), sc, (shx, 0.0))
        assert np.allclose(m, R.inverse_matrix_ref(W, H, angle, (tx, ty), sc, (shx, 0.0)), rtol=0, atol=0)
        assert np.array_equal(_emulate_affine(img, D.pil_fixed_coefficients(m)), ref)          # bit-exact with PIL's transform
        assert np.allclose(D.transform_keypoints(kp, angle, shx, 0.0, tx, ty, sc, W, H), kref, rtol=0, atol=1e-12)


def test_parameter_draws_follow_the_reference_order_and_ranges():
    cfg = D.ViewConfig(rotation=60, shear=(-30, 30), translate=(0.05, 0.05), scale=(0.6, 1.3), color=0.25)
    a = cfg.draw_affine(random.Random(5), (256, 256))
    r = random.Random(5)          # RandomAffineRotation.get_params: angle, shear_x, trans_x, trans_y, scale
    exp = (r.uniform(-60, 60), r.uniform(-30, 30), 0.0, int(round(r.uniform(-12.8, 12.8))), int(round(r.uniform(-12.8, 12.8))), r.uniform(0.6, 1.3))
    assert a == exp
    ops, fs = cfg.draw_jitter(random.Random(6))
    assert sorted(ops) == [1, 2, 3] and all(0.75 <= f <= 1.25 for f in fs)


def test_color_jitter_blend_semantics():
    """ImageEnhance = blend(degenerate, image, f) in float32, clipped, TRUNCATED - what aug_color_op_k implements"""
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, (32, 32, 3)).astype(np.uint8)

    def lum(im):
        r, g, b = (im[..., i].astype(np.int64) for i in range(3))
        return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16

    def blend(deg, im, f):
        t = deg.astype(np.float32) + np.float32(f) * (im.astype(np.float32) - deg.astype(np.float32))
        return np.clip(t, 0, 255).astype(np.uint8)
    for f in (0.75, 0.9, 1.0, 1.13, 1.25):
        L = lum(img)
        mean = int(L.astype(np.float64).sum() / L.size + 0.5)
        assert np.array_equal(R.color_jitter_ref(img, [1], [f]), blend(np.zeros_like(img), img, f))
        assert np.array_equal(R.color_jitter_ref(img, [2], [f]), blend(np.full_like(img, mean), img, f))
        assert np.array_equal(R.color_jitter_ref(img, [3], [f]), blend(np.repeat(L[..., None], 3, -1).astype(np.uint8), img, f))


def _emulate_resample(img, bounds, coef):
    """the two passes of aug_resized_crop_u8 in numpy (what resample_u8_k computes), for a square crop `img`"""
    def one(a):             # along axis 1
        out = np.zeros((a.shape[0], bounds.shape[0], 3), np.uint8)
        for o, (first, cnt) in enumerate(bounds):
            s = (a[:, first:first + cnt].astype(np.int64) * coef[o, :cnt, None]).sum(1) + (1 << 21)
            out[:, o] = np.clip(s >> 22, 0, 255)
        return out
    return one(one(img).transpose(1, 0, 2)).transpose(1, 0, 2)


def test_resized_crop_host_tables_match_pil_bilinear_resize():
    """data_gpu.pil_resample_coeffs (the tables the device kernel walks) against PIL's own Image.resize(BILINEAR): reductions (antialiased,
    support > 1), enlargements, the identity; and the key-point algebra / parameter draws of RandomResizedCrop."""
    rs = np.random.RandomState(11)
    raw = rs.randint(0, 256, (96, 96, 3)).astype(np.uint8)
    kp = rs.uniform(0, 96, (5, 2))
    for (top, left, w) in ((0, 0, 96), (3, 7, 64), (10, 2, 75), (20, 30, 37), (0, 0, 64), (5, 5, 90)):
        ref, kref = R.resized_crop_ref(raw, kp, top, left, w, w, 64)
        bounds, coef, ksize = D.pil_resample_coeffs(w, 64)
        assert ksize == int(math.ceil(max(w / 64.0, 1.0))) * 2 + 1
        assert np.array_equal(_emulate_resample(raw[top:top + w, left:left + w], bounds, coef), ref), (top, left, w)
        k2 = kp.copy(); k2[:, 0] -= left; k2[:, 1] -= top; k2 *= 64.0 / w
        assert np.array_equal(k2, kref)
    # RandomResizedCrop.get_params: uniform area, two randint corners, ten attempts, whole-image fallback
    box = D.draw_resized_crop(random.Random(3), 512, 512, (0.6, 1.3))
    r = random.Random(3)
    exp = None
    for _ in range(10):
        side = int(round(math.sqrt(r.uniform(0.6, 1.3) * 512 * 512)))
        if 0 < side <= 512:
            exp = (r.randint(0, 512 - side), r.randint(0, 512 - side), side, side)
            break
    assert box == (exp or (0, 0, 512, 512))
    assert D.draw_resized_crop(random.Random(0), 100, 100, (1.5, 2.0)) == (0, 0, 100, 100)
