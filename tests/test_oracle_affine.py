"""Known-answer cases for the tF.affine restatement.  torchvision is absent (its version is unpinned by the reference), so the
restatement cannot be run against torchvision itself; it is pinned as far as the reference's OWN code allows: the sign /
direction conventions against the reference's key-point algebra (lib/transforms/keypoint_detection.py:141-165, restated in
oracle/transforms_ref.py) - image content and key points must move together, the training labels depend on it - and the
loop's inverse re-warp (train_human.py:366-368) against a forward warp done by PIL, the reference's image backend."""
import numpy as np
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
    # positive angles turn the content CLOCKWISE on the screen (y down): the sense in which the reference's own key-point
    # transform moves the key points (x' = cos x - sin y, y' = sin x + cos y about the centre: right-of-centre -> below-centre)
    assert torch.equal(y, torch.rot90(x, -1, dims=(1, 2)))


def test_content_moves_with_the_references_keypoint_transform():
    """A blob drawn at key point p lands at keypoints_affine_ref(p) for rotation, scale, shear and translation: pins the
    direction conventions of the restated inverse matrix against the reference's own algebra."""
    from oracle.transforms_ref import keypoints_affine_ref
    S = 96
    pts = np.array([[60.0, 40.0], [30.0, 55.0], [48.0, 70.0], [70.0, 62.0]])
    for angle, shx, sc, tx, ty in ((90.0, 0.0, 1.0, 0, 0), (-37.0, 12.0, 0.8, 4, -6), (141.0, -25.0, 1.25, -3, 5), (20.0, 30.0, 0.6, 0, 0)):
        img = torch.zeros(1, S, S)
        for (x, y) in pts.astype(int):
            img[0, y - 1:y + 2, x - 1:x + 2] = 1.0
        out = affine_nearest_ref(img, angle, [tx, ty], sc, [shx, 0.0])
        k2 = keypoints_affine_ref(pts, angle, shx, 0.0, tx, ty, sc, S, S)
        for (x, y) in k2:
            xi, yi = int(round(x)), int(round(y))
            if 3 <= xi < S - 3 and 3 <= yi < S - 3:
                assert out[0, yi - 2:yi + 3, xi - 2:xi + 3].max() == 1.0, (angle, x, y)


def test_loop_rewarp_inverts_a_pil_forward_warp():
    """train_human.py:361-372: the three sequential nearest warps with aug_param (the INVERSE augmentation,
    keypoint_detection.py:139) bring the label blobs of a PIL-warped view back to where the un-augmented labels are."""
    from oracle.transforms_ref import affine_view_ref
    S, ratio = 128, 4.0
    pts = np.array([[70.0, 52.0], [44.0, 70.0], [64.0, 88.0], [84.0, 72.0]])
    for angle, shx, sc, tx, ty in ((35.0, 10.0, 0.9, 4, -8), (-58.0, -20.0, 1.2, -4, 4), (12.0, 28.0, 0.7, 8, 0)):
        img = np.zeros((S, S, 3), np.uint8)
        _, k2, aug = affine_view_ref(img, pts, angle, shx, 0.0, tx, ty, sc)
        hm = torch.zeros(len(pts), S // 4, S // 4)                       # label maps of the AUGMENTED view (a 3x3 core per key point)
        for i, (x, y) in enumerate(k2):
            cy, cx = int(y / ratio + 0.5), int(x / ratio + 0.5)
            hm[i, cy - 1:cy + 2, cx - 1:cx + 2] = 1.0
        back = warp3_ref(hm, aug[0], aug[1][0], aug[1][1], aug[2][0], aug[2][1], aug[3], ratio=ratio)
        for i, (x, y) in enumerate(pts):
            cy, cx = int(y / ratio + 0.5), int(x / ratio + 0.5)
            assert back[i, cy - 2:cy + 3, cx - 2:cx + 3].max() == 1.0, (angle, i)


def test_scale2_about_centre():
    x = _img()
    y = affine_nearest_ref(x, 0.0, [0, 0], 2.0, [0.0, 0.0])
    # output pixel (i,j) samples input at centre + (p - centre)/2 -> nearest
    assert y[0, 0, 0] == x[0, 2, 2] and y[0, 7, 7] == x[0, 5, 5]


def test_fp32_restatement_equals_a_float64_walk_of_the_chain_except_at_provable_rounding_ties():
    """The tie analysis the GPU tests rely on (tests/helpers/warp_ties.py), checked on the CPU: the fp32 restatement of the three chained nearest
    warps against a float64 evaluation of the same index chain from the same float32 matrices - every output pixel either comes from the pixel the
    float64 walk names, or its walk passes within 1e-4 of a half-integer source coordinate at some stage (a rounding tie: two correct fp32
    evaluations may differ there, and nowhere else)."""
    from helpers.warp_ties import tie_exposed
    from uda_poseestimation_amd import synthetic
    for (B, H, W, seed) in ((6, 64, 64, 5), (3, 24, 40, 17), (2, 96, 72, 33), (5, 7, 9, 1)):
        ap = synthetic.aug_params(B, np.random.RandomState(seed))
        angle, (tx, ty), (sx, sy), sc = ap
        x = torch.arange(B * H * W, dtype=torch.float32).reshape(B, 1, H, W) + 1.0          # every pixel its own value: the output names its source
        y = torch.stack([warp3_ref(x[i], float(angle[i]), float(tx[i]), float(ty[i]), float(sx[i]), float(sy[i]), float(sc[i]), 4.0) for i in range(B)])
        exposed, src = tie_exposed(ap, B, H, W, 4.0, return_source=True)
        want = torch.where(src >= 0, src.float() + 1.0 + (torch.arange(B).view(B, 1, 1) * H * W).float(), torch.zeros(()))
        differ = (y[:, 0] != want)
        assert not bool((differ & ~exposed).any()), (B, H, W, int((differ & ~exposed).sum()))
        assert exposed.float().mean().item() < 0.05 or (H & (H - 1)) or (W & (W - 1))
