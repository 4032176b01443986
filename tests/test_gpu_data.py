"""GPU data pipeline of the target views (uda_poseestimation_amd/data_gpu.py, csrc/augment.hip) against the CPU oracle
(oracle/transforms_ref.py = PIL's own arithmetic + the reference's key-point / label code): warped and colour-jittered
uint8 images bit-exact, normalised tensors bit-exact, Gaussian label maps bit-exact, aug_param = the inverse augmentation,
and the collated 8-tuple drives a mean-teacher step (the aug_param contract of SURVEY.md Appendix D)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu



@pytest.fixture(autouse=True)
def _bf16_unless_stated(monkeypatch):
    """The tests of this module exercise the 16-bit executor: networks start in 'bf16' (BASELINE.json's benched precision) unless a
    test sets another precision.  (A new module's default is 'auto': autocast dtype / fp32-grade teacher, tests/test_gpu_dropin_loop.py.)"""
    from uda_poseestimation_amd.lib.models.pose_resnet import PoseResNet
    monkeypatch.setattr(PoseResNet, "default_precision", "bf16")

def test_views_bit_exact_with_pil_oracle():
    from oracle import transforms_ref as R
    from oracle.mean_teacher_ref import generate_target_ref
    from uda_poseestimation_amd import data_gpu as D
    rs = np.random.RandomState(0)
    N, S, K = 6, 128, 16
    base = rs.randint(0, 256, (N, S, S, 3)).astype(np.uint8)
    base[1, 40:60, 30:90] = 250; base[2] = (base[2] // 3)            # some structure / a dark image (contrast mean matters)
    kps = rs.uniform(-10, S + 10, (N, K, 2))                         # some key points outside the image: weight 0
    pipe = D.TargetViewPipeline(image_size=S, heatmap_size=S // 4, sigma=2, rng=random.Random(3))
    cfg = D.ViewConfig(rotation=60, color=0.25)
    params = [cfg.draw_affine(random.Random(10 + i), (S, S)) for i in range(N)]
    jit = [cfg.draw_jitter(random.Random(20 + i)) for i in range(N)]
    jit[0] = ([2, 1, 3], [1.25, 0.75, 1.1]); jit[1] = ([3, 2, 1], [0.8, 1.2, 1.25])      # every order position of the contrast step
    # the separate steps on their own first: warp only / jitter only / tensor conversion only
    w_only = pipe.warp_images(torch.from_numpy(base).cuda(), params).cpu().numpy()
    for i in range(N):
        ref_i = R.affine_view_ref(base[i], kps[i], *params[i])[0]
        assert np.array_equal(w_only[i], ref_i), f"warp {i}: {(w_only[i] != ref_i).sum()} bytes differ from PIL"
    j_only = pipe.jitter_(torch.from_numpy(base).cuda().clone(), [j[0] for j in jit], [j[1] for j in jit]).cpu().numpy()
    for i in range(N):
        ref_i = R.color_jitter_ref(base[i], *jit[i])
        assert np.array_equal(j_only[i], ref_i), f"jitter {i} {jit[i]}: {(j_only[i] != ref_i).sum()} bytes differ from PIL"
    t_only = pipe.to_tensor(torch.from_numpy(base).cuda()).cpu()
    for i in range(N):
        ref_i = R.to_tensor_normalize_ref(base[i], D.IMAGENET_MEAN, D.IMAGENET_STD)
        assert torch.equal(t_only[i], ref_i), f"to_tensor {i}: max diff {(t_only[i] - ref_i).abs().max().item():.3e}"
    x, kp_t, aug, target, weight = pipe.view(torch.from_numpy(base).cuda(), kps, cfg, params=params, jitter=jit)
    assert x.shape == (N, 3, S, S) and target.shape == (N, K, S // 4, S // 4) and weight.shape == (N, K, 1)
    for i in range(N):
        w_ref, k_ref, aug_ref = R.affine_view_ref(base[i], kps[i], *params[i])
        j_ref = R.color_jitter_ref(w_ref, *jit[i])
        t_ref = R.to_tensor_normalize_ref(j_ref, D.IMAGENET_MEAN, D.IMAGENET_STD)
        assert torch.equal(x[i].cpu(), t_ref), f"view {i}: image differs from PIL"
        np.testing.assert_allclose(kp_t[i], k_ref, rtol=0, atol=1e-12)
        assert float(aug[0][i]) == aug_ref[0] and int(aug[1][0][i]) == aug_ref[1][0] and int(aug[1][1][i]) == aug_ref[1][1]
        assert float(aug[2][0][i]) == aug_ref[2][0] and float(aug[3][i]) == aug_ref[3]
        lt, lw = generate_target_ref(k_ref, np.ones((K, 1), np.float32), (S // 4, S // 4), 2, (S, S))
        assert np.array_equal(target[i].cpu().numpy(), lt) and np.array_equal(weight[i].cpu().numpy(), lw)
    assert (weight == 0).any() and (weight == 1).any()


def test_pipeline_batch_feeds_a_training_step():
    """The collated 8-tuple has the layout the loop reads (train_human.py:330-345, Appendix D) and its aug_param really is
    the inverse of the image warp: re-warping the labels of the augmented view with aug_param recovers the original labels
    around every key point that stayed inside the image."""
    from uda_poseestimation_amd import data_gpu as D, synthetic, warp
    from uda_poseestimation_amd.engine import MeanTeacherTrainer
    import uda_poseestimation_amd.lib.models.pose_resnet as pr
    rs = np.random.RandomState(1)
    N, S, K = 4, 128, 16
    base = torch.from_numpy(rs.randint(0, 256, (N, S, S, 3)).astype(np.uint8)).cuda()
    kps = rs.uniform(30, S - 30, (N, K, 2))
    pipe = D.TargetViewPipeline(image_size=S, heatmap_size=S // 4, sigma=2, k=1, student=D.ViewConfig(rotation=60), teacher=D.ViewConfig(rotation=60),
                                rng=random.Random(7))
    x_t_stu, t_stu, w_stu, meta_stu, x_t_teas, ts_tea, ws_tea, metas_tea = pipe(base, kps)
    assert x_t_stu.shape == (N, 3, S, S) and len(x_t_teas) == 1 and set(meta_stu) >= {"aug_param_stu", "target_ori", "target_weight_ori"}
    # inverse property on the label maps
    recon = warp.recon_heatmaps(t_stu, meta_stu["aug_param_stu"], ratio=4.0)
    ori = meta_stu["target_ori"]
    hits = total = 0
    offs = []
    for n in range(N):
        for k in range(K):
            if float(w_stu[n, k]) > 0.5:
                total += 1
                py, px = divmod(int(ori[n, k].argmax()), S // 4)
                qy, qx = divmod(int(recon[n, k].argmax()), S // 4)
                offs.append((qx - px, qy - py, round(float(recon[n, k].max()), 2)))
                # (label centres are rounded to cells twice and the blob goes through three nearest resamplings: 2 cells)
                hits += abs(py - qy) <= 2 and abs(px - qx) <= 2 and float(recon[n, k].max()) > 0.3
    print("re-warped label offsets (dx, dy, peak):", offs)
    assert total >= 30 and hits >= 0.85 * total, (hits, total)
    # ... and the batch drives the step
    torch.manual_seed(0)
    stu = pr._pose_resnet("t", K, pr.Bottleneck_default, [1, 1, 1, 1], False, False).cuda()
    tea = pr._pose_resnet("t", K, pr.Bottleneck_default, [1, 1, 1, 1], False, False).cuda()
    trainer = MeanTeacherTrainer(stu, tea, image_size=S, heatmap_size=S // 4)
    src = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=S // 4, seed=2)
    out = trainer.train_step(src["x_s"].cuda(), src["label_s"].cuda(), src["weight_s"].cuda(), x_t_stu, x_t_teas, meta_stu["aug_param_stu"],
                             [m["aug_param_tea"] for m in metas_tea])
    assert torch.isfinite(out["loss_all"]) and float(out["loss_c"]) > 0


def test_gaussian_blur_bit_exact_with_pil():
    """T.GaussianBlur (lib/transforms/keypoint_detection.py:216-225, `--blur_stu / --blur_tea`): PIL's three-pass box blur in 8.24
    fixed point on the device, byte for byte, for radii from PIL's copy (0) over the reference's typical U(0, 0.8) to boxes wider
    than the image; non-square images; and inside a whole view (after the colour jitter, before ToTensor)."""
    from oracle import transforms_ref as R
    from uda_poseestimation_amd import data_gpu as D
    rs = np.random.RandomState(5)
    pipe = D.TargetViewPipeline(image_size=64, heatmap_size=16, sigma=2, rng=random.Random(1))
    for (N, H, W) in ((7, 64, 64), (3, 17, 40)):
        base = rs.randint(0, 256, (N, H, W, 3)).astype(np.uint8)
        base[0, H // 3:H // 2, W // 4:W // 2] = 255
        radii = [0.0, 0.05, 0.37, 0.8, 1.9, 4.2, 30.0][:N]
        out = pipe.blur_(torch.from_numpy(base).cuda().clone(), radii).cpu().numpy()
        for i, r in enumerate(radii):
            ref = R.gaussian_blur_ref(base[i], r)
            assert np.array_equal(out[i], ref), f"blur {H}x{W} radius {r}: {(out[i] != ref).sum()} bytes differ from PIL (params {D.pil_box_blur_params(r)})"
    # a whole student view with blur: warp -> jitter -> blur -> ToTensor + Normalize
    N, S, K = 4, 64, 16
    base = rs.randint(0, 256, (N, S, S, 3)).astype(np.uint8)
    kps = rs.uniform(5, S - 5, (N, K, 2))
    cfg = D.ViewConfig(rotation=40, color=0.25, blur=0.8)
    params = [cfg.draw_affine(random.Random(30 + i), (S, S)) for i in range(N)]
    jit = [cfg.draw_jitter(random.Random(40 + i)) for i in range(N)]
    blur = [cfg.draw_blur(random.Random(50 + i)) for i in range(N)]
    assert all(0.0 <= b <= 0.8 for b in blur)
    x = pipe.view(torch.from_numpy(base).cuda(), kps, cfg, params=params, jitter=jit, blur=blur)[0]
    for i in range(N):
        w_ref = R.affine_view_ref(base[i], kps[i], *params[i])[0]
        t_ref = R.to_tensor_normalize_ref(R.gaussian_blur_ref(R.color_jitter_ref(w_ref, *jit[i]), blur[i]), D.IMAGENET_MEAN, D.IMAGENET_STD)
        assert torch.equal(x[i].cpu(), t_ref), f"blurred view {i} differs from PIL"


def test_resized_crop_bit_exact_with_pil():
    """T.RandomResizedCrop (train_human.py:55,64; lib/transforms/keypoint_detection.py:456-521): crop + PIL bilinear resize on the device,
    byte for byte with PIL, at the `_mt` datasets' sizes (512x512 crops -> 256x256; the reference's scale range gives reductions by up
    to 2x, where PIL widens the filter) and at small odd sizes incl. enlargement and the identity; key points shifted and scaled; and
    the whole pipeline from raw images (`raw=True`) against the oracle chain."""
    from oracle import transforms_ref as R
    from uda_poseestimation_amd import data_gpu as D
    rs = np.random.RandomState(8)
    # (1) the real sizes, boxes drawn like the reference draws them
    pipe = D.TargetViewPipeline(image_size=256, heatmap_size=64, sigma=2, rng=random.Random(2))
    N, Hs, K = 6, 512, 16
    raw = rs.randint(0, 256, (N, Hs, Hs, 3)).astype(np.uint8)
    raw[:, 100:140, 200:300] = 255                       # (edges, so that the filter's support shows)
    kps = rs.uniform(0, Hs, (N, K, 2))
    boxes = [D.draw_resized_crop(random.Random(60 + i), Hs, Hs, (0.6, 1.3)) for i in range(N)]
    boxes[0] = (0, 0, Hs, Hs)                            # the fall-back: whole image
    boxes[1] = (256, 256, 256, 256)                      # the identity (PIL returns a copy)
    assert len({b[2] for b in boxes}) >= 4
    out, kp = pipe.resized_crop(torch.from_numpy(raw).cuda(), kps, boxes=boxes)
    for i, (top, left, h, w) in enumerate(boxes):
        ref, kref = R.resized_crop_ref(raw[i], kps[i], top, left, h, w, 256)
        assert np.array_equal(out[i].cpu().numpy(), ref), f"box {boxes[i]}: {(out[i].cpu().numpy() != ref).sum()} bytes differ from PIL"
        assert np.array_equal(kp[i], kref)
    # (2) small sizes: reduction by 3.1x (ksize 9), enlargement, non-multiple sizes
    pipe2 = D.TargetViewPipeline(image_size=31, heatmap_size=8, sigma=1, rng=random.Random(3))
    raw2 = rs.randint(0, 256, (5, 97, 97, 3)).astype(np.uint8)
    kps2 = rs.uniform(0, 97, (5, 3, 2))
    boxes2 = [(0, 0, 97, 97), (1, 2, 95, 95), (40, 13, 17, 17), (66, 66, 31, 31), (10, 20, 50, 50)]
    out2, kp2 = pipe2.resized_crop(torch.from_numpy(raw2).cuda(), kps2, boxes=boxes2)
    for i, (top, left, h, w) in enumerate(boxes2):
        ref, kref = R.resized_crop_ref(raw2[i], kps2[i], top, left, h, w, 31)
        assert np.array_equal(out2[i].cpu().numpy(), ref), f"box {boxes2[i]}"
        assert np.array_equal(kp2[i], kref)
    with pytest.raises(ValueError):
        pipe2.resized_crop(torch.from_numpy(raw2).cuda(), kps2, boxes=[(90, 0, 17, 17)] * 5)
    # (3) the whole `_mt` sample from raw images: same RNG stream on both sides (crop draws first, then the views')
    pipe3 = D.TargetViewPipeline(image_size=64, heatmap_size=16, sigma=2, k=1, rng=random.Random(77))
    raw3 = rs.randint(0, 256, (3, 128, 128, 3)).astype(np.uint8)
    kps3 = rs.uniform(20, 108, (3, K, 2))
    x_s, t_s, w_s, meta, xs, ts, ws, metas = pipe3(torch.from_numpy(raw3).cuda(), kps3, raw=True)
    r = random.Random(77)
    boxes3 = [D.draw_resized_crop(r, 128, 128, (0.6, 1.3)) for _ in range(3)]
    base = [R.resized_crop_ref(raw3[i], kps3[i], *boxes3[i], 64) for i in range(3)]
    assert np.array_equal(meta["keypoint2d_ori"], np.stack([b[1] for b in base]))
    prm = [pipe3.stu.draw_affine(r, (64, 64)) for _ in range(3)]
    jit = [pipe3.stu.draw_jitter(r) for _ in range(3)]
    for i in range(3):
        w_ref = R.affine_view_ref(base[i][0], base[i][1], *prm[i])[0]
        t_ref = R.to_tensor_normalize_ref(R.color_jitter_ref(w_ref, *jit[i]), D.IMAGENET_MEAN, D.IMAGENET_STD)
        assert torch.equal(x_s[i].cpu(), t_ref), f"student view {i} from the raw image differs from the PIL chain"


def test_draw_labelmap_ori_device_bit_exact_with_reference_golden_and_oracle():
    """udapose_draw_labelmap_ori (data_gpu.TargetViewPipeline.labels_animal) = the animal datasets' label loop over draw_labelmap_ori
    (lib/datasets/util.py:326-363; BASELINE.json configs[4]'s labels): bit-exact against the reference's own outputs (labelmap.npz)
    for Gaussian / Cauchy, sigma 1.0 / 2 / 1.5, 64x64 / 96x96, and against the oracle on a configs[4]-shaped batch (N=8, K=18, 96x96)."""
    import os
    from oracle.mean_teacher_ref import animal_labels_ref
    from uda_poseestimation_amd import data_gpu as D
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "labelmap.npz"))
    pipe = D.TargetViewPipeline(image_size=384, heatmap_size=96, sigma=1.0)
    dev = torch.device("cuda")
    for ci, (sg, typ, res) in enumerate(zip(z["sigmas"], z["types"], z["sizes"])):
        sg = int(sg) if float(sg).is_integer() and sg >= 2 else float(sg)
        pts = z[f"pts{ci}"]
        t, w = pipe.labels_animal(pts[None], pts[None, :, 2], z[f"gate{ci}"][None], dev, sigma=sg, label_type=str(typ), out_res=int(res))
        assert np.array_equal(t[0].cpu().numpy(), z[f"target{ci}"]), (ci, sg, typ)
        assert np.array_equal(w[0].cpu().numpy(), z[f"weight{ci}"]), (ci, sg, typ)
    rs = np.random.RandomState(11)
    N, K, R = 8, 18, 96
    tp = rs.uniform(-4, R + 4, (N, K, 3)).astype(np.float32)
    vis = (rs.rand(N, K) > 0.1).astype(np.float32)
    gate = rs.rand(N, K) > 0.1
    t, w = pipe.labels_animal(tp, vis, gate, dev)
    for n in range(N):
        tr, wr = animal_labels_ref(tp[n], vis[n], gate[n], R, 1.0, "Gaussian")
        assert np.array_equal(t[n].cpu().numpy(), tr) and np.array_equal(w[n].cpu().numpy(), wr), n
