"""Pin the CPU oracle to vectors captured from the reference's own Python (tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from oracle import keypoints_ref, losses_ref, mean_teacher_ref, style_ref
from oracle.pose_resnet_ref import UpsamplingRef, pose_resnet50_ref, pose_resnet101_ref


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_losses_match_reference(golden_dir):
    z = _g(golden_dir, "losses.npz")
    pred, gt, w = (torch.from_numpy(z[k]) for k in ("pred", "gt", "w"))
    mask = torch.from_numpy(z["mask"])
    np.testing.assert_allclose(losses_ref.joints_mse_ref(pred, gt, w).numpy(), z["mse_mean"], rtol=1e-6)
    np.testing.assert_allclose(losses_ref.joints_mse_ref(pred, gt).numpy(), z["mse_mean_now"], rtol=1e-6)
    np.testing.assert_allclose(losses_ref.joints_mse_ref(pred, gt, w, "none").numpy(), z["mse_none"], rtol=1e-6)
    np.testing.assert_allclose(losses_ref.cons_loss_ref(pred, gt, tea_mask=mask).numpy(), z["cons_masked"], rtol=1e-6)
    np.testing.assert_allclose(losses_ref.cons_loss_ref(pred, gt).numpy(), z["cons_plain"], rtol=1e-6)
    valid = torch.from_numpy(z["valid"])          # ConsLoss(valid_mask=) (loss.py:129-130)
    np.testing.assert_allclose(losses_ref.cons_loss_ref(pred, gt, valid_mask=valid, tea_mask=mask).numpy(), z["cons_valid"], rtol=1e-6)
    np.testing.assert_allclose(losses_ref.cons_loss_ref(pred, gt, valid_mask=valid).numpy(), z["cons_valid_only"], rtol=1e-6)
    # closed form of SURVEY Appendix F
    closed = (mask[:, :, None, None] * (pred - gt) ** 2).sum() / pred.numel()
    np.testing.assert_allclose(closed.numpy(), z["cons_masked"], rtol=1e-5)
    assert losses_ref.joints_mse_ref(pred, gt, w, "sum") is None


def test_decode_pck_rectify_match_reference(golden_dir):
    z = _g(golden_dir, "decode.npz")
    noisy = z["noisy"]
    p, v = keypoints_ref.get_max_preds_ref(noisy)
    np.testing.assert_array_equal(p, z["preds_np"])
    np.testing.assert_array_equal(v, z["maxv_np"])
    assert tuple(p[0, 0]) == (0.0, 0.0)            # max <= 0 -> (0,0)
    assert tuple(p[0, 1]) == (2.0, 1.0)            # tie -> first flat index
    acc, avg, cnt, pk = keypoints_ref.accuracy_ref(noisy, z["labels"])
    np.testing.assert_allclose(acc, z["acc"], rtol=0, atol=1e-12)
    assert abs(avg - float(z["avg_acc"])) < 1e-12 and cnt == int(z["cnt"])
    np.testing.assert_array_equal(pk, z["pred_kp"])
    pt, vt = keypoints_ref.get_max_preds_torch_ref(torch.from_numpy(noisy))
    np.testing.assert_array_equal(pt.numpy(), z["preds_t"])
    np.testing.assert_array_equal(vt.numpy(), z["maxv_t"])
    np.testing.assert_array_equal(mean_teacher_ref.rectify_ref(torch.from_numpy(noisy), 2).numpy(), z["rect_s2"])
    np.testing.assert_array_equal(mean_teacher_ref.rectify_ref(torch.from_numpy(noisy), 1.0).numpy(), z["rect_s1"])
    # known-answer: peak at (x=63,y=0): 49 non-zero cells, sum 9.0244
    r = z["rect_s2"][1, 2]
    assert (r != 0).sum() == 49 and abs(r.sum() - 9.0244) < 1e-3 and r[0, 63] == 1.0


def test_generate_target_matches_reference(golden_dir):
    z = _g(golden_dir, "decode.npz")
    kp = z["kp"]
    for b in range(kp.shape[0]):
        t, w = mean_teacher_ref.generate_target_ref(kp[b], np.ones((kp.shape[1], 1), np.float32), (64, 64), 2, (256, 256))
        np.testing.assert_array_equal(t, z["labels"][b])
        np.testing.assert_array_equal(w, z["weights"][b])
    assert (z["weights"] == 0).any() and (z["weights"] == 1).any()


def test_ema_matches_reference(golden_dir):
    z = _g(golden_dir, "ema.npz")
    tea = [torch.zeros_like(torch.from_numpy(z[f"src{i}"])) for i in range(3)]
    stu = [torch.from_numpy(z[f"src{i}"]).clone() for i in range(3)]
    mean_teacher_ref.ema_init_ref(tea, stu)
    for i in range(3):
        np.testing.assert_array_equal(tea[i].numpy(), z[f"init{i}"])
    for it in range(3):
        stu = [torch.from_numpy(z[f"stu_it{it}_{i}"]) for i in range(3)]
        mean_teacher_ref.ema_step_ref(tea, stu, 0.999)
    for i in range(3):
        np.testing.assert_array_equal(tea[i].numpy(), z[f"final{i}"])   # bit-exact


def test_style_matches_reference(golden_dir):
    from seeded import fill_style_weights
    z = _g(golden_dir, "style.npz")
    c, s = torch.from_numpy(z["c"]), torch.from_numpy(z["s"])
    m, sd = style_ref.calc_mean_std_ref(c)
    np.testing.assert_allclose(m.numpy(), z["mean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(sd.numpy(), z["std"], rtol=1e-6)
    np.testing.assert_allclose(style_ref.adain_ref(c, s).numpy(), z["adain"], rtol=1e-5, atol=1e-6)
    vgg = style_ref.make_vgg_ref()
    dec = style_ref.make_decoder_ref()
    fill_style_weights(vgg, 11)
    fill_style_weights(dec, 12)
    vgg31 = torch.nn.Sequential(*list(vgg.children())[:31])
    with torch.no_grad():
        feat = vgg31(torch.from_numpy(z["content"]))
        g_t = style_ref.style_forward_ref(vgg31, dec, torch.from_numpy(z["content"]), torch.from_numpy(z["style"]),
                                          float(z["alpha"]))
    np.testing.assert_allclose(feat.numpy(), z["feat"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g_t.numpy(), z["g_t"], rtol=1e-4, atol=1e-4)
    # the full forward (Style_net.py:163-177): content loss on relu4_1 + four Gram-matrix style losses
    np.testing.assert_allclose(style_ref.gram_matrix_ref(feat).numpy(), z["gram"], rtol=1e-5, atol=1e-7)
    with torch.no_grad():
        lc, ls, g2 = style_ref.style_forward_full_ref(vgg31, dec, torch.from_numpy(z["content"]), torch.from_numpy(z["style"]),
                                                      float(z["alpha"]))
    np.testing.assert_allclose(g2.numpy(), z["g_t"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(float(lc), float(z["loss_c"]), rtol=1e-4)
    np.testing.assert_allclose(float(ls), float(z["loss_s"]), rtol=1e-4)


def test_upsampling_head_match_reference(golden_dir):
    z = _g(golden_dir, "upsampling.npz")
    up = UpsamplingRef(64, hidden=(32, 32, 32))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("up_")}
    assert list(up.state_dict().keys()) == list(sd.keys())
    up.load_state_dict(sd)
    head = torch.nn.Conv2d(32, 5, 1)
    with torch.no_grad():
        head.weight.copy_(torch.from_numpy(z["head_w"]))
        head.bias.copy_(torch.from_numpy(z["head_b"]))
    up.train()
    y = head(up(torch.from_numpy(z["x"])))
    np.testing.assert_allclose(y.detach().numpy(), z["y"], rtol=1e-5, atol=1e-7)


def test_draw_labelmap_ori_matches_reference_bit_exact(golden_dir):
    """A14 (animal pipelines): oracle.mean_teacher_ref.animal_labels_ref / draw_labelmap_ori_ref against the reference's own
    draw_labelmap_ori run through its datasets' label loop (tests/golden/make_golden.py::labelmap): Gaussian and Cauchy, sigma 1.0 / 2 /
    1.5, 64x64 and 96x96 maps, centres on both sides of the whole-stamp-inside rule - maps and weights bit for bit."""
    from oracle.mean_teacher_ref import animal_labels_ref
    z = _g(golden_dir, "labelmap.npz")
    drawn = 0
    for ci, (sg, typ, res) in enumerate(zip(z["sigmas"], z["types"], z["sizes"])):
        sg = int(sg) if float(sg).is_integer() and sg >= 2 else float(sg)
        pts = z[f"pts{ci}"]
        t, w = animal_labels_ref(pts, pts[:, 2], z[f"gate{ci}"], int(res), sg, str(typ))
        assert np.array_equal(t, z[f"target{ci}"]) and np.array_equal(w, z[f"weight{ci}"]), (ci, sg, typ)
        drawn += int((t.reshape(len(t), -1).max(1) > 0).sum())
        # the rule itself: a row is drawn iff its gate is open and the whole (6 sigma + 1)^2 stamp is inside
        assert ((w[:, 0] > 0) <= (pts[:, 2] > 0)).all()
    assert drawn > 60


def test_state_dict_contract():
    """646 entries / 55,040,568 params / 325 tensors (R101, K=16); R50 36,048,440 / 172."""
    m = pose_resnet101_ref(16)
    sd = m.state_dict()
    assert len(sd) == 646
    assert sum(p.numel() for p in m.parameters()) == 55040568
    assert len(list(m.parameters())) == 325
    assert tuple(sd["upsampling.0.weight"].shape) == (2048, 256, 4, 4)
    assert tuple(sd["head.weight"].shape) == (16, 256, 1, 1)
    assert tuple(sd["backbone.fc.weight"].shape) == (1000, 2048)
    m50 = pose_resnet50_ref(16)
    assert sum(p.numel() for p in m50.parameters()) == 36048440 and len(list(m50.parameters())) == 172
    for K, n in ((18, 55041082), (21, 55041853), (14, 55040054)):
        assert sum(p.numel() for p in pose_resnet101_ref(K).parameters()) == n


def test_storage_emulation_with_per_stage_switches_reduces_to_its_two_ends():
    """oracle/bf16_emulation.forward_emulated (round 6: any 16-bit type, rounding switched per stage and tensor kind - the attribution tool's
    instrument): everything switched on IS forward_bf16_emulated, everything off IS the plain fp32 forward, and one stage's rounding alone moves
    the heat-maps by less than all of them together."""
    from oracle.bf16_emulation import STAGES, forward_bf16_emulated, forward_emulated
    from oracle.pose_resnet_ref import tiny_pose_resnet_ref
    torch.manual_seed(0)
    m = tiny_pose_resnet_ref(4).train()
    x = torch.randn(2, 3, 64, 64)
    keep = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        full = forward_emulated(m, x, torch.bfloat16, lambda s, k: True)
        assert torch.equal(full, forward_bf16_emulated(m, x))
        none = forward_emulated(m, x, torch.bfloat16, lambda s, k: False)
        m.load_state_dict(keep)
        assert (none - m(x)).abs().max().item() < 1e-6
        e_all = (full - none).abs().max().item()
        e_one = max((forward_emulated(m, x, torch.float16, lambda s, k, st=st: s == st) - none).abs().max().item() for st in STAGES)
        assert 0 < e_one < e_all
