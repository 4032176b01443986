"""A train_human.py-shaped mini loop on the MI355X written with the REFERENCE'S CALL FORMS ONLY (train_human.py:19-29,
117-148, 262-302, 326-444): drop-in imports, `models.__dict__[arch](num_keypoints=K).cuda()`, `torch.optim.Adam`,
`OldWeightEMA`, `MultiStepLR`, single-device `torch.nn.DataParallel(...).cuda()`, `torch.cuda.amp.autocast()` +
`GradScaler` (scale -> backward -> step -> update), per-sample `tF.affine` triplets, `rectify`, `torch.kthvalue` mask,
`accuracy(y.detach().cpu().numpy(), ...)`, the recover clamp, `Style_net.Net(vgg, decoder)` under DataParallel.
One `pretrain` and one `train` iteration; losses / PCK / parameter updates are checked against the CPU oracle
(oracle/step_ref.py) stepped from the same weights on the same batch.

torchvision is not installed in this image: `tF` is `uda_poseestimation_amd.warp` (same `affine` signature and semantics);
every other name below is what the reference's script itself imports."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Args:
    arch = "pose_resnet50"
    image_size, heatmap_size, sigma, k = 128, 32, 2, 1
    lr, teacher_alpha, lambda_c, mask_ratio = 1e-4, 0.999, 1.0, 0.5
    lr_step, lr_factor = [45, 60], 0.1
    s2t_freq, s2t_alpha = 1.0, (0.5, 0.5)
    occlude_rate = -1


def test_reference_call_forms_pretrain_and_train_iteration():
    # ---- the two INTEGRATION.md lines + the reference's import block (train_human.py:11-29)
    pkg = os.path.join(ROOT, "uda_poseestimation_amd")
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    from torch.optim import Adam
    from torch.optim.lr_scheduler import MultiStepLR
    import lib.models as models
    from lib.models.loss import JointsMSELoss, ConsLoss
    from lib.keypoint_detection import accuracy
    from lib.models import Style_net
    import utils as ref_named_utils
    OldWeightEMA, rectify = ref_named_utils.OldWeightEMA, ref_named_utils.rectify      # (`from utils import *`)
    from uda_poseestimation_amd import warp as tF                                      # torchvision.transforms.functional stand-in

    from oracle.keypoints_ref import accuracy_ref
    from oracle.pose_resnet_ref import pose_resnet50_ref
    from oracle.step_ref import pretrain_step_ref, train_step_ref
    from seeded import fill_style_weights
    from uda_poseestimation_amd import synthetic

    args = _Args()
    device = torch.device("cuda")
    recover_min = torch.tensor([-2.1179, -2.0357, -1.8044]).to(device)
    recover_max = torch.tensor([2.2489, 2.4285, 2.64]).to(device)
    K, N, S = 16, 4, args.image_size

    # ---- main(): model / optimizer / EMA / scheduler / DataParallel, in the reference's order (train_human.py:117-148)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # pretrained ImageNet weights cannot be downloaded here: random init kept
        student = models.__dict__[args.arch](num_keypoints=K).cuda()
        teacher = models.__dict__[args.arch](num_keypoints=K).cuda()
    ref_s, ref_t = pose_resnet50_ref(K), pose_resnet50_ref(K)
    with torch.no_grad():      # trained-like conditioning (small residual branches) keeps the bf16 storage noise small
        for m in ref_s.modules():
            if hasattr(m, "bn3"):
                m.bn3.weight.fill_(0.25)
    student.load_state_dict(ref_s.state_dict())
    fill_style_weights(Style_net.vgg, 11)
    fill_style_weights(Style_net.decoder, 12)
    decoder = Style_net.decoder
    vgg = Style_net.vgg
    vgg = torch.nn.Sequential(*list(vgg.children())[:31])
    style_net = Style_net.Net(vgg, decoder)
    style_net.requires_grad = False
    criterion = JointsMSELoss()
    con_criterion = ConsLoss()
    stu_optimizer = Adam(student.parameters(), lr=args.lr)
    tea_optimizer = OldWeightEMA(teacher, student, alpha=args.teacher_alpha)
    lr_scheduler = MultiStepLR(stu_optimizer, args.lr_step, args.lr_factor)
    student = torch.nn.DataParallel(student).cuda()
    teacher = torch.nn.DataParallel(teacher).cuda()
    style_net = torch.nn.DataParallel(style_net).cuda()
    for a, b in zip(teacher.module.parameters(), student.module.parameters()):
        assert torch.equal(a, b)                                       # EMA ctor copied student -> teacher
    ref_t.load_state_dict(ref_s.state_dict())
    ref_opt = torch.optim.Adam(ref_s.parameters(), lr=args.lr)

    b = synthetic.mean_teacher_batch(N, num_keypoints=K, image_size=S, heatmap_size=args.heatmap_size, seed=21)

    # =================================================================== pretrain(), one iteration (train_human.py:262-289)
    student.train()
    scaler = torch.cuda.amp.GradScaler()
    stu_optimizer.zero_grad()
    x_s, label_s, weight_s = b["x_s"].to(device), b["label_s"].to(device), b["weight_s"].to(device)
    if style_net is not None and args.s2t_freq > np.random.rand():
        with torch.no_grad():
            x_t = b["x_t_tea"].to(device)
            _a = np.random.uniform(*args.s2t_alpha)
            x_s = style_net(x_s, x_t, _a)[2]
            x_s = torch.maximum(torch.minimum(x_s.permute(0, 2, 3, 1), recover_max), recover_min).permute(0, 3, 1, 2)
    assert x_s.shape == (N, 3, S, S) and torch.isfinite(x_s).all()
    assert (x_s.amax(dim=(0, 2, 3)) <= recover_max + 1e-6).all() and (x_s.amin(dim=(0, 2, 3)) >= recover_min - 1e-6).all()
    with torch.cuda.amp.autocast():
        y_s = student(x_s)
        loss_s = criterion(y_s, label_s, weight_s)
    loss_all = loss_s
    # the reference's precisions, selected the way the reference selects them: autocast() -> fp16 student (train_human.py:280),
    # the style network outside autocast -> fp32-grade
    assert student.module._last_hd.precision == "fp16" and style_net.module.precision == "f16x2"
    scaler.scale(loss_all).backward()
    w0 = [p.detach().clone() for p in student.module.parameters()]
    scaler.step(stu_optimizer)
    scaler.update()
    _, avg_acc_s, cnt_s, pred_s = accuracy(y_s.detach().cpu().numpy(), label_s.detach().cpu().numpy())

    x_s_cpu = x_s.detach().cpu().contiguous()            # the oracle gets the SAME (stylised) source images
    ref = pretrain_step_ref(ref_s, ref_opt, x_s_cpu, b["label_s"], b["weight_s"])
    assert y_s.shape == (N, K, args.heatmap_size, args.heatmap_size) and y_s.dtype == torch.float32
    assert abs(float(loss_s) - float(ref["loss_all"])) <= 2e-2 * float(ref["loss_all"]), (float(loss_s), float(ref["loss_all"]))
    # PCK of the device heat-maps: the drop-in accuracy() equals the reference's numpy algorithm on the same arrays
    acc_ref = accuracy_ref(y_s.detach().cpu().numpy(), b["label_s"].numpy())
    assert avg_acc_s == pytest.approx(acc_ref[1], abs=1e-6) and cnt_s == acc_ref[2]
    assert np.array_equal(pred_s, acc_ref[3])
    # GradScaler unscaled the gradients and torch.optim.Adam stepped them: first Adam step = -lr * sign(grad)
    assert scaler.get_scale() == 65536.0                              # no inf/nan found: the step was NOT skipped
    agree = total = moved = 0
    for p, p_ref, p0, (name, _) in zip(student.module.parameters(), ref_s.parameters(), w0, ref_s.named_parameters()):
        if name.startswith("backbone.fc."):
            assert p.grad is None and torch.equal(p.detach(), p0)      # no gradient, untouched (as under autograd)
            continue
        d_dev, d_ref = (p.detach() - p0).cpu(), p_ref.detach() - p0.cpu()
        moved += int((d_dev != 0).sum())
        sel = d_ref.abs() > 5e-5
        agree += int((torch.sign(d_dev[sel]) == torch.sign(d_ref[sel])).sum())
        total += int(sel.sum())
    assert moved > 0.9 * total and agree / max(total, 1) > 0.8, (moved, agree, total)

    # =================================================================== train(), one iteration (train_human.py:326-444)
    # (oracle re-synchronised with the device weights so that this iteration is compared from identical state)
    ref_s.load_state_dict({k: v.cpu() for k, v in student.module.state_dict().items()})
    ref_t.load_state_dict({k: v.cpu() for k, v in teacher.module.state_dict().items()})
    ref_opt = torch.optim.Adam(ref_s.parameters(), lr=args.lr)
    lr_scheduler.step()
    student.train()
    teacher.train()
    scaler = torch.cuda.amp.GradScaler()
    stu_optimizer.zero_grad()
    x_s = b["x_s"].to(device)
    x_t_stu = b["x_t_stu"].to(device)
    x_t_teas = [b["x_t_tea"].to(device)]
    meta_t_stu = {"aug_param_stu": b["aug_param_stu"]}
    meta_t_tea = [{"aug_param_tea": b["aug_param_tea"]}]
    ratio = args.image_size / args.heatmap_size
    with torch.no_grad():
        y_t_teas = [teacher(x_t_tea) for x_t_tea in x_t_teas]
        y_t_tea_recon = torch.zeros_like(y_t_teas[0]).cuda()
        tea_mask = torch.zeros(y_t_teas[0].shape[:2]).cuda()
        for ind in range(x_t_teas[0].size(0)):
            recons = torch.zeros(args.k, *y_t_teas[0].size()[1:])
            for _k in range(args.k):
                angle, [trans_x, trans_y], [shear_x, shear_y], scale = meta_t_tea[_k]['aug_param_tea']
                _angle, _trans_x, _trans_y, _shear_x, _shear_y, _scale = angle[ind].item(), trans_x[ind].item(), trans_y[ind].item(), \
                    shear_x[ind].item(), shear_y[ind].item(), scale[ind].item()
                temp = tF.affine(y_t_teas[_k][ind], 0., translate=[_trans_x / ratio, _trans_y / ratio], shear=[0., 0.], scale=1.)
                temp = tF.affine(temp, _angle, translate=[0., 0.], shear=[0., 0.], scale=_scale)
                temp = tF.affine(temp, 0., translate=[0, 0], shear=[_shear_x, _shear_y], scale=1.)
                recons[_k] = temp
            y_t_tea_recon[ind] = torch.mean(recons, dim=0)
            tea_mask[ind] = 1.
        angle, [trans_x, trans_y], [shear_x, shear_y], scale = meta_t_stu['aug_param_stu']
    # the teacher ran under no_grad OUTSIDE autocast: fp32 in the reference (train_human.py:358), the fp32-grade mode here - its
    # heat-maps meet north_star's 1e-3 bar against the fp32 oracle teacher on the same weights
    assert teacher.module._last_hd.precision == "f16x2"
    ref_t.train()
    bufs_t = {k: v.clone() for k, v in ref_t.state_dict().items() if "running" in k or "num_batches" in k}
    with torch.no_grad():
        y_t_ref = ref_t(b["x_t_tea"])
    ref_t.load_state_dict(bufs_t, strict=False)          # (train_step_ref below runs the oracle teacher's forward itself)
    e_tea = (y_t_teas[0].cpu() - y_t_ref).abs().max().item()
    print(f"drop-in teacher (auto -> f16x2) vs fp32 oracle: max|dheatmap| {e_tea:.2e} (max|y| {y_t_ref.abs().max().item():.3f})")
    assert e_tea < 1e-3
    with torch.cuda.amp.autocast():
        y_s = student(x_s)
        y_t_stu = student(x_t_stu)
        y_t_stu_recon = torch.zeros_like(y_t_stu).cuda()
        for ind in range(x_t_stu.size(0)):
            _angle, _trans_x, _trans_y, _shear_x, _shear_y, _scale = angle[ind].item(), trans_x[ind].item(), trans_y[ind].item(), \
                shear_x[ind].item(), shear_y[ind].item(), scale[ind].item()
            temp = tF.affine(y_t_stu[ind], 0., translate=[_trans_x / ratio, _trans_y / ratio], shear=[0., 0.], scale=1.)
            temp = tF.affine(temp, _angle, translate=[0., 0.], shear=[0., 0.], scale=_scale)
            y_t_stu_recon[ind] = tF.affine(temp, 0., translate=[0., 0.], shear=[_shear_x, _shear_y], scale=1.)
        loss_s = criterion(y_s, label_s, weight_s)
        activates = y_t_tea_recon.amax(dim=(2, 3))
        y_t_tea_recon = rectify(y_t_tea_recon, sigma=args.sigma)
        mask_thresh = torch.kthvalue(activates.view(-1), int(args.mask_ratio * activates.numel()))[0].item()
        tea_mask = tea_mask * activates > mask_thresh
        loss_c = con_criterion(y_t_stu_recon, y_t_tea_recon, tea_mask=tea_mask)
    loss_all = loss_s + args.lambda_c * loss_c
    assert student.module._last_hd.precision == "fp16"
    t0 = [p.detach().clone() for p in teacher.module.parameters()]
    scaler.scale(loss_all).backward()
    scaler.step(stu_optimizer)
    tea_optimizer.step()
    scaler.update()
    _, avg_acc_s, cnt_s, pred_s = accuracy(y_s.detach().cpu().numpy(), label_s.detach().cpu().numpy())

    ref = train_step_ref(ref_s, ref_t, ref_opt, b["x_s"], b["label_s"], b["weight_s"], b["x_t_stu"], b["x_t_tea"], b["aug_param_stu"],
                         b["aug_param_tea"], lambda_c=args.lambda_c, mask_ratio=args.mask_ratio, sigma=args.sigma, ratio=ratio,
                         alpha=args.teacher_alpha)
    assert abs(float(loss_s) - float(ref["loss_s"])) <= 2e-2 * float(ref["loss_s"]), (float(loss_s), float(ref["loss_s"]))
    assert abs(float(loss_c) - float(ref["loss_c"])) <= 8e-2 * float(ref["loss_c"]) + 1e-6, (float(loss_c), float(ref["loss_c"]))
    assert tea_mask.dtype == torch.bool and int(tea_mask.sum()) == int(ref["tea_mask"].sum())       # k-th value mask: same count
    assert scaler.get_scale() == 65536.0
    # EMA after the optimizer step (train_human.py:437-438), bit-exact two-rounding form on the device student
    for p_t, p_s, p0 in zip(teacher.module.parameters(), student.module.parameters(), t0):
        assert torch.equal(p_t.detach(), p0.mul(args.teacher_alpha).add(p_s.detach() * (1.0 - args.teacher_alpha)))
    # checkpoints carry the DataParallel `module.` prefix (train_human.py:229-230) and load back into a bare model
    sd = student.state_dict()
    assert all(k.startswith("module.") for k in sd) and len(sd) == len(ref_s.state_dict())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fresh = models.__dict__[args.arch](num_keypoints=K).cuda()
    fresh.load_state_dict(sd)
    # ... and the reloaded copy gives the student's heat-maps (eval mode: running statistics, fresh weight packs)
    student.eval(); fresh.eval()
    with torch.no_grad():
        assert torch.equal(fresh(x_s), student(x_s))
