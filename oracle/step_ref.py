"""CPU restatement of one training step of the reference loop (oracle / CPU baseline, test-only).

  * pretrain_step_ref  <- train_human.py:262-289 (source-only: forward, JointsMSE, backward, Adam)
  * train_step_ref     <- train_human.py:326-440 (no style, no occlusion: BASELINE.json configs[1]; x_t_tea / aug_param_tea may be
                          LISTS of k teacher views, :358-372: the re-warped maps are averaged per sample)
  * train_step_full_ref <- the same iteration with the AdaIN style passes (:345-356) and the adaptive occlusion (:374-412) in the
    reference's order of host random draws: BASELINE.json configs[2]
Plain torch fp32 on the host; autocast/GradScaler of the reference are CUDA-only and are not restated (fp32 is the
stated oracle precision, BASELINE.json north_star).
"""
import torch

from .affine_ref import warp3_ref
from .losses_ref import cons_loss_ref, joints_mse_ref
from .mean_teacher_ref import conf_mask_ref, ema_step_ref, rectify_ref


def _recon(y, aug_param, ratio):
    angle, (tx, ty), (sx, sy), scale = aug_param
    out = torch.zeros_like(y)
    for i in range(y.shape[0]):
        out[i] = warp3_ref(y[i], float(angle[i]), float(tx[i]), float(ty[i]), float(sx[i]), float(sy[i]), float(scale[i]), ratio)
    return out


def pretrain_step_ref(student, optimizer, x_s, label_s, weight_s):
    student.train()
    optimizer.zero_grad()
    y_s = student(x_s)
    loss = joints_mse_ref(y_s, label_s, weight_s)
    loss.backward()
    optimizer.step()
    return {"loss_all": loss.detach(), "y_s": y_s.detach()}


def train_step_ref(student, teacher, optimizer, x_s, label_s, weight_s, x_t_stu, x_t_tea, aug_param_stu, aug_param_tea, lambda_c=1.0,
                   mask_ratio=0.5, sigma=2, ratio=4.0, alpha=0.999):
    """configs[1]: the step without style transfer and occlusion."""
    return train_step_full_ref(student, teacher, optimizer, x_s, label_s, weight_s, x_t_stu, x_t_tea, aug_param_stu, aug_param_tea,
                               lambda_c=lambda_c, mask_ratio=mask_ratio, sigma=sigma, ratio=ratio, alpha=alpha)


def train_step_full_ref(student, teacher, optimizer, x_s, label_s, weight_s, x_t_stu, x_t_tea, aug_param_stu, aug_param_tea, lambda_c=1.0,
                        mask_ratio=0.5, sigma=2, ratio=4.0, alpha=0.999, style=None, rng=None, s2t_freq=0.5, t2s_freq=0.5,
                        s2t_alpha=(0.0, 1.0), t2s_alpha=(0.0, 1.0), recover=None, occlude_rate=-1.0, occlude_thresh=0.9, occlude_size=10,
                        image_size=256):
    """One iteration of train() (train_human.py:326-440, k = 1) INCLUDING the configs[2] branches, in the reference's order of
    operations and of host random draws:
        rand -> [uniform alpha -> style s2t of x_s from (x_s, x_t_tea_ori), recover clamp]          (:347-351)
        rand -> [uniform alpha -> style t2s of x_t_tea from (x_t_tea, x_s_ori), recover clamp]      (:353-356)
        teacher forward, re-warp                                                                     (:358-372)
        occlusion: per sample with a confident key point: rand, [choice, randint, randint]           (:374-412)
        student forwards, re-warp, JointsMSE, confidences / rectify / k-th value mask, ConsLoss      (:414-432)
        backward, Adam, EMA                                                                          (:434-438)
    `style` = (vgg31, decoder) CPU modules or None; `recover` = (min[3], max[3]); `rng` = the numpy generator standing for the
    reference's global np.random (needed when style or occlusion is on).  Returns the losses, heat-maps, mask, the drawn alphas
    (None when a direction was not drawn), the occluded sample indices and the step's effective inputs."""
    from .occlusion_ref import occlude_ref
    from .style_ref import style_forward_ref
    student.train()
    teacher.train()
    optimizer.zero_grad()
    x_s_ori, x_t_tea_ori = x_s, x_t_tea
    a_s2t = a_t2s = None

    def _clamp(x):
        if recover is None:
            return x
        lo, hi = recover
        return torch.maximum(torch.minimum(x.permute(0, 2, 3, 1), hi), lo).permute(0, 3, 1, 2)

    with torch.no_grad():
        if style is not None and s2t_freq > rng.rand():
            a_s2t = rng.uniform(*s2t_alpha)
            x_s = _clamp(style_forward_ref(style[0], style[1], x_s_ori, x_t_tea_ori, a_s2t))
        if style is not None and t2s_freq > rng.rand():
            a_t2s = rng.uniform(*t2s_alpha)
            x_t_tea = _clamp(style_forward_ref(style[0], style[1], x_t_tea_ori, x_s_ori, a_t2s))
        if isinstance(x_t_tea, (list, tuple)):
            # k teacher views (`--k`, train_human.py:358-372): one teacher forward per view, each re-warped with its own aug_param,
            # `torch.mean(recons, dim=0)` per sample (style t2s above applies to every view in the reference; k > 1 is restated without style)
            assert style is None and len(x_t_tea) == len(aug_param_tea)
            recs = [_recon(teacher(xv), ap, ratio) for xv, ap in zip(x_t_tea, aug_param_tea)]
            y_t_tea_recon = torch.mean(torch.stack(recs), dim=0)
        else:
            y_t_tea = teacher(x_t_tea)
            y_t_tea_recon = _recon(y_t_tea, aug_param_tea, ratio)
        occluded = []
        if occlude_rate > -1:
            x_t_stu, occluded = occlude_ref(x_t_stu, y_t_tea_recon, aug_param_stu, ratio, image_size, occlude_rate, occlude_thresh,
                                            occlude_size, rng)
    y_s = student(x_s)
    y_t_stu = student(x_t_stu)
    y_t_stu_recon = _recon(y_t_stu, aug_param_stu, ratio)     # grid_sample(nearest) routes gradients
    loss_s = joints_mse_ref(y_s, label_s, weight_s)
    with torch.no_grad():
        tea_mask, activates, thr = conf_mask_ref(y_t_tea_recon, mask_ratio)
        rect = rectify_ref(y_t_tea_recon, sigma)
    loss_c = cons_loss_ref(y_t_stu_recon, rect, tea_mask=tea_mask)
    loss = loss_s + lambda_c * loss_c
    loss.backward()
    optimizer.step()
    ema_step_ref(list(teacher.parameters()), list(student.parameters()), alpha)
    return {"loss_all": loss.detach(), "loss_s": loss_s.detach(), "loss_c": loss_c.detach(), "y_s": y_s.detach(),
            "tea_mask": tea_mask, "thr": thr, "alpha_s2t": a_s2t, "alpha_t2s": a_t2s, "occluded": occluded, "x_s_in": x_s,
            "x_t_tea_in": x_t_tea, "x_t_stu_in": x_t_stu, "y_t_tea_recon": y_t_tea_recon}


def validate_ref(batches, model):
    """validate() of the reference (train_human.py:461-500) with its meters (lib/meter.py:8-40,65-82): eval mode, no grad;
    per-key-point running average weighted by batch size, a value of -1 (key point absent from the batch) is skipped;
    mean loss weighted by batch size.  Returns (acc_per_keypoint list, mean_loss)."""
    from .keypoints_ref import accuracy_ref
    from .losses_ref import joints_mse_ref
    was_training = model.training
    model.eval()
    sums = cnts = None
    loss_sum, n_seen = 0.0, 0
    with torch.no_grad():
        for x, label, weight in batches:
            y = model(x)
            loss = float(joints_mse_ref(y, label, weight))
            acc, _, _, _ = accuracy_ref(y.numpy(), label.numpy())
            n = x.shape[0]
            if sums is None:
                sums, cnts = [0.0] * len(acc), [0] * len(acc)
            for k, a in enumerate(acc):
                if a != -1:
                    sums[k] += float(a) * n
                    cnts[k] += n
            loss_sum += loss * n
            n_seen += n
    if was_training:
        model.train()
    return [s / c if c else 0 for s, c in zip(sums, cnts)], loss_sum / n_seen
