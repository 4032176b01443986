"""CPU restatement of one training step of the reference loop (oracle / CPU baseline, test-only).

  * pretrain_step_ref  <- train_human.py:262-289 (source-only: forward, JointsMSE, backward, Adam)
  * train_step_ref     <- train_human.py:326-440 (k=1, no style, no occlusion: BASELINE.json configs[1])
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
    student.train()
    teacher.train()
    optimizer.zero_grad()
    with torch.no_grad():
        y_t_tea = teacher(x_t_tea)
        y_t_tea_recon = _recon(y_t_tea, aug_param_tea, ratio)
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
            "tea_mask": tea_mask, "thr": thr}


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
