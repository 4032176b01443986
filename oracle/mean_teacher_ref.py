"""Mean-teacher helpers, CPU oracle (test-only).

  * ema_step_ref      <- utils.py:21-25   (p = p*alpha ; p = p + src*(1-alpha), parameters only, two roundings)
  * ema_init_ref      <- utils.py:18-19   (teacher <- student)
  * rectify_ref       <- utils.py:77-109  (unnormalised Gaussian (6s+1)^2 stamped at each channel's arg-max)
  * conf_mask_ref     <- train_human.py:427-430 (k-th smallest of pre-rectify maxima; (mask*act) > thr)
  * generate_target_ref <- lib/datasets/util.py:12-70 (label heat-maps; weight 0 when centre is outside)
  * clamp_recover_ref <- train_human.py:32-33,276 (per-channel clamp to the normalised [0,1] image range)
  * draw_labelmap_ori_ref / animal_labels_ref <- lib/datasets/util.py:326-363 and its call loop lib/datasets/real_animal_all_mt.py:274-283,
                         animal_pose_mt.py:169-177,200-205 (the animal pipelines' label generator: int32-truncated centre, patch dropped
                         when ANY part of it is outside, Gaussian / Cauchy)
"""
import numpy as np
import torch

from .keypoints_ref import get_max_preds_torch_ref


def ema_init_ref(teacher_params, student_params):
    for p, s in zip(teacher_params, student_params):
        p.data[:] = s.data[:]


def ema_step_ref(teacher_params, student_params, alpha=0.999):
    oma = 1.0 - alpha
    for p, s in zip(teacher_params, student_params):
        p.data.mul_(alpha)
        p.data.add_(s.data * oma)


def rectify_ref(hm, sigma):
    B, C, H, W = hm.shape
    out = torch.zeros_like(hm)
    coord, _ = get_max_preds_torch_ref(hm)
    r = 3 * sigma
    size = 2 * r + 1
    ax = torch.arange(0, size, 1).float()
    c0 = size // 2
    g = torch.exp(-((ax[None, :] - c0) ** 2 + (ax[:, None] - c0) ** 2) / (2 * sigma ** 2))
    for b in range(B):
        for c in range(C):
            mx, my = coord[b, c, 0], coord[b, c, 1]
            if mx >= H or my >= W or mx < 0 or my < 0:   # utils.py:89 (h/w swapped, harmless when square)
                continue
            ulx, uly = int(mx - r), int(my - r)
            brx, bry = int(mx + r + 1), int(my + r + 1)
            gx0, gx1 = max(0, -ulx), min(brx, H) - ulx
            gy0, gy1 = max(0, -uly), min(bry, W) - uly
            ix0, ix1 = max(0, ulx), min(brx, H)
            iy0, iy1 = max(0, uly), min(bry, W)
            out[b, c, iy0:iy1, ix0:ix1] = g[gy0:gy1, gx0:gx1]
    return out


def conf_mask_ref(recon, mask_ratio, tea_mask=None):
    act = recon.amax(dim=(2, 3))
    k = int(mask_ratio * act.numel())
    thr = torch.kthvalue(act.reshape(-1), k)[0].item()
    if tea_mask is None:
        tea_mask = torch.ones_like(act)
    return (tea_mask * act) > thr, act, thr


def generate_target_ref(joints, joints_vis, heatmap_size, sigma, image_size):
    K = joints.shape[0]
    weight = np.ones((K, 1), np.float32)
    weight[:, 0] = joints_vis[:, 0]
    Wd, Hd = heatmap_size
    target = np.zeros((K, Hd, Wd), np.float32)
    r = sigma * 3
    stride = np.array(image_size) / np.array(heatmap_size)
    size = 2 * r + 1
    ax = np.arange(0, size, 1, np.float32)
    c0 = size // 2
    g = np.exp(-((ax[None, :] - c0) ** 2 + (ax[:, None] - c0) ** 2) / (2 * sigma ** 2))
    for j in range(K):
        mx = int(joints[j][0] / stride[0] + 0.5)
        my = int(joints[j][1] / stride[1] + 0.5)
        if mx >= Wd or my >= Hd or mx < 0 or my < 0:
            weight[j] = 0
            continue
        ulx, uly, brx, bry = int(mx - r), int(my - r), int(mx + r + 1), int(my + r + 1)
        if weight[j] > 0.5:
            target[j][max(0, uly):min(bry, Hd), max(0, ulx):min(brx, Wd)] = \
                g[max(0, -uly):min(bry, Hd) - uly, max(0, -ulx):min(brx, Wd) - ulx]
    return target, weight


def clamp_recover_ref(x, lo, hi):
    """x [N,3,H,W]; lo/hi [3]."""
    return torch.maximum(torch.minimum(x.permute(0, 2, 3, 1), hi), lo).permute(0, 3, 1, 2)


def draw_labelmap_ori_patch_ref(sigma, type="Gaussian"):
    """The (6*sigma + 1)^2 stamp of lib/datasets/util.py:343-352, in float64 as the reference builds it."""
    size = 6 * sigma + 1
    x = np.arange(0, size, 1, float)
    y = x[:, np.newaxis]
    x0 = y0 = size // 2
    if type == "Gaussian":
        return np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
    if type == "Cauchy":
        return sigma / (((x - x0) ** 2 + (y - y0) ** 2 + sigma ** 2) ** 1.5)
    raise ValueError(type)


def draw_labelmap_ori_ref(img, pt, sigma, type="Gaussian"):
    """lib/datasets/util.py:326-363.  img: float32 [H, W] array (modified in place and returned), pt: float32 (x, y[, ...]).
    The centre is truncated to int32 (`pt.to(torch.int32)`), the corners are `int(centre -+ 3 sigma (+ 1))` with the subtraction done
    in float32 (an int32 tensor minus a Python float), and the stamp is dropped - weight 0 - unless ALL of it lies inside the map."""
    H, W = img.shape
    cx, cy = int(np.trunc(np.float32(pt[0]))), int(np.trunc(np.float32(pt[1])))
    r = np.float32(3 * sigma)
    ulx, uly = int(np.float32(cx) - r), int(np.float32(cy) - r)
    brx, bry = int(np.float32(cx) + r + np.float32(1)), int(np.float32(cy) + r + np.float32(1))
    if brx >= W or bry >= H or ulx < 0 or uly < 0:
        return img, 0
    g = draw_labelmap_ori_patch_ref(sigma, type)
    gx0, gx1 = max(0, -ulx), min(brx, W) - ulx
    gy0, gy1 = max(0, -uly), min(bry, H) - uly
    img[max(0, uly):min(bry, H), max(0, ulx):min(brx, W)] = g[gy0:gy1, gx0:gx1]
    return img, 1


def animal_labels_ref(tpts, vis, gate, out_res, sigma, type="Gaussian"):
    """The label loop of the animal `_mt` datasets (real_animal_all_mt.py:274-283): tpts [K, 2+] float32 = the transformed key points
    (1-based, as `transform(pts + 1, ...)` returns them), vis [K] = pts[:, 2], gate [K] bool = `tpts[i, 1] > 0` BEFORE the transform.
    -> target [K, out_res, out_res] float32, weight [K, 1] float32 (= vis * drawn where the gate is open, vis elsewhere)."""
    K = tpts.shape[0]
    target = np.zeros((K, out_res, out_res), np.float32)
    weight = np.asarray(vis, np.float32).reshape(K, 1).copy()
    for i in range(K):
        if gate[i]:
            _, drawn = draw_labelmap_ori_ref(target[i], np.asarray(tpts[i], np.float32)[:2] - np.float32(1), sigma, type)
            weight[i, 0] *= drawn
    return target, weight
