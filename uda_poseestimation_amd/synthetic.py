"""Synthetic batches with the shapes and statistics of the reference's data contract: used by
bench.py, smoke() and the tests because no dataset is available offline.  Host-side numpy only, except
`trained_like_state_dict`, which trains on the device.

Label heat-maps follow lib/datasets/util.py:12-70 (`generate_target`): centre int(kp/stride + 0.5), un-normalised
(6*sigma+1)^2 Gaussian clipped at the borders, weight 0 when the centre falls outside.
"""
import numpy as np
import torch

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)


def gaussian_labels(keypoints, visible, heatmap_size, sigma, image_size):
    """keypoints [K,2] px, visible [K,1] -> (target [K,H,W] f32, weight [K,1] f32)."""
    K = keypoints.shape[0]
    Wd, Hd = heatmap_size
    target = np.zeros((K, Hd, Wd), np.float32)
    weight = np.ones((K, 1), np.float32)
    weight[:, 0] = visible[:, 0]
    rad = sigma * 3
    size = 2 * rad + 1
    ax = np.arange(0, size, 1, np.float32)
    c0 = size // 2
    g = np.exp(-((ax[None, :] - c0) ** 2 + (ax[:, None] - c0) ** 2) / (2 * sigma ** 2))
    stride = np.array(image_size) / np.array(heatmap_size)
    for j in range(K):
        mx = int(keypoints[j][0] / stride[0] + 0.5)
        my = int(keypoints[j][1] / stride[1] + 0.5)
        if mx >= Wd or my >= Hd or mx < 0 or my < 0:
            weight[j] = 0
            continue
        ulx, uly, brx, bry = int(mx - rad), int(my - rad), int(mx + rad + 1), int(my + rad + 1)
        if weight[j] > 0.5:
            target[j][max(0, uly):min(bry, Hd), max(0, ulx):min(brx, Wd)] = \
                g[max(0, -uly):min(bry, Hd) - uly, max(0, -ulx):min(brx, Wd) - ulx]
    return target, weight


def images(n, size, seed, normalise="imagenet"):
    """N(0,1) noise clipped to the range a normalised [0,1] image can take."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, size, size, generator=g)
    if normalise == "imagenet":
        lo = torch.tensor((0 - IMAGENET_MEAN) / IMAGENET_STD).view(1, 3, 1, 1)
        hi = torch.tensor((1 - IMAGENET_MEAN) / IMAGENET_STD).view(1, 3, 1, 1)
    else:   # animal pipeline: mean subtraction only (train_animal.py:34-35)
        m = torch.tensor([0.3999, 0.3909, 0.3871]).view(1, 3, 1, 1)
        lo, hi = -m, 1 - m
    return torch.maximum(torch.minimum(x, hi), lo)


def aug_params(n, rs, max_angle=60.0, max_shear=30.0, max_trans=12.8, scale=(0.6, 1.3)):
    """Collated inverse-augmentation tuple as produced by lib/transforms/keypoint_detection.py:139,396-412."""
    angle = torch.tensor(-rs.uniform(-max_angle, max_angle, n), dtype=torch.float64)
    tx = torch.tensor(-np.round(rs.uniform(-max_trans, max_trans, n)).astype(np.int64))
    ty = torch.tensor(-np.round(rs.uniform(-max_trans, max_trans, n)).astype(np.int64))
    sx = torch.tensor(-rs.uniform(-max_shear, max_shear, n), dtype=torch.float64)
    sy = torch.zeros(n, dtype=torch.float64)
    sc = torch.tensor(1.0 / rs.uniform(scale[0], scale[1], n), dtype=torch.float64)
    return [angle, [tx, ty], [sx, sy], sc]


def mean_teacher_batch(n, num_keypoints=16, image_size=256, heatmap_size=64, sigma=2, seed=0, normalise="imagenet"):
    """One (source, target) batch pair with the fields the loop consumes (train_human.py:329-340)."""
    rs = np.random.RandomState(seed)
    kp = rs.uniform(0, image_size, size=(n, num_keypoints, 2)).astype(np.float32)
    vis = np.ones((num_keypoints, 1), np.float32)
    lab = [gaussian_labels(kp[i], vis, (heatmap_size, heatmap_size), sigma, (image_size, image_size)) for i in range(n)]
    return {
        "x_s": images(n, image_size, seed, normalise),
        "label_s": torch.from_numpy(np.stack([l[0] for l in lab])),
        "weight_s": torch.from_numpy(np.stack([l[1] for l in lab])),
        "x_t_stu": images(n, image_size, seed + 1, normalise),
        "x_t_tea": images(n, image_size, seed + 2, normalise),
        "aug_param_stu": aug_params(n, rs),
        "aug_param_tea": aug_params(n, rs),
    }


# 16+ well separated colours in the normalised-image range (one per key point): what makes `keypoint_images` learnable
_BLOB_LEVELS = (-1.6, 0.2, 2.0)


def keypoint_images(keypoints, image_size, seed, noise=0.35, radius=7.0):
    """Images whose CONTENT determines the labels (unlike `images`, whose noise carries no information about the key points): weak
    noise plus one Gaussian blob per key point at its location, coloured by the key-point index.  A network can learn the
    task in a few hundred steps and generalises to unseen images - used to produce trained-like weights for the parity tests.
    keypoints [N,K,2] px -> [N,3,S,S] f32."""
    n, K, _ = keypoints.shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, image_size, image_size, generator=g) * noise
    cols = [(a, b, c) for a in _BLOB_LEVELS for b in _BLOB_LEVELS for c in _BLOB_LEVELS if not (a == b == c)]
    ys = torch.arange(image_size, dtype=torch.float32).view(1, -1, 1)
    xs = torch.arange(image_size, dtype=torch.float32).view(1, 1, -1)
    kp = torch.as_tensor(keypoints, dtype=torch.float32)
    for k in range(K):
        col = torch.tensor(cols[k % len(cols)], dtype=torch.float32).view(1, 3, 1, 1)
        w = torch.exp(-((xs - kp[:, k, 0].view(-1, 1, 1)) ** 2 + (ys - kp[:, k, 1].view(-1, 1, 1)) ** 2) / (2 * radius ** 2)).unsqueeze(1)
        x = x * (1 - w) + col * w
    lo = torch.tensor((0 - IMAGENET_MEAN) / IMAGENET_STD).view(1, 3, 1, 1)
    hi = torch.tensor((1 - IMAGENET_MEAN) / IMAGENET_STD).view(1, 3, 1, 1)
    return torch.maximum(torch.minimum(x, hi), lo)


def keypoint_batch(n, num_keypoints=16, image_size=256, heatmap_size=64, sigma=2, seed=0):
    """(x [n,3,S,S], label [n,K,h,w], weight [n,K,1]) with learnable content (keypoint_images); key points kept off the border."""
    rs = np.random.RandomState(seed)
    kp = rs.uniform(0.08 * image_size, 0.92 * image_size, size=(n, num_keypoints, 2)).astype(np.float32)
    vis = np.ones((num_keypoints, 1), np.float32)
    lab = [gaussian_labels(kp[i], vis, (heatmap_size, heatmap_size), sigma, (image_size, image_size)) for i in range(n)]
    return (keypoint_images(kp, image_size, seed), torch.from_numpy(np.stack([l[0] for l in lab])),
            torch.from_numpy(np.stack([l[1] for l in lab])))


def keypoint_mean_teacher_batch(n, num_keypoints=16, image_size=256, sigma=2, seed=0):
    """A mean-teacher batch (the fields of train_human.py:329-340) whose images carry key points a network trained on `keypoint_batch`
    recognises: source images + labels, two further image sets as the student's and the teacher's target views, random
    inverse-augmentation tuples."""
    K, S = num_keypoints, image_size
    x_s, lab, wt = keypoint_batch(n, num_keypoints=K, image_size=S, heatmap_size=S // 4, sigma=sigma, seed=seed)
    x_t_stu = keypoint_batch(n, num_keypoints=K, image_size=S, heatmap_size=S // 4, sigma=sigma, seed=seed + 1)[0]
    x_t_tea = keypoint_batch(n, num_keypoints=K, image_size=S, heatmap_size=S // 4, sigma=sigma, seed=seed + 2)[0]
    rs = np.random.RandomState(seed + 3)
    return {"x_s": x_s, "label_s": lab, "weight_s": wt, "x_t_stu": x_t_stu, "x_t_tea": x_t_tea,
            "aug_param_stu": aug_params(n, rs), "aug_param_tea": aug_params(n, rs)}


def trained_like_state_dict(num_keypoints=16, steps=400, seed=0, lr=2e-4, arch="pose_resnet101"):
    """A TRAINED-LIKE pose network: the reference initialisation trained ON THE DEVICE (fp16 student precision under the device-side
    GradScaler, Adam, `steps` steps of 8 fresh images each from keypoint_batch), so that whole-network parity is not measured on the
    worst case for 16-bit storage only (a randomly initialised train-mode-BatchNorm ResNet amplifies any rounding ~1.25x per bottleneck;
    no pretrained weights are available offline).  Returns (state_dict on the CPU, loss history, held-out eval-mode PCK@0.05).  With
    the deterministic weight-gradient accumulation (round 6) every run of this function yields the SAME network, bit for bit."""
    from . import optim as fused_optim
    from .lib import keypoint_detection as kd
    from .lib import models
    from .lib.models.loss import JointsMSELoss
    torch.manual_seed(seed)
    net = models.__dict__[arch](num_keypoints=num_keypoints, pretrained_backbone=False).cuda().train()
    net.precision = "fp16"
    opt = fused_optim.FusedAdam(net.parameters(), lr=lr, dynamic_loss_scale=True, init_scale=1024.0)
    crit = JointsMSELoss()
    hist = []
    for it in range(steps):
        x, lab, wt = (t.cuda() for t in keypoint_batch(8, num_keypoints=num_keypoints, seed=1000 + it))
        opt.zero_grad()
        loss = crit(net(x), lab, wt)
        opt.scale_loss(loss).backward()
        opt.step()
        if it % 80 == 0 or it == steps - 1:
            hist.append(float(loss.detach()))
    x, lab, wt = (t.cuda() for t in keypoint_batch(8, num_keypoints=num_keypoints, seed=5))
    net.eval()
    with torch.no_grad():
        pck = float(kd.accuracy(net(x), lab)[1])
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    del net, opt
    torch.cuda.empty_cache()
    return sd, hist, pck
