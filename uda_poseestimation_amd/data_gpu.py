"""GPU data pipeline of the mean-teacher target views.

The reference feeds its loop from DataLoader workers (`--workers 1` by default, train_human.py:591) that run, per sample and
per view, PIL / torchvision transforms on the host (train_human.py:63-78, lib/datasets/human36m_mt.py:76-161):

    base crop (RandomResizedCrop = F.crop + PIL bilinear F.resize)            -> image [H,W,3] uint8, keypoint2d [K,2]
    student view:  RandomAffineRotation -> ColorJitter -> GaussianBlur(high=0) -> ToTensor -> Normalize, generate_target
    teacher views: the same with the teacher's ranges, k times

At ~1900 img/s that loader cannot keep up.  Here everything after the image decode runs on the device for the whole batch:
the base crop + resize (round 3: PIL's two-pass 8-bit resampler), the affine warp, the colour jitter, the tensor conversion and normalisation, and the Gaussian label maps are HIP kernels
(csrc/augment.hip) that reproduce PIL's integer arithmetic bit for bit; the random parameters (a few floats per sample) and
the key-point algebra (K x 2 per sample) are drawn / computed on the host exactly as lib/transforms/keypoint_detection.py
does, including the `aug_param` tuple that the loop consumes (inverse augmentation, :139).  The result has the collated
8-tuple layout of Appendix D / `default_collate`, ready for MeanTeacherTrainer.train_step or GraphedTrainStep.prefetch.

GaussianBlur (`--blur_stu / --blur_tea`, radius U(0, high) per sample; 0 by default = a copy) is PIL's three-pass box blur in
8.24 fixed point, reproduced bit for bit on the device (udapose_aug_gaussian_blur_u8).
"""
import math
import random

import numpy as np
import torch

from . import _hip
from ._hip import check, lib, ptr

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def inverse_affine_matrix_pil(center, angle, translate, scale, shear):
    """torchvision `_get_inverse_affine_matrix` as `F.affine` calls it for a PIL image: centre = (w/2, h/2), angle / shear in
    degrees, translation in pixels -> the 6 coefficients handed to Image.transform(AFFINE)."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def pil_fixed_coefficients(m):
    """PIL's 16.16 fixed-point form of an affine matrix (libImaging affine_fixed): FIX(v) = floor(v * 65536 + 0.5), with the
    half-pixel offset of the first sample folded into the two constants."""
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def pil_box_blur_params(radius, passes=3):
    """PIL's ImageFilter.GaussianBlur(radius) -> (r, ww, fw) of its box-blur passes (libImaging BoxBlur.c `_gaussian_blur_radius`
    and `ImagingLineBoxBlur*`, float32 / uint32 arithmetic as in the C source); radius 0 -> None (PIL returns a copy)."""
    if radius == 0:
        return None
    f = np.float32
    sigma2 = f(f(radius) * f(radius) / f(passes))
    L = f(np.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(np.floor((float(L) - 1.0) / 2.0))
    a = f(f(f(2) * l + f(1)) * f(f(l * f(l + f(1))) - f(f(3) * sigma2)))
    a = f(a / f(f(6) * f(sigma2 - f(f(l + f(1)) * f(l + f(1))))))
    fr = f(l + a)
    r = int(fr)
    ww = int(f(f(1 << 24) / f(fr * f(2) + f(1))))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww & 0xFFFFFFFF, fw & 0xFFFFFFFF


def pil_resample_coeffs(in_size, out_size):
    """PIL's BILINEAR resampling coefficients for one axis of Image.resize (libImaging Resample.c `precompute_coeffs` +
    `normalize_coeffs_8bpc`; box = the whole input): -> (bounds [out,2] int32 = (first source index, tap count), coef [out,ksize] int32
    with 22 fractional bits, ksize).  Python floats are C doubles, the statements follow the C source one by one."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale                   # bilinear_filter.support = 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coef = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k, ww = [], 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            w = 1.0 - t if t < 1.0 else 0.0
            k.append(w)
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
            v = k[x] * (1 << 22)
            coef[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, coef, ksize


def draw_resized_crop(rng, width, height, scale=(0.6, 1.3)):
    """RandomResizedCrop.get_params (lib/transforms/keypoint_detection.py:478-505): up to ten draws of a square crop of U(scale) x the
    image area (aspect ratio 1), then its corner; the whole image if none fits.  Same draws in the same order from `rng` (the reference
    uses the global `random` module).  -> (top, left, h, w)"""
    area = height * width
    for _ in range(10):
        target_area = rng.uniform(*scale) * area
        w = int(round(math.sqrt(target_area * 1)))
        h = int(round(math.sqrt(target_area / 1)))
        if 0 < w <= width and 0 < h <= height:
            i = rng.randint(0, height - h)
            j = rng.randint(0, width - w)
            return i, j, h, w
    return 0, 0, height, width


def transform_keypoints(kp, angle, shear_x, shear_y, trans_x, trans_y, scale, width, height):
    """The key-point side of `affine` (lib/transforms/keypoint_detection.py:141-165): rotate / shear / scale about the image
    centre, then translate."""
    ang, sx, sy = np.deg2rad(angle), np.deg2rad(shear_x), np.deg2rad(shear_y)
    a = np.cos(ang - sy) / np.cos(sy)
    b = (-np.cos(ang - sy) * np.tan(sx) / np.cos(sy) - np.sin(ang))
    c = np.sin(ang - sy) / np.cos(sy)
    d = (-np.sin(ang - sy) * np.tan(sx) / np.cos(sy) + np.cos(ang))
    rot = np.array([[scale * a, scale * b], [scale * c, scale * d]])
    kp = np.copy(kp).astype(np.float64)
    kp[:, 0] -= width / 2
    kp[:, 1] -= height / 2
    kp = np.matmul(rot, kp.T).T
    kp[:, 0] += width / 2 + trans_x
    kp[:, 1] += height / 2 + trans_y
    return kp


class ViewConfig:
    """Ranges of one view's augmentation = the reference's `--rotation_* --shear_* --translate_* --scale_* --color_*` flags
    (train_human.py:535-557)."""

    def __init__(self, rotation=180, shear=(-30, 30), translate=(0.05, 0.05), scale=(0.6, 1.3), color=0.25, blur=0.0):
        self.degrees = (-rotation, rotation) if isinstance(rotation, (int, float)) else tuple(rotation)
        self.shear = (-shear, shear) if isinstance(shear, (int, float)) else tuple(shear)
        self.translate = (translate, translate) if isinstance(translate, (int, float)) else tuple(translate)
        self.scale = (scale, scale) if isinstance(scale, (int, float)) else tuple(scale)
        self.color = float(color)
        self.blur = float(blur)          # T.GaussianBlur(high=blur): radius ~ U(0, blur) per sample (`--blur_stu / --blur_tea`)

    def draw_blur(self, rng):
        """GaussianBlur.__call__ (keypoint_detection.py:221-223): one uniform radius per sample (the reference draws from np.random)."""
        return rng.uniform(0.0, self.blur)

    def draw_affine(self, rng, img_size):
        """RandomAffineRotation.get_params (lib/transforms/keypoint_detection.py:396-412), same draws in the same order from
        `rng` (the reference uses the global `random` module)."""
        angle = rng.uniform(self.degrees[0], self.degrees[1])
        shear_x, shear_y = rng.uniform(self.shear[0], self.shear[1]), 0.0
        if len(self.shear) == 4:
            shear_y = rng.uniform(self.shear[2], self.shear[3])
        max_dx, max_dy = float(self.translate[0] * img_size[0]), float(self.translate[1] * img_size[1])
        trans_x = int(round(rng.uniform(-max_dx, max_dx)))
        trans_y = int(round(rng.uniform(-max_dy, max_dy)))
        scale = rng.uniform(self.scale[0], self.scale[1])
        return angle, shear_x, shear_y, trans_x, trans_y, scale

    def draw_jitter(self, rng):
        """ColorJitter(brightness=c, contrast=c, saturation=c): a random order of the three steps and one factor U(1-c, 1+c)
        each.  (torchvision draws these from torch's RNG in a version-dependent way; any order / factors are valid samples of
        the same distribution.)  -> ([op codes], [factors]) with op 1 brightness, 2 contrast, 3 saturation."""
        ops = [1, 2, 3]
        rng.shuffle(ops)
        c = self.color
        return ops, [rng.uniform(max(0.0, 1 - c), 1 + c) for _ in ops]


class TargetViewPipeline:
    """Batched device pipeline for the `_mt` datasets' student / teacher views."""

    def __init__(self, image_size=256, heatmap_size=64, sigma=2, k=1, student=None, teacher=None, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                 rng=None, resize_scale=(0.6, 1.3), np_rng=None):
        self.image_size, self.heatmap_size, self.sigma, self.k = int(image_size), int(heatmap_size), sigma, int(k)
        self.stu, self.tea = student or ViewConfig(), teacher or ViewConfig()
        self.mean, self.std = tuple(mean), tuple(std)
        self.rng = rng if rng is not None else random
        # the reference draws the blur radius from the GLOBAL np.random (lib/transforms/keypoint_detection.py:221), everything else of a view
        # from python's `random`: two generators, so a seeded pipeline consumes each stream as the reference does
        self.np_rng = np_rng if np_rng is not None else np.random
        self.resize_scale = tuple(resize_scale)          # `--resize-scale` (train_human.py:523)
        self._dev = {}
        self._coef_cache = {}

    # ------------------------------------------------------------------ device constants
    def _consts(self, device):
        c = self._dev.get(device)
        if c is None:
            tmp = self.sigma * 3
            size = 2 * tmp + 1
            if float(size) != int(size):
                raise NotImplementedError("label patch needs an integer 3*sigma")
            x = np.arange(0, size, 1, np.float32)
            y = x[:, np.newaxis]
            x0 = y0 = size // 2
            g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * self.sigma ** 2)).astype(np.float32)      # util.py:48-53
            c = dict(mean=torch.tensor(self.mean, dtype=torch.float32, device=device), std=torch.tensor(self.std, dtype=torch.float32, device=device),
                     patch=torch.from_numpy(np.ascontiguousarray(g)).to(device), rad=int(tmp))
            self._dev[device] = c
        return c

    # ------------------------------------------------------------------ building blocks (each = one kernel family)
    def warp_images(self, base_u8, params):
        """F.affine on the whole batch: base_u8 [N,H,W,3] uint8 CUDA, params = list of (angle, shear_x, shear_y, tx, ty, scale)."""
        _hip.require_cuda(base_u8)
        N, H, W, C3 = base_u8.shape
        assert C3 == 3 and base_u8.dtype == torch.uint8 and base_u8.is_contiguous()
        coef = [pil_fixed_coefficients(inverse_affine_matrix_pil((W * 0.5, H * 0.5), a, (tx, ty), sc, (shx, shy)))
                for (a, shx, shy, tx, ty, sc) in params]
        cdev = torch.tensor(coef, dtype=torch.int64).to(base_u8.device, non_blocking=True)
        out = torch.empty_like(base_u8)
        check(lib().udapose_aug_affine_u8(_hip.stream(), ptr(base_u8), ptr(out), ptr(cdev), N, H, W), "aug_affine_u8")
        return out

    def resized_crop(self, raw_u8, keypoints, boxes=None):
        """T.RandomResizedCrop(size=image_size, scale=resize_scale) on the whole batch (train_human.py:55,64): raw_u8 [N,Hs,Ws,3] uint8 CUDA
        (the decoded dataset images; the `_mt` datasets hold square crops, `resize` asserts it), keypoints [N,K,2] pixels;
        boxes = per-sample (top, left, h, w), drawn like the reference when None -> (base_u8 [N,S,S,3], key points [N,K,2])."""
        _hip.require_cuda(raw_u8)
        N, Hs, Ws, C3 = raw_u8.shape
        assert C3 == 3 and raw_u8.dtype == torch.uint8 and raw_u8.is_contiguous()
        S = self.image_size
        if boxes is None:
            boxes = [draw_resized_crop(self.rng, Ws, Hs, self.resize_scale) for _ in range(N)]
        tabs, ksize = [], 1
        for (top, left, h, w) in boxes:
            if not (0 <= top and 0 <= left and 0 < h and 0 < w and top + h <= Hs and left + w <= Ws):
                raise ValueError(f"crop box {(top, left, h, w)} outside a {Hs}x{Ws} image")
            if w != h:
                raise AssertionError("resize() asserts width == height (lib/transforms/keypoint_detection.py:47)")
            key = (w, S)
            if key not in self._coef_cache:
                self._coef_cache[key] = pil_resample_coeffs(w, S)
            tabs.append(self._coef_cache[key])
            ksize = max(ksize, tabs[-1][2])
        bounds = np.zeros((N, 2, S, 2), np.int32)
        coef = np.zeros((N, 2, S, ksize), np.int32)
        for n, (b, c, ks) in enumerate(tabs):
            bounds[n, 0] = bounds[n, 1] = b        # square crop to a square image: both axes share one table
            coef[n, :, :, :ks] = c
        dev = raw_u8.device
        box_d = torch.tensor(boxes, dtype=torch.int32).to(dev, non_blocking=True)
        bounds_d = torch.from_numpy(bounds).to(dev, non_blocking=True)
        coef_d = torch.from_numpy(coef).to(dev, non_blocking=True)
        tmp = torch.empty(N, Hs, S, 3, dtype=torch.uint8, device=dev)
        out = torch.empty(N, S, S, 3, dtype=torch.uint8, device=dev)
        check(lib().udapose_aug_resized_crop_u8(_hip.stream(), ptr(raw_u8), ptr(out), ptr(tmp), ptr(box_d), ptr(bounds_d), ptr(coef_d), N, Hs, Ws,
                                                S, ksize), "aug_resized_crop_u8")
        # key points: crop() shifts them, resize() scales them by size / width (keypoint_detection.py:47-49, 59-64)
        kp = np.array(keypoints, dtype=np.float64, copy=True)
        for n, (top, left, h, w) in enumerate(boxes):
            kp[n, :, 0] -= left
            kp[n, :, 1] -= top
            kp[n] *= float(S) / float(w)
        return out, kp

    def jitter_(self, img_u8, ops, factors):
        """ColorJitter in place: ops / factors = per-sample lists of three op codes / factors (applied in list order)."""
        N, H, W, _ = img_u8.shape
        dev = img_u8.device
        op_t = torch.tensor(ops, dtype=torch.int32).t().contiguous().to(dev, non_blocking=True)          # [3][N]
        f_t = torch.tensor(factors, dtype=torch.float32).t().contiguous().to(dev, non_blocking=True)
        scratch = torch.empty(N, dtype=torch.int32, device=dev)
        for i in range(op_t.shape[0]):
            check(lib().udapose_aug_color_op(_hip.stream(), ptr(img_u8), ptr(op_t[i]), ptr(f_t[i]), ptr(scratch), N, H * W), "aug_color_op")
        return img_u8

    def blur_(self, img_u8, radii):
        """T.GaussianBlur in place: radii = one Gaussian radius per sample (0: unchanged, like PIL's copy)."""
        N, H, W, _ = img_u8.shape
        prm = [pil_box_blur_params(float(r)) or (0xFFFFFFFF, 0, 0) for r in radii]
        if all(p[0] == 0xFFFFFFFF for p in prm):
            return img_u8
        pdev = torch.from_numpy(np.asarray(prm, dtype=np.uint32).view(np.int32)).to(img_u8.device, non_blocking=True)
        tmp = torch.empty_like(img_u8)
        check(lib().udapose_aug_gaussian_blur_u8(_hip.stream(), ptr(img_u8), ptr(tmp), ptr(pdev), N, H, W), "aug_gaussian_blur_u8")
        return img_u8

    def to_tensor(self, img_u8):
        N, H, W, _ = img_u8.shape
        c = self._consts(img_u8.device)
        out = torch.empty(N, 3, H, W, dtype=torch.float32, device=img_u8.device)
        check(lib().udapose_aug_to_tensor(_hip.stream(), ptr(img_u8), ptr(out), N, H * W, ptr(c["mean"]), ptr(c["std"])), "aug_to_tensor")
        return out

    def labels(self, keypoints, visible, device):
        """generate_target for [N,K,2] key points (numpy float64, image pixels) -> (target [N,K,Hh,Wh], weight [N,K,1]) on device."""
        c = self._consts(device)
        N, K, _ = keypoints.shape
        kp = torch.from_numpy(np.ascontiguousarray(keypoints, dtype=np.float64)).to(device, non_blocking=True)
        vis = torch.from_numpy(np.ascontiguousarray(visible, dtype=np.float32).reshape(N * K)).to(device, non_blocking=True)
        Hh = Wh = self.heatmap_size
        target = torch.empty(N, K, Hh, Wh, dtype=torch.float32, device=device)
        weight = torch.empty(N, K, 1, dtype=torch.float32, device=device)
        stride = self.image_size / self.heatmap_size
        check(lib().udapose_gaussian_labels(_hip.stream(), ptr(kp), ptr(vis), ptr(target), ptr(weight), N * K, Hh, Wh, float(stride), float(stride),
                                            ptr(c["patch"]), c["rad"]), "gaussian_labels")
        return target, weight

    def labels_animal(self, tpts, visible, gate, device, sigma=None, label_type="Gaussian", out_res=None):
        """The animal `_mt` datasets' label loop (lib/datasets/real_animal_all_mt.py:274-283 -> draw_labelmap_ori, lib/datasets/util.py:326-363)
        for a batch: tpts [N,K,2+] float32 = the transformed key points as `transform(pts + 1, ...)` returns them (1-based; the datasets pass
        `tpts - 1`), visible [N,K] = pts[:, 2], gate [N,K] bool = `tpts[i, 1] > 0` of the UN-transformed points
        -> (target [N,K,R,R], weight [N,K,1]) on the device.  The stamp is built as the reference builds it (float64, Gaussian or Cauchy)."""
        sigma = self.sigma if sigma is None else sigma
        R = int(out_res or self.heatmap_size)
        tp = np.ascontiguousarray(np.asarray(tpts, dtype=np.float32)[..., :2])
        N, K, _ = tp.shape
        size = 6 * sigma + 1
        x = np.arange(0, size, 1, float)
        y = x[:, np.newaxis]
        x0 = y0 = size // 2
        if label_type == "Gaussian":
            g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
        elif label_type == "Cauchy":
            g = sigma / (((x - x0) ** 2 + (y - y0) ** 2 + sigma ** 2) ** 1.5)
        else:
            raise ValueError("label_type must be 'Gaussian' or 'Cauchy' (lib/datasets/util.py:349-352)")
        patch = torch.from_numpy(np.ascontiguousarray(g, dtype=np.float32)).to(device, non_blocking=True)
        pt = torch.from_numpy(tp - np.float32(1)).to(device, non_blocking=True)
        vis = torch.from_numpy(np.ascontiguousarray(visible, dtype=np.float32).reshape(N * K)).to(device, non_blocking=True)
        gt = torch.from_numpy(np.ascontiguousarray(np.asarray(gate).reshape(N * K), dtype=np.uint8)).to(device, non_blocking=True)
        target = torch.empty(N, K, R, R, dtype=torch.float32, device=device)
        weight = torch.empty(N, K, 1, dtype=torch.float32, device=device)
        check(lib().udapose_draw_labelmap_ori(_hip.stream(), ptr(pt), ptr(vis), ptr(gt), ptr(target), ptr(weight), N * K, R, R,
                                              float(np.float32(3 * sigma)), ptr(patch), int(g.shape[0])), "draw_labelmap_ori")
        return target, weight

    # ------------------------------------------------------------------ one view of the whole batch
    def view(self, base_u8, keypoints, cfg, params=None, jitter=None, blur=None):
        """-> (image [N,3,H,W] fp32 normalised, key points [N,K,2], aug_param collated, target, weight)"""
        N, H, W, _ = base_u8.shape
        if params is None:
            params = [cfg.draw_affine(self.rng, (W, H)) for _ in range(N)]
        if jitter is None:
            jitter = [cfg.draw_jitter(self.rng) for _ in range(N)]
        img = self.warp_images(base_u8, params)
        self.jitter_(img, [j[0] for j in jitter], [j[1] for j in jitter])
        if blur is None and cfg.blur > 0:
            blur = [cfg.draw_blur(self.np_rng) for _ in range(N)]
        if blur is not None:
            self.blur_(img, blur)                     # (after the colour jitter, before ToTensor: train_human.py:67-71)
        x = self.to_tensor(img)
        kp = np.stack([transform_keypoints(keypoints[i], *params[i], W, H) for i in range(N)])
        # aug_param = the INVERSE augmentation (keypoint_detection.py:139), collated like default_collate does
        aug = [torch.tensor([-p[0] for p in params], dtype=torch.float64),
               [torch.tensor([-p[3] for p in params], dtype=torch.int64), torch.tensor([-p[4] for p in params], dtype=torch.int64)],
               [torch.tensor([-p[1] for p in params], dtype=torch.float64), torch.tensor([-p[2] for p in params], dtype=torch.float64)],
               torch.tensor([1.0 / p[5] for p in params], dtype=torch.float64)]
        vis = np.ones((N, keypoints.shape[1], 1), np.float32)
        target, weight = self.labels(kp, vis, base_u8.device)
        return x, kp, aug, target, weight

    def __call__(self, base_u8, keypoints, raw=False):
        """The collated 8-tuple of the `_mt` datasets (human36m_mt.py:161): (x_t_stu, target_stu, weight_stu, meta_stu, x_t_teas,
        targets_tea, weights_tea, metas_tea) with the meta fields the loop reads (train_human.py:330-345).  raw=True: `base_u8` are the
        decoded dataset images and the base transform (RandomResizedCrop, human36m_mt.py:86) runs here first."""
        keypoints = np.asarray(keypoints, dtype=np.float64)
        if raw:
            base_u8, keypoints = self.resized_crop(base_u8, keypoints)
        x_s, kp_s, aug_s, t_s, w_s = self.view(base_u8, keypoints, self.stu)
        vis = np.ones((keypoints.shape[0], keypoints.shape[1], 1), np.float32)
        t_ori, w_ori = self.labels(keypoints, vis, base_u8.device)
        meta_stu = {"aug_param_stu": aug_s, "keypoint2d_stu": kp_s, "keypoint2d_ori": keypoints, "target_ori": t_ori, "target_weight_ori": w_ori}
        xs, ts, ws, metas = [], [], [], []
        for _ in range(self.k):
            x_t, kp_t, aug_t, t_t, w_t = self.view(base_u8, keypoints, self.tea)
            xs.append(x_t); ts.append(t_t); ws.append(w_t)
            metas.append({"aug_param_tea": aug_t, "keypoint2d_tea": kp_t})
        return x_s, t_s, w_s, meta_stu, xs, ts, ws, metas
