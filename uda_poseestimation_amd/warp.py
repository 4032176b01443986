"""Batched heat-map / image re-warp on the device.

Stands in for the per-sample `tF.affine` triplets of the training loop (train_human.py:361-372, 385-391, 412, 418-423):
translate by (tx/ratio, ty/ratio); rotate by `angle` and scale; shear by (sx, sy) - three sequential nearest-neighbour
resamplings, evaluated by ONE kernel as a chain of index maps (results identical to the sequential form).  The inverse
affine matrices are built on the host in double precision exactly like torchvision's `_get_inverse_affine_matrix`
(torchvision itself is not a dependency); parameters arrive as the collated `aug_param` structure of
lib/transforms/keypoint_detection.py:139: [angle[N], [tx[N], ty[N]], [sx[N], sy[N]], scale[N]].
"""
import math

import torch

from . import _hip
from ._hip import check, lib, ptr


def inverse_affine_matrix(angle, translate, scale, shear):
    """torchvision.transforms.functional._get_inverse_affine_matrix with center=(0,0) (tensor inputs)."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0]
    m[2] += m[0] * (-tx) + m[1] * (-ty)
    m[5] += m[3] * (-tx) + m[4] * (-ty)
    return m


def _as_list(v, n):
    if torch.is_tensor(v):
        return [float(x) for x in v.tolist()]
    if isinstance(v, (list, tuple)):
        return [float(x) for x in v]
    return [float(v)] * n


def _inv_mats(angle, tx, ty, scale, sx, sy):
    """Vectorised (float64) inverse_affine_matrix for arrays of parameters -> [n, 6]."""
    import numpy as np
    rot, sx, sy = np.radians(angle), np.radians(sx), np.radians(sy)
    a = np.cos(rot - sy) / np.cos(sy)
    b = -np.cos(rot - sy) * np.tan(sx) / np.cos(sy) - np.sin(rot)
    c = np.sin(rot - sy) / np.cos(sy)
    d = -np.sin(rot - sy) * np.tan(sx) / np.cos(sy) + np.cos(rot)
    m0, m1, m3, m4 = d / scale, -b / scale, -c / scale, a / scale
    m2 = m0 * (-tx) + m1 * (-ty)
    m5 = m3 * (-tx) + m4 * (-ty)
    return np.stack([m0, m1, m2, m3, m4, m5], -1)


def pack_aug_param(aug_param, n, out=None):
    """The collated aug_param structure -> one [N,6] float64 host tensor (angle, tx, ty, shear_x, shear_y, scale): what
    udapose_recon_thetas reads.  `out`: a (pinned) host tensor to fill instead of a new one."""
    import numpy as np
    angle, (tx, ty), (sx, sy), scale = aug_param
    cols = [np.asarray(_as_list(v, n), np.float64) for v in (angle, tx, ty, sx, sy, scale)]
    t = out if out is not None else torch.empty(n, 6, dtype=torch.float64)
    t.numpy()[...] = np.stack(cols, 1)
    return t


def thetas_from_packed(params_dev, ratio, want_fwd=True, want_back=False, fwd=None, back=None):
    """Device-side matrices (udapose_recon_thetas) from a packed [N,6] float64 DEVICE tensor: (theta_fwd [N,3,6] | None,
    theta_back [N,1,6] | None); `fwd` / `back` name existing output tensors (a captured step writes its static ones)."""
    _hip.require_cuda(params_dev)
    assert params_dev.dtype == torch.float64 and params_dev.is_contiguous() and params_dev.shape[1] == 6
    n = params_dev.shape[0]
    if want_fwd and fwd is None:
        fwd = torch.empty(n, 3, 6, dtype=torch.float32, device=params_dev.device)
    if want_back and back is None:
        back = torch.empty(n, 1, 6, dtype=torch.float32, device=params_dev.device)
    check(lib().udapose_recon_thetas(_hip.stream(), ptr(params_dev), n, float(ratio), ptr(fwd) if want_fwd else None,
                                     ptr(back) if want_back else None), "recon_thetas")
    return (fwd if want_fwd else None), (back if want_back else None)


def recon_thetas(aug_param, n, ratio=1.0, device=None):
    """[N,3,6] fp32 matrices of the loop's translate -> rotate+scale -> shear chain for a collated aug_param.  With a CUDA
    `device` the matrices are computed there (double precision, udapose_recon_thetas); without, on the host."""
    import numpy as np
    if device is not None and torch.device(device).type == "cuda":
        return thetas_from_packed(pack_aug_param(aug_param, n).to(device), ratio)[0]
    angle, (tx, ty), (sx, sy), scale = aug_param
    angle, tx, ty, sx, sy, scale = (np.asarray(_as_list(v, n), np.float64) for v in (angle, tx, ty, sx, sy, scale))
    z, o = np.zeros(n), np.ones(n)
    th = np.stack([_inv_mats(z, tx / ratio, ty / ratio, o, z, z), _inv_mats(angle, z, z, scale, z, z), _inv_mats(z, z, z, o, sx, sy)], 1)
    t = torch.from_numpy(th.astype(np.float32))
    return t.to(device) if device is not None else t


def single_thetas(angle, translate, scale, shear, n, device=None):
    """[N,1,6]: one warp per sample (the occlusion path's warp-back, train_human.py:412)."""
    angle, scale = _as_list(angle, n), _as_list(scale, n)
    tx, ty = _as_list(translate[0], n), _as_list(translate[1], n)
    sx, sy = _as_list(shear[0], n), _as_list(shear[1], n)
    th = torch.empty(n, 1, 6, dtype=torch.float32)
    for i in range(n):
        th[i, 0] = torch.tensor(inverse_affine_matrix(angle[i], [tx[i], ty[i]], scale[i], [sx[i], sy[i]]))
    return th.to(device) if device is not None else th


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, theta):
        _hip.require_cuda(x, theta)
        xin = x.detach().float().contiguous()
        N, C, H, W = xin.shape
        th = theta.detach().float().contiguous()
        assert th.shape[0] == N and th.shape[2] == 6
        out = torch.empty_like(xin)
        check(lib().udapose_affine_nearest(_hip.stream(), ptr(xin), ptr(out), ptr(th), N, C, H, W, th.shape[1], 0), "affine_nearest")
        ctx.save_for_backward(th)
        ctx.in_dtype = x.dtype
        return out.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (th,) = ctx.saved_tensors
        gin = g.detach().float().contiguous()
        N, C, H, W = gin.shape
        dx = torch.empty_like(gin)
        check(lib().udapose_affine_nearest(_hip.stream(), ptr(gin), ptr(dx), ptr(th), N, C, H, W, th.shape[1], 1), "affine_nearest_bwd")
        return dx.to(ctx.in_dtype), None


def warp_chain(x, theta):
    """x [N,C,H,W]; theta [N,S,6] inverse affine matrices applied in order 0..S-1 (nearest, zero fill); differentiable."""
    return _WarpFn.apply(x, theta)


def mean_views(views):
    """Mean of k re-warped teacher heat-map tensors (train_human.py:361-372, `--k` > 1: `torch.mean(recons, dim=0)`): one launch, the k
    values added in view order in fp32 and divided by k."""
    import ctypes as C
    if len(views) == 1:
        return views[0]
    if len(views) > 8:
        raise NotImplementedError("at most 8 teacher views per step (the reference's --k defaults to 1)")
    vs = [v.detach().float().contiguous() for v in views]
    _hip.require_cuda(*vs)
    out = torch.empty_like(vs[0])
    arr = (C.c_void_p * len(vs))(*[v.data_ptr() for v in vs])
    check(lib().udapose_mean_views(_hip.stream(), arr, len(vs), ptr(out), out.numel()), "mean_views")
    return out


def affine(img, angle, translate, scale, shear, interpolation=None, fill=None):
    """`torchvision.transforms.functional.affine` for CUDA tensors as the reference's loop calls it (train_human.py:366-368,
    388-390, 412, 421-423): img [C,H,W] or [N,C,H,W], nearest interpolation, zero fill, centre of rotation = image centre;
    differentiable (the student's heat-maps are warped under autograd, :421-423).  One launch of the batched kernel with a
    single stage; the loop's three-call chains can use `warp_chain` / `recon_heatmaps` instead (one launch for the batch)."""
    if interpolation not in (None, 0, "nearest") and getattr(interpolation, "value", interpolation) != "nearest":
        raise NotImplementedError("only nearest interpolation (torchvision's default, the one the reference uses)")
    if fill not in (None, 0, 0.0):
        raise NotImplementedError("only zero fill")
    if not isinstance(shear, (list, tuple)):
        shear = [float(shear), 0.0]
    squeeze = img.dim() == 3
    x = img.unsqueeze(0) if squeeze else img
    m = inverse_affine_matrix(float(angle), [float(translate[0]), float(translate[1])], float(scale), [float(shear[0]), float(shear[1])])
    theta = torch.tensor(m, dtype=torch.float32).reshape(1, 1, 6).expand(x.shape[0], 1, 6).contiguous().to(x.device, non_blocking=True)
    out = warp_chain(x, theta)
    return out.squeeze(0) if squeeze else out


def recon_heatmaps(y, aug_param, ratio):
    """The loop's heat-map re-warp (train_human.py:361-372 / 418-423) for the whole batch."""
    return warp_chain(y, recon_thetas(aug_param, y.shape[0], ratio, y.device))


def occlude_keypoints(x_t_stu, y_t_tea_recon, aug_param_stu, ratio, image_size, occlude_rate, occlude_thresh, occlude_size, rng):
    """Adaptive key-point occlusion of the student's target images (train_human.py:374-412), batched on the device.

    Host side (exactly the reference's draws, in its order, from `rng` = np.random): per sample with at least one confidence
    >= thresh: rand() <= rate ?, choice(candidates), randint (rows), randint (cols).  The confidences / arg-max positions
    come back to the host once (the reference moves them with .cpu() as well).  Device side: the selected images are
    re-warped to the canonical frame (three nearest warps, with the reference's translate/ratio quirk), a random patch
    of the same image is pasted over the chosen key-point, and one inverse warp takes them back.
    Naming follows the reference's index math, not its variable names: `left/right` slice ROWS, `upper/bottom` slice COLUMNS.
    """
    import numpy as np
    from . import utils as U
    B, K, h, w = y_t_tea_recon.shape
    preds, conf = U.get_max_preds_torch_raw(y_t_tea_recon)          # un-masked positions, like view().argmax(-1)
    conf_np = conf.reshape(B, K).cpu().numpy()
    pos_np = preds.cpu().numpy().astype(np.int64)                   # [B,K,2] = (x, y)
    table = conf_np >= occlude_thresh
    angle, (tx, ty), (sx, sy), scale = aug_param_stu
    a_, tx_, ty_, sx_, sy_, sc_ = (_as_list(v, B) for v in (angle, tx, ty, sx, sy, scale))
    sel, boxes = [], []
    for b in range(B):
        if table[b].sum() > 0 and rng.rand() <= occlude_rate:
            cand = np.arange(0, K)[table[b]]
            c = rng.choice(cand)
            position = (pos_np[b, c] * ratio).astype(int)
            left, right = max(position[1] - occlude_size, 0), min(position[1] + occlude_size, image_size)
            upper, bottom = max(position[0] - occlude_size, 0), min(position[0] + occlude_size, image_size)
            left_src = rng.randint(image_size - (right - left) + 1)
            upper_src = rng.randint(image_size - (bottom - upper) + 1)
            sel.append(b)
            boxes.append([left, right, upper, bottom, left_src, upper_src])
    if not sel:
        return x_t_stu, sel
    dev = x_t_stu.device
    idx = torch.tensor(sel, device=dev)
    sub = [[v[i] for i in sel] for v in (a_, tx_, ty_, sx_, sy_, sc_)]
    n = len(sel)
    fwd = recon_thetas([sub[0], [sub[1], sub[2]], [sub[3], sub[4]], sub[5]], n, ratio, dev)
    back = single_thetas([-v for v in sub[0]], ([-v / ratio for v in sub[1]], [-v / ratio for v in sub[2]]), [1.0 / v for v in sub[5]],
                         ([-v for v in sub[3]], [-v for v in sub[4]]), n, dev)
    temp = warp_chain(x_t_stu.index_select(0, idx).float().contiguous(), fwd).contiguous()
    bx = torch.tensor(boxes, dtype=torch.int32, device=dev)
    maxel = int(max((b[1] - b[0]) * (b[3] - b[2]) for b in boxes)) * temp.shape[1]
    check(lib().udapose_patch_paste(_hip.stream(), ptr(temp), ptr(bx), n, temp.shape[1], temp.shape[2], temp.shape[3], maxel), "patch_paste")
    out = x_t_stu.clone()
    out.index_copy_(0, idx, warp_chain(temp, back).to(x_t_stu.dtype))
    return out, sel


def occlusion_back_thetas(aug_param_stu, n, ratio, device=None):
    """[N,1,6]: the warp back of the occlusion path (train_human.py:412): -angle, (-tx/ratio, -ty/ratio), 1/scale, (-shear)."""
    if device is not None and torch.device(device).type == "cuda":
        return thetas_from_packed(pack_aug_param(aug_param_stu, n).to(device), ratio, want_fwd=False, want_back=True)[1]
    angle, (tx, ty), (sx, sy), scale = aug_param_stu
    a_, tx_, ty_, sx_, sy_, sc_ = (_as_list(v, n) for v in (angle, tx, ty, sx, sy, scale))
    return single_thetas([-v for v in a_], ([-v / ratio for v in tx_], [-v / ratio for v in ty_]), [1.0 / v for v in sc_],
                         ([-v for v in sx_], [-v for v in sy_]), n, device)


def occlude_keypoints_device(x_t_stu, y_t_tea_recon, theta_fwd, theta_back, u, ratio, image_size, occlude_rate, occlude_thresh, occlude_size):
    """The same occlusion with its DECISIONS taken on the device (udapose_occlusion_pick): no read-back, fixed shapes - the step
    stays capturable.  theta_fwd = recon_thetas(aug_param_stu), theta_back = occlusion_back_thetas(aug_param_stu), u = [N,4]
    uniform [0,1) draws (rate test, key-point choice, patch row / column origin).  Every image goes through the warps; only the
    selected samples keep the result (the others are returned bit-identical).  Returns (images, apply [N] uint8)."""
    _hip.require_cuda(x_t_stu, y_t_tea_recon, theta_fwd, theta_back, u)
    B, K, h, w = y_t_tea_recon.shape
    hm = y_t_tea_recon.detach().float().contiguous()
    conf = torch.empty(B, K, dtype=torch.float32, device=hm.device)
    idx = torch.empty(B, K, dtype=torch.int32, device=hm.device)
    check(lib().udapose_heatmap_argmax(_hip.stream(), ptr(hm), B * K, h, w, ptr(conf), ptr(idx), None, None, None, 0), "heatmap_argmax")
    boxes = torch.empty(B, 6, dtype=torch.int32, device=hm.device)
    apply = torch.empty(B, dtype=torch.uint8, device=hm.device)
    uu = u.detach().float().contiguous()
    assert tuple(uu.shape) == (B, 4)
    check(lib().udapose_occlusion_pick(_hip.stream(), ptr(conf), ptr(idx), ptr(uu), B, K, w, float(ratio), int(image_size), float(occlude_rate),
                                       float(occlude_thresh), int(occlude_size), ptr(boxes), ptr(apply)), "occlusion_pick")
    x = x_t_stu.detach().float().contiguous()
    temp = warp_chain(x, theta_fwd).contiguous()
    C_ = temp.shape[1]
    check(lib().udapose_patch_paste(_hip.stream(), ptr(temp), ptr(boxes), B, C_, temp.shape[2], temp.shape[3], 4 * occlude_size * occlude_size * C_),
          "patch_paste")
    back = warp_chain(temp, theta_back).contiguous()
    out = torch.empty_like(x)
    check(lib().udapose_select_rows(_hip.stream(), ptr(out), ptr(back), ptr(x), ptr(apply), B, x[0].numel()), "select_rows")
    return out.to(x_t_stu.dtype), apply
