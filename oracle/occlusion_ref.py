"""Adaptive key-point occlusion, CPU oracle (test-only): restatement of train_human.py:374-412.
Same host random draws in the same order (rand -> choice -> randint -> randint per selected sample); warps through the
torchvision restatement (oracle/affine_ref.py)."""
import numpy as np
import torch

from .affine_ref import affine_nearest_ref


def occlude_ref(x_t_stu, y_t_tea_recon, aug_param_stu, ratio, image_size, occlude_rate, occlude_thresh, occlude_size, rng):
    x_t_stu = x_t_stu.clone()
    b, k, h, w = y_t_tea_recon.shape
    conf = y_t_tea_recon.amax(dim=(2, 3))
    pred_position = y_t_tea_recon.view(b, k, -1).argmax(-1)
    pred_position = torch.stack([pred_position % w, pred_position // w], -1).numpy()
    conf_table = conf >= occlude_thresh
    angle, (trans_x, trans_y), (shear_x, shear_y), scale = aug_param_stu
    chosen = []
    for _b in range(b):
        if conf_table[_b].sum() > 0 and rng.rand() <= occlude_rate:
            a, tx, ty, sx, sy, sc = (float(v[_b]) for v in (angle, trans_x, trans_y, shear_x, shear_y, scale))
            temp = affine_nearest_ref(x_t_stu[_b], 0.0, [tx / ratio, ty / ratio], 1.0, [0.0, 0.0])
            temp = affine_nearest_ref(temp, a, [0.0, 0.0], sc, [0.0, 0.0])
            temp = affine_nearest_ref(temp, 0.0, [0.0, 0.0], 1.0, [sx, sy])
            candidates = torch.arange(0, k)[conf_table[_b]]
            _c = rng.choice(candidates)
            position = (pred_position[_b, _c] * ratio).astype(int)
            left, right = max(position[1] - occlude_size, 0), min(position[1] + occlude_size, image_size)
            upper, bottom = max(position[0] - occlude_size, 0), min(position[0] + occlude_size, image_size)
            left_src = rng.randint(image_size - (right - left) + 1)
            right_src = left_src + right - left
            upper_src = rng.randint(image_size - (bottom - upper) + 1)
            bottom_src = upper_src + bottom - upper
            temp[:, left:right, upper:bottom] = temp[:, left_src:right_src, upper_src:bottom_src].clone()
            x_t_stu[_b] = affine_nearest_ref(temp, -a, [-tx / ratio, -ty / ratio], 1.0 / sc, [-sx, -sy])
            chosen.append(_b)
    return x_t_stu, chosen


class _UniformDraws:
    """The rng interface occlude_ref uses, fed from four uniform [0,1) numbers of ONE sample: rand() = u0, choice(c) =
    c[floor(u1 * len(c))], the two randint(m) calls = floor(u2 * m), floor(u3 * m) (float32 products, clamped to m - 1) - the
    mapping of the device-side decision kernel (udapose_occlusion_pick), whose distribution is the reference's."""

    def __init__(self, u4):
        self.u = [np.float32(v) for v in u4]
        self.n_randint = 0

    def rand(self):
        return float(self.u[0])

    def _pick(self, u, m):
        return min(int(u * np.float32(m)), m - 1)

    def choice(self, cands):
        return cands[self._pick(self.u[1], len(cands))]

    def randint(self, m):
        self.n_randint += 1
        return self._pick(self.u[1 + self.n_randint], m)


def occlude_from_uniforms_ref(x_t_stu, y_t_tea_recon, aug_param_stu, ratio, image_size, occlude_rate, occlude_thresh, occlude_size, u):
    """occlude_ref sample by sample with the draws taken from u [B,4] instead of a sequential generator."""
    angle, (tx, ty), (sx, sy), scale = aug_param_stu
    out, chosen = x_t_stu.clone(), []
    for b in range(x_t_stu.shape[0]):
        ap = ([angle[b]], ([tx[b]], [ty[b]]), ([sx[b]], [sy[b]]), [scale[b]])
        o, c = occlude_ref(x_t_stu[b:b + 1], y_t_tea_recon[b:b + 1], ap, ratio, image_size, occlude_rate, occlude_thresh, occlude_size,
                           _UniformDraws(u[b].tolist()))
        out[b] = o[0]
        if c:
            chosen.append(b)
    return out, chosen
