"""Which pixels of the loop's three-warp chain (train_human.py:366-368) may legitimately differ between two correct fp32 evaluations: shared by
tests/test_gpu_hotpath.py (device against the torchvision restatement) and tests/test_oracle_affine.py (the restatement against a float64 walk)."""
import numpy as np
import torch


def tie_exposed(ap, B, H, W, ratio, tol=1e-4, return_source=False):
    """For every output pixel of the three-warp chain (train_human.py:366-368): does its fp64 index walk pass within `tol` of a HALF-INTEGER
    source coordinate at some stage?  Nearest-neighbour resampling rounds there, and two correct fp32 evaluations of the same formula (ATen's
    grid generator through a BLAS bmm, the kernel's fused multiply-adds) may land on different sides: those pixels - and only those - may differ
    between the device and the oracle.  The walk follows the chain from the output back (last warp first) in float64 from the SAME float32
    matrices and stops at the first near-tie (the path beyond it is ambiguous) or when it leaves the map.  fp32 error of a source coordinate:
    a few ulp(1) on the normalised grid x W / 2 ~ 1e-5; tol = 1e-4.  VERDICT r5 #8."""
    from oracle.affine_ref import inverse_affine_matrix
    angle, (tx, ty), (sx, sy), sc = ap
    out = np.zeros((B, H, W), bool)
    src = np.full((B, H, W), -1, np.int64)            # flat source pixel of the float64 walk (-1: outside the map, or ambiguous)
    for n in range(B):
        mats = [inverse_affine_matrix(0.0, [float(tx[n]) / ratio, float(ty[n]) / ratio], 1.0, [0.0, 0.0]),
                inverse_affine_matrix(float(angle[n]), [0.0, 0.0], float(sc[n]), [0.0, 0.0]),
                inverse_affine_matrix(0.0, [0.0, 0.0], 1.0, [float(sx[n]), float(sy[n])])]
        py, px = np.mgrid[0:H, 0:W].astype(np.int64)
        alive = np.ones((H, W), bool)
        for m in reversed(mats):
            m = np.asarray(m, np.float32).astype(np.float64)
            bx, by = px - 0.5 * W + 0.5, py - 0.5 * H + 0.5
            gx = bx * (m[0] / (0.5 * W)) + by * (m[1] / (0.5 * W)) + m[2] / (0.5 * W)
            gy = bx * (m[3] / (0.5 * H)) + by * (m[4] / (0.5 * H)) + m[5] / (0.5 * H)
            ix, iy = ((gx + 1.0) * W - 1.0) * 0.5, ((gy + 1.0) * H - 1.0) * 0.5
            dx_, dy_ = np.abs(ix - np.floor(ix) - 0.5), np.abs(iy - np.floor(iy) - 0.5)
            near = (dx_ < tol) | (dy_ < tol)
            # (a pure translation on a power-of-two map is computed EXACTLY in fp32 by both sides - the grid scale is a power of two -, so an exact
            # half-pixel shift is an exact tie that both round half-to-even: not ambiguous)
            if m[0] == 1.0 and m[1] == 0.0 and m[3] == 0.0 and m[4] == 1.0 and (W & (W - 1)) == 0 and (H & (H - 1)) == 0:
                near = ((dx_ < tol) & (dx_ > 1e-12)) | ((dy_ < tol) & (dy_ > 1e-12))
            out[n] |= alive & near
            alive &= ~near
            rx, ry = np.rint(ix), np.rint(iy)
            alive &= (rx >= 0) & (rx <= W - 1) & (ry >= 0) & (ry <= H - 1)
            px, py = np.where(alive, rx, 0).astype(np.int64), np.where(alive, ry, 0).astype(np.int64)
        src[n] = np.where(alive, py * W + px, -1)
    if return_source:
        return torch.from_numpy(out), torch.from_numpy(src)
    return torch.from_numpy(out)


