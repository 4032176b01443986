"""Heat-map losses (API mirror of the reference's lib/models/loss.py:11-49,119-132) on MI355X kernels.

Only the two classes the training scripts instantiate (train_human.py:133-134) exist; the reference's unused
JointsKLLoss / EntLoss / ConsSoftmaxLoss / ConsKLLoss / CoralLoss are out of scope.
Each forward is one sweep over the operands (per-(b,k) row partial + a tiny row reduction), each backward one sweep.
"""
import torch
import torch.nn as nn

from ... import _hip
from ..._hip import check, lib, ptr


def _rows(t):
    B, K = t.shape[:2]
    return B * K, t.numel() // (B * K)


def _f32c(t):
    return t.detach().float().contiguous()


class _JointsMSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target, weight, reduce_mean):
        _hip.require_cuda(output, target, weight)
        R, HW = _rows(output)
        o, t = _f32c(output), _f32c(target)
        w = None if weight is None else _f32c(weight).reshape(-1)
        if w is not None and w.numel() != R:
            raise ValueError("target_weight must have B*K elements")
        rows = torch.empty(R, dtype=torch.float32, device=o.device)
        mean = torch.empty((), dtype=torch.float32, device=o.device)
        check(lib().udapose_joints_mse_fwd(_hip.stream(), ptr(o), ptr(t), ptr(w), R, HW, ptr(rows), ptr(mean)), "joints_mse_fwd")
        ctx.save_for_backward(o, t, w if w is not None else torch.empty(0, device=o.device))
        ctx.has_w, ctx.reduce_mean, ctx.shape, ctx.in_dtype = w is not None, reduce_mean, output.shape, output.dtype
        return mean if reduce_mean else rows.reshape(output.shape[0], output.shape[1])

    @staticmethod
    def backward(ctx, g):
        o, t, w = ctx.saved_tensors
        if not ctx.reduce_mean:
            raise NotImplementedError("backward of reduction='none' is not on the hot path")
        R, HW = _rows(o)
        d = torch.empty_like(o)
        gs = g.detach().float().reshape(1).contiguous()
        check(lib().udapose_joints_mse_bwd(_hip.stream(), ptr(o), ptr(t), ptr(w) if ctx.has_w else None, ptr(gs), R, HW, ptr(d)), "joints_mse_bwd")
        return d.reshape(ctx.shape).to(ctx.in_dtype), None, None, None


class JointsMSELoss(nn.Module):
    """0.5 * (pred - gt)^2 * target_weight[b,k], mean over everything ('mean') or per-(b,k) means ('none')."""

    def __init__(self, reduction='mean'):
        super(JointsMSELoss, self).__init__()
        self.reduction = reduction

    def forward(self, output, target, target_weight=None):
        if self.reduction == 'mean':
            return _JointsMSEFn.apply(output, target, target_weight, True)
        elif self.reduction == 'none':
            return _JointsMSEFn.apply(output, target, target_weight, False)
        # the reference silently returns None for any other string (loss.py:46-49)


class _ConsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stu, tea, mask, valid):
        _hip.require_cuda(stu, tea, mask, valid)
        R, HW = _rows(stu)
        K = stu.shape[1]
        s, t = _f32c(stu), _f32c(tea)
        if mask is not None and mask.dtype == torch.bool and mask.is_contiguous():
            m = mask.detach().view(torch.uint8).reshape(-1)        # (bool storage is 0 / 1: no conversion launches)
        else:
            m = None if mask is None else (mask.detach() != 0).to(torch.uint8).reshape(-1).contiguous()
        rows = torch.empty(R, dtype=torch.float32, device=s.device)
        mean = torch.empty((), dtype=torch.float32, device=s.device)
        v = cnt = None
        if valid is None:
            check(lib().udapose_cons_loss_fwd(_hip.stream(), ptr(s), ptr(t), ptr(m), R, HW, ptr(rows), ptr(mean)), "cons_loss_fwd")
        else:
            # loss_map[valid_mask].mean() (loss.py:129-130): boolean selection over (b, h, w)
            if tuple(valid.shape) != (stu.shape[0],) + tuple(stu.shape[2:]):
                raise IndexError(f"valid_mask shape {tuple(valid.shape)} does not index loss_map {(stu.shape[0],) + tuple(stu.shape[2:])}")
            v = (valid.detach() != 0).to(torch.uint8).reshape(-1).contiguous()
            cnt = torch.empty((), dtype=torch.float32, device=s.device)
            check(lib().udapose_mask_count(_hip.stream(), ptr(v), v.numel(), ptr(cnt)), "mask_count")
            check(lib().udapose_cons_loss_valid_fwd(_hip.stream(), ptr(s), ptr(t), ptr(m), ptr(v), ptr(cnt), R, K, HW, ptr(rows), ptr(mean)),
                  "cons_loss_valid_fwd")
        empty = torch.empty(0, dtype=torch.uint8, device=s.device)
        ctx.save_for_backward(s, t, m if m is not None else empty, v if v is not None else empty,
                              cnt if cnt is not None else torch.empty(0, device=s.device))
        ctx.has_m, ctx.has_v, ctx.shape, ctx.in_dtype = m is not None, v is not None, stu.shape, stu.dtype
        return mean

    @staticmethod
    def backward(ctx, g):
        s, t, m, v, cnt = ctx.saved_tensors
        R, HW = _rows(s)
        d = torch.empty_like(s)
        gs = g.detach().float().reshape(1).contiguous()
        if ctx.has_v:
            check(lib().udapose_cons_loss_valid_bwd(_hip.stream(), ptr(s), ptr(t), ptr(m) if ctx.has_m else None, ptr(v), ptr(cnt), ptr(gs), R,
                                                    ctx.shape[1], HW, ptr(d)), "cons_loss_valid_bwd")
        else:
            check(lib().udapose_cons_loss_bwd(_hip.stream(), ptr(s), ptr(t), ptr(m) if ctx.has_m else None, ptr(gs), R, HW, ptr(d)), "cons_loss_bwd")
        return d.reshape(ctx.shape).to(ctx.in_dtype), None, None, None


class ConsLoss(nn.Module):
    """mean over (b,h,w) of mean_c (mask[b,c] * (stu - tea))^2  ==  sum(mask*(stu-tea)^2) / (B*C*H*W); with `valid_mask`
    (bool [B,H,W]) the mean runs over the selected positions only (loss.py:129-130)."""

    def __init__(self):
        super(ConsLoss, self).__init__()

    def forward(self, stu_out, tea_out, valid_mask=None, tea_mask=None):
        return _ConsFn.apply(stu_out, tea_out, tea_mask, valid_mask)
