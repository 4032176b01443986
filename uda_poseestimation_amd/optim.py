"""Fused multi-tensor optimizers on MI355X kernels: drop-in torch.optim.Optimizer subclasses for the reference's
`Adam(student.parameters(), lr)` / `SGD(..., momentum=0.9, weight_decay=1e-4, nesterov=True)` (train_human.py:136-139).
One kernel launch sweeps all parameter tensors (28 B/param for Adam), instead of torch's per-group foreach chains.
"""
import torch

from . import _hip
from ._hip import check, lib, ptr
from .utils import _MultiTensorTable, _bump_versions


class _FusedBase(torch.optim.Optimizer):
    def _gather(self, group, state_names):
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return None
        for p in ps:
            _hip.require_cuda(p)
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                raise RuntimeError("fused optimizers need fp32 parameters and gradients")
            if p.grad.stride() != p.stride():
                p.grad = p.grad.clone(memory_format=torch.preserve_format).as_strided(p.shape, p.stride()).copy_(p.grad)
            st = self.state[p]
            for n in state_names:
                if n not in st:
                    st[n] = torch.zeros_like(p, memory_format=torch.preserve_format)
        lists = [[p.data for p in ps], [p.grad for p in ps]] + [[self.state[p][n] for p in ps] for n in state_names]
        key = _MultiTensorTable.key_of(lists)
        tab = group.get("_table")
        if tab is None or tab.key != key:
            tab = _MultiTensorTable(lists)
            group["_table"] = tab
        return ps, tab


class FusedAdam(_FusedBase):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale, step=0))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            got = self._gather(group, ("exp_avg", "exp_avg_sq"))
            if got is None:
                continue
            ps, t = got
            group["step"] += 1
            b1, b2 = group["betas"]
            # the step counter / bias corrections live on the device so that a captured step replays correctly
            ds = group.get("_dev_state")
            if ds is None or ds.device != ps[0].device:
                ds = torch.zeros(4, dtype=torch.float32, device=ps[0].device)
                ds[0] = group["step"] - 1
                group["_dev_state"] = ds
            check(lib().udapose_adam_multi(_hip.stream(), ptr(t.ptrs[0]), ptr(t.ptrs[1]), ptr(t.ptrs[2]), ptr(t.ptrs[3]), ptr(t.sizes), ptr(t.blk_t),
                                           ptr(t.blk_o), t.nblocks, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), int(group["step"]), float(group["grad_scale"]), ptr(ds)), "adam_multi")
            _bump_versions(ps)
        return loss


class FusedSGD(_FusedBase):
    def __init__(self, params, lr, momentum=0.9, weight_decay=0.0, nesterov=False, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov, grad_scale=grad_scale, step=0))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            got = self._gather(group, ("momentum_buffer",))
            if got is None:
                continue
            ps, t = got
            group["step"] += 1
            check(lib().udapose_sgd_multi(_hip.stream(), ptr(t.ptrs[0]), ptr(t.ptrs[1]), ptr(t.ptrs[2]), ptr(t.sizes), ptr(t.blk_t), ptr(t.blk_o),
                                          t.nblocks, float(group["lr"]), float(group["momentum"]), float(group["weight_decay"]),
                                          int(group["nesterov"]), int(group["step"] == 1), float(group["grad_scale"])), "sgd_multi")
            _bump_versions(ps)
        return loss
