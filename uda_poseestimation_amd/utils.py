"""Mean-teacher helpers (API mirror of the reference's utils.py:9-109) on MI355X kernels."""
if not __package__:
    # `from utils import *` with this package's directory on sys.path (train_human.py:29): bind the name `utils` to
    # uda_poseestimation_amd.utils (import returns that module; this top-level copy is discarded)
    import _dropin
    _dropin.alias(["utils"])
    __package__ = "uda_poseestimation_amd"
import ctypes as C

import numpy as np
import torch

from . import _hip
from ._hip import check, lib, ptr


class _MultiTensorTable:
    """Device tables (addresses, sizes, block -> (tensor, offset)) for the multi-tensor sweeps."""

    def __init__(self, tensor_lists):
        dev = tensor_lists[0][0].device
        chunk = lib().udapose_multi_chunk()
        sizes = [t.numel() for t in tensor_lists[0]]
        blk_t, blk_o = [], []
        for i, n in enumerate(sizes):
            for off in range(0, n, chunk):
                blk_t.append(i)
                blk_o.append(off)
        self.key = tuple(t.data_ptr() for lst in tensor_lists for t in (lst[0], lst[-1])) + (len(sizes),)
        self.ptrs = [torch.tensor([t.data_ptr() for t in lst], dtype=torch.int64, device=dev) for lst in tensor_lists]
        self.sizes = torch.tensor(sizes, dtype=torch.int64, device=dev)
        self.blk_t = torch.tensor(blk_t, dtype=torch.int32, device=dev)
        self.blk_o = torch.tensor(blk_o, dtype=torch.int64, device=dev)
        self.nblocks = len(blk_t)

    @staticmethod
    def key_of(tensor_lists):
        return tuple(t.data_ptr() for lst in tensor_lists for t in (lst[0], lst[-1])) + (len(tensor_lists[0]),)


class OldWeightEMA(object):
    """Exponential moving average of the student's PARAMETERS into the teacher (utils.py:9-25), one fused sweep."""

    def __init__(self, target_net, source_net, alpha=0.999):
        self.target_params = list(target_net.parameters())
        self.source_params = list(source_net.parameters())
        self.alpha = alpha
        self._table = None
        for p, src_p in zip(self.target_params, self.source_params):
            p.data[:] = src_p.data[:]
        _bump_versions(self.target_params)       # (`.data` writes are invisible to the version counters the pack cache reads)

    def step(self):
        tp = [p.data for p in self.target_params]
        sp = [p.data for p in self.source_params]
        for a, b in zip(tp, sp):
            _hip.require_cuda(a, b)
            if a.dtype != torch.float32 or b.dtype != torch.float32 or a.stride() != b.stride() or a.numel() != b.numel():
                raise RuntimeError("OldWeightEMA needs fp32 teacher/student parameters of identical layout")
        key = _MultiTensorTable.key_of([tp, sp])
        if self._table is None or self._table.key != key:
            self._table = _MultiTensorTable([tp, sp])
        t = self._table
        one_minus_alpha = 1.0 - self.alpha
        check(lib().udapose_ema_multi(_hip.stream(), ptr(t.ptrs[0]), ptr(t.ptrs[1]), ptr(t.sizes), ptr(t.blk_t), ptr(t.blk_o), t.nblocks,
                                      float(self.alpha), float(one_minus_alpha)), "ema_multi")
        _bump_versions(self.target_params)


def _bump_versions(params):
    """The kernels wrote through raw pointers: bump torch's version counters so that autograd and the executor's
    weight-pack cache (PoseResNet._run_forward) see that the data changed."""
    params = list(params)
    setter = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
    if setter is not None:
        setter(params, [p._version + 1 for p in params])
    else:   # older torch: a zero-size in-place op through a version-sharing alias
        for p in params:
            p.detach()[:0].zero_()


def get_max_preds_torch_raw(batch_heatmaps):
    """(x = idx % W, y = idx // W) [B,K,2] and the maxima [B,K,1] WITHOUT the max>0 masking: what the occlusion code
    computes inline with amax / view().argmax(-1) (train_human.py:378-381)."""
    _hip.require_cuda(batch_heatmaps)
    hm = batch_heatmaps.detach().float().contiguous()
    B, K, H, W = hm.shape
    idx = torch.empty(B, K, dtype=torch.int32, device=hm.device)
    maxv = torch.empty(B, K, 1, dtype=torch.float32, device=hm.device)
    check(lib().udapose_heatmap_argmax(_hip.stream(), ptr(hm), B * K, H, W, ptr(maxv), ptr(idx), None, None, None, 0), "heatmap_argmax")
    idx = idx.long()
    return torch.stack([idx % W, idx // W], -1), maxv


def get_max_preds_torch(batch_heatmaps):
    """utils.py:54-75: (preds [B,K,2] float (x,y) zeroed where max<=0, maxvals [B,K,1])."""
    _hip.require_cuda(batch_heatmaps)
    hm = batch_heatmaps.detach().float().contiguous()
    B, K, H, W = hm.shape
    preds = torch.empty(B, K, 2, dtype=torch.float32, device=hm.device)
    maxv = torch.empty(B, K, 1, dtype=torch.float32, device=hm.device)
    check(lib().udapose_heatmap_argmax(_hip.stream(), ptr(hm), B * K, H, W, ptr(maxv), None, ptr(preds), None, None, 0), "heatmap_argmax")
    return preds, maxv.to(batch_heatmaps.dtype)


_PATCH_CACHE = {}


def _gauss_patch(sigma, device):
    """The (6*sigma+1)^2 un-normalised Gaussian exactly as utils.py:93-98 builds it (host, once per sigma)."""
    key = (float(sigma), type(sigma) is int, str(device))
    if key not in _PATCH_CACHE:
        tmp_size = 3 * sigma
        if float(tmp_size) != int(tmp_size):
            raise NotImplementedError("rectify on the device needs 3*sigma to be an integer (reference uses sigma=2 or 1.0)")
        size = 2 * tmp_size + 1
        x = torch.arange(0, size, 1).float()
        y = x.unsqueeze(1)
        x0 = y0 = size // 2
        g = torch.exp(- ((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
        _PATCH_CACHE[key] = (g.contiguous().to(device), int(tmp_size))
    return _PATCH_CACHE[key]


def rectify(hm, sigma):  # b, c, h, w -> b, c, h, w
    """utils.py:77-109 as ONE kernel: arg-max per (b,c) then the Gaussian patch stamped into a zero map (clipped)."""
    _hip.require_cuda(hm)
    src = hm.detach().float().contiguous()
    B, K, H, W = src.shape
    patch, rad = _gauss_patch(sigma, src.device)
    out = torch.empty_like(src)
    check(lib().udapose_heatmap_argmax(_hip.stream(), ptr(src), B * K, H, W, None, None, None, ptr(out), ptr(patch), rad), "rectify")
    return out.to(hm.dtype)


def heatmap_activations(recon):
    """activates = amax over (h,w) per (b,k)  (train_human.py:427), one sweep."""
    _hip.require_cuda(recon)
    src = recon.detach().float().contiguous()
    B, K, H, W = src.shape
    act = torch.empty(B, K, dtype=torch.float32, device=src.device)
    check(lib().udapose_heatmap_argmax(_hip.stream(), ptr(src), B * K, H, W, ptr(act), None, None, None, None, 0), "amax")
    return act


def activations_and_rectify(recon, sigma):
    """heatmap_activations(recon) and rectify(recon, sigma) from ONE sweep (the arg-max of a (b,k) plane gives both; the mean-teacher step
    needs both from the teacher's re-warped heat-maps, train_human.py:427 and :431).  Returns (activates [B,K], rectified [B,K,H,W])."""
    _hip.require_cuda(recon)
    src = recon.detach().float().contiguous()
    B, K, H, W = src.shape
    patch, rad = _gauss_patch(sigma, src.device)
    act = torch.empty(B, K, dtype=torch.float32, device=src.device)
    out = torch.empty_like(src)
    check(lib().udapose_heatmap_argmax(_hip.stream(), ptr(src), B * K, H, W, ptr(act), None, None, ptr(out), ptr(patch), rad), "amax + rectify")
    return act, out.to(recon.dtype)


def confidence_mask(recon, mask_ratio, tea_mask=None, gathered_activates=None, activates=None):
    """train_human.py:427-430 without the host round trip: activates = amax_{hw}(recon); thr = k-th smallest with
    k = int(mask_ratio * numel); mask = (tea_mask * activates) > thr.  `gathered_activates` (all ranks' activates,
    flattened) makes thr the GLOBAL-batch statistic under data parallelism.  Returns (mask bool [B,K], activates [B,K], thr)."""
    act = heatmap_activations(recon) if activates is None else activates
    B, K = act.shape
    pool = act.reshape(-1) if gathered_activates is None else gathered_activates.detach().float().reshape(-1).contiguous()
    k = int(mask_ratio * pool.numel())
    mask = torch.empty(B, K, dtype=torch.uint8, device=act.device)
    thr = torch.empty((), dtype=torch.float32, device=act.device)
    tm = None if tea_mask is None else tea_mask.detach().float().contiguous()
    check(lib().udapose_kth_mask(_hip.stream(), ptr(pool), ptr(tm), pool.numel(), k, ptr(thr), ptr(mask), ptr(act), B * K), "kth_mask")
    return mask.view(torch.bool), act, thr        # (the kernel writes 0 / 1: a bool view, not a conversion launch)


def split_saturations(reset=True):
    """How many f16x2 ('fp32-grade' mode) stores since the last reset hit a value outside fp16's range (|v| > 65504 saturates; NaN) - the
    teacher / validate() / style-network forwards of the reference's precision mix.  Sums both library builds; synchronises the device."""
    import ctypes as C
    from . import _hip
    total = 0
    for kind in ("bf16", "fp16"):
        if kind in _hip._libs or kind == "bf16":
            c = C.c_ulonglong(0)
            check(_hip.lib(kind).udapose_split_saturations(int(bool(reset)), C.byref(c)), "split_saturations")
            total += int(c.value)
    return total
