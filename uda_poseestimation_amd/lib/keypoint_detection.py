"""Arg-max decode and PCK (API mirror of the reference's lib/keypoint_detection.py:9-94) on MI355X kernels.

The reference takes numpy arrays (it is called on `.cpu().numpy()` copies, train_human.py:289,443).  The same calls work
here; torch CUDA tensors are accepted as well and avoid the 2 x 8.4 MB device->host copy per iteration: decode and PCK
run on the device and only K+2 floats and the [B,K,2] coordinates come back.
"""
import numpy as np
import torch

from .. import _hip
from .._hip import check, lib, ptr


def _dev_f32(a):
    if isinstance(a, np.ndarray):
        assert a.ndim == 4, 'batch_images should be 4-ndim'
        if not torch.cuda.is_available():
            raise RuntimeError("uda_poseestimation_amd.lib.keypoint_detection needs the MI355X (no CPU fallback)")
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    assert torch.is_tensor(a) and a.dim() == 4, 'batch_heatmaps should be numpy.ndarray or a 4-d tensor'
    _hip.require_cuda(a)
    return a.detach().float().contiguous()


def _decode(hm):
    B, K, H, W = hm.shape
    preds = torch.empty(B, K, 2, dtype=torch.float32, device=hm.device)
    maxv = torch.empty(B, K, 1, dtype=torch.float32, device=hm.device)
    check(lib().udapose_heatmap_argmax(_hip.stream(), ptr(hm), B * K, H, W, ptr(maxv), None, ptr(preds), None, None, 0), "heatmap_argmax")
    return preds, maxv


def get_max_preds(batch_heatmaps):
    """[B,K,H,W] -> (preds [B,K,2] float32 (x,y), maxvals [B,K,1]); numpy in -> numpy out, tensor in -> tensor out."""
    is_np = isinstance(batch_heatmaps, np.ndarray)
    if not is_np and not torch.is_tensor(batch_heatmaps):
        raise AssertionError('batch_heatmaps should be numpy.ndarray')
    preds, maxv = _decode(_dev_f32(batch_heatmaps))
    if is_np:
        return preds.cpu().numpy(), maxv.cpu().numpy().astype(batch_heatmaps.dtype, copy=False)
    return preds, maxv


def accuracy_device(output, target, thr=0.5):
    """`accuracy` without the read-back: (acc [K] float32 with -1 for key points absent from the batch, [avg_acc, cnt],
    pred [B,K,2]) as CUDA tensors on the current stream - no host synchronisation (validate() accumulates them on the
    device and reads the set's averages back once)."""
    o, t = _dev_f32(output), _dev_f32(target)
    B, K, H, W = o.shape
    pred, _ = _decode(o)
    gt, _ = _decode(t)
    acc = torch.empty(K, dtype=torch.float32, device=o.device)
    avg_cnt = torch.empty(2, dtype=torch.float32, device=o.device)
    check(lib().udapose_pck(_hip.stream(), ptr(pred), ptr(gt), B, K, H / 10.0, W / 10.0, float(thr), ptr(acc), ptr(avg_cnt)), "pck")
    return acc, avg_cnt, pred


def accuracy(output, target, hm_type='gaussian', thr=0.5):
    """PCK@(thr/10 of the heat-map size) from GT heat-maps; returns (acc[K], avg_acc, cnt, pred[B,K,2]) like the reference."""
    if hm_type != 'gaussian':
        raise NotImplementedError("only hm_type='gaussian' is used by the reference scripts")
    is_np = isinstance(output, np.ndarray)
    o, t = _dev_f32(output), _dev_f32(target)
    B, K, H, W = o.shape
    pred, _ = _decode(o)
    gt, _ = _decode(t)
    acc = torch.empty(K, dtype=torch.float32, device=o.device)
    avg_cnt = torch.empty(2, dtype=torch.float32, device=o.device)
    check(lib().udapose_pck(_hip.stream(), ptr(pred), ptr(gt), B, K, H / 10.0, W / 10.0, float(thr), ptr(acc), ptr(avg_cnt)), "pck")
    ac = avg_cnt.cpu()
    acc_np = acc.cpu().numpy().astype(np.float64)
    return acc_np, float(ac[0]), int(ac[1]), (pred.cpu().numpy() if is_np else pred)
