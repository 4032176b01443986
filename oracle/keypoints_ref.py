"""Arg-max decode and PCK, CPU oracle (test-only).

  * get_max_preds_ref <- lib/keypoint_detection.py:9-37 (first flat arg-max; x = idx % W, y = idx // W; zero if max<=0)
  * accuracy_ref      <- lib/keypoint_detection.py:40-94 (PCK: GT needs x>1 and y>1; dist / ([h,w]/10) < thr)
  * get_max_preds_torch_ref <- utils.py:54-75
"""
import numpy as np
import torch


def get_max_preds_ref(hm):
    assert isinstance(hm, np.ndarray) and hm.ndim == 4
    B, K, H, W = hm.shape
    flat = hm.reshape(B, K, -1)
    idx = flat.argmax(2)
    maxvals = flat.max(2)[..., None]
    preds = np.stack([idx % W, idx // W], -1).astype(np.float32)
    preds *= (maxvals > 0.0).astype(np.float32)
    return preds, maxvals


def accuracy_ref(output, target, thr=0.5):
    pred, _ = get_max_preds_ref(output)
    gt, _ = get_max_preds_ref(target)
    B, K = pred.shape[:2]
    h, w = output.shape[2:]
    norm = np.array([h, w], dtype=np.float64) / 10
    acc = np.zeros(K)
    tot, cnt = 0.0, 0
    for c in range(K):
        hits, n = 0, 0
        for b in range(B):
            if gt[b, c, 0] > 1 and gt[b, c, 1] > 1:
                d = np.linalg.norm(pred[b, c] / norm - gt[b, c] / norm)
                n += 1
                hits += d < thr
        acc[c] = hits / n if n else -1
        if acc[c] >= 0:
            tot += acc[c]
            cnt += 1
    return acc, (tot / cnt if cnt else 0), cnt, pred


def get_max_preds_torch_ref(hm):
    B, K, H, W = hm.shape
    flat = hm.reshape(B, K, -1)
    idx = flat.argmax(2)
    maxvals = flat.amax(2).reshape(B, K, 1)
    preds = torch.stack([(idx % W).float(), torch.floor(idx.float() / W)], -1)
    preds = preds * (maxvals > 0.0).float()
    return preds, maxvals
