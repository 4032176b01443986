"""Heatmap losses, CPU oracle (test-only).

  * joints_mse_ref  <- lib/models/loss.py:39-49  (0.5*(pred-gt)^2 * w[b,k], mean over all, or [B,K] means)
  * cons_loss_ref   <- lib/models/loss.py:124-132 (masked squared diff, mean over C then over B*H*W)
"""
import torch


def joints_mse_ref(output, target, target_weight=None, reduction="mean"):
    B, K = output.shape[:2]
    d = (output.reshape(B, K, -1).float() - target.reshape(B, K, -1).float()) ** 2 * 0.5
    if target_weight is not None:
        d = d * target_weight.reshape(B, K, 1)
    if reduction == "mean":
        return d.mean()
    if reduction == "none":
        return d.mean(dim=-1)
    return None  # loss.py:46-49 falls through silently for other strings


def cons_loss_ref(stu_out, tea_out, valid_mask=None, tea_mask=None):
    diff = stu_out - tea_out
    if tea_mask is not None:
        diff = diff * tea_mask[:, :, None, None]
    loss_map = (diff ** 2).mean(dim=1)
    if valid_mask is not None:
        loss_map = loss_map[valid_mask]
    return loss_map.mean()
