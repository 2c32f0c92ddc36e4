"""Losses of the reference's two training loops."""
from __future__ import annotations

import torch


def charbonnier_loss(x: torch.Tensor, y: torch.Tensor, mean_res: bool = False, eps: float = 1e-4) -> torch.Tensor:
    """sum(sqrt((x - y)^2 + eps)), eps = 1e-4 (reference CVSR_train/opt/loss.py:20-31; `mean_res` averages the difference
    per sample first).  A SUM over all elements: gradients scale with the batch, so data-parallel ranks all-reduce with SUM."""
    diff = x - y
    if mean_res:
        diff = diff.reshape(x.shape[0], -1).mean(1, keepdim=True)
    return torch.sum(torch.sqrt(diff * diff + eps))


def charbonnier_loss_mmedit(pred: torch.Tensor, target: torch.Tensor, loss_weight: float = 1.0, eps: float = 1e-12,
                            reduction: str = "mean") -> torch.Tensor:
    """mmedit's CharbonnierLoss (reference mmedit_train/mmedit/models/losses/pixelwise_loss.py:41-51, :120-160):
    sqrt((pred - target)^2 + eps), eps = 1e-12, mean by default (data-parallel ranks average)."""
    v = torch.sqrt((pred - target) ** 2 + eps)
    if reduction == "mean":
        v = v.mean()
    elif reduction == "sum":
        v = v.sum()
    elif reduction != "none":
        raise ValueError(f"Unsupported reduction mode: {reduction}. Supported ones are: ['none', 'mean', 'sum']")
    return loss_weight * v
