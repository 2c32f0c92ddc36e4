"""Sequence inference harness: the counterpart of the reference's evaluation loop
(CVSR_train/test_LD_freqCVSR_S_22.py:48-123) around the drop-in model.

For every output frame i of a sequence: take the 7 LR frames window_indices(i) (edge replicate by default), pad rows to a
multiple of 4 the way the reference pads 270 -> 272 (zero rows appended at the bottom, :24-26), run the model on batches
of windows, crop the padding off the SR frame (:81-86), clamp to [0,1], scale by 255 and TRUNCATE to uint8 (:88-89;
mmedit's tensor2img rounds instead - `quantise="round"`).  Frames are independent, so windows are batched and, across
GPUs, sharded by `fcvsr_amd.harness.sharding`.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import numpy as np
import torch

from .metrics import psnr
from .windows import window_indices


def pad_to_multiple(frames: torch.Tensor, mult: int = 4) -> torch.Tensor:
    """(N,C,H,W) -> zero-padded at the bottom/right so that H, W are multiples of `mult`."""
    H, W = frames.shape[-2:]
    ph, pw = (-H) % mult, (-W) % mult
    if ph == 0 and pw == 0:
        return frames
    return torch.nn.functional.pad(frames, (0, pw, 0, ph))


@torch.no_grad()
def super_resolve_sequence(model, lr: torch.Tensor, *, num_frames: int = 7, padding: str = "replicate", batch: int = 8,
                           centres: Optional[Iterable[int]] = None, quantise: str = "truncate") -> np.ndarray:
    """lr: (N,C,H,W) float in [0,1] (host or device).  Returns uint8 (len(centres),C,4H,4W) SR frames."""
    N, C, H, W = lr.shape
    dev = next(model.parameters()).device
    x = pad_to_multiple(lr.float(), 4).to(dev)
    centres = list(range(N)) if centres is None else list(centres)
    out: List[np.ndarray] = []
    for s in range(0, len(centres), batch):
        idx = [window_indices(i, num_frames, N, padding) for i in centres[s:s + batch]]
        win = torch.stack([x[j] for j in idx], 0)                 # (b, 7, C, Hp, Wp)
        sr = model(win)[:, :, :4 * H, :4 * W]
        sr = sr.clamp(0, 1) * 255.0
        sr = sr.round() if quantise == "round" else sr            # uint8 cast truncates
        out.append(sr.to(torch.uint8).cpu().numpy())
    return np.concatenate(out, 0)


def sequence_psnr(sr_u8: np.ndarray, hr_u8: np.ndarray, crop_border: int = 4) -> float:
    """Mean per-frame PSNR over a sequence, first channel, borders cropped (reference metric/psnr_ssim.py:447-485)."""
    vals = [psnr(a[0], b[0], crop_border) for a, b in zip(sr_u8, hr_u8)]
    return float(np.mean(vals))
