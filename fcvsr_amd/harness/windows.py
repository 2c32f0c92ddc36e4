"""Sliding-window frame indices of the reference's inference harnesses (host logic)."""
from __future__ import annotations

from typing import List


def window_indices(center: int, num_frames: int, seq_len: int, padding: str = "replicate") -> List[int]:
    """Indices of the ``num_frames`` LR frames centred on ``center`` in a sequence of ``seq_len`` frames.

    ``replicate``: clip to [0, seq_len-1]  (behaviour of reference CVSR_train/test_LD_freqCVSR_S_22.py:13-16).
    ``reflection`` / ``reflection_circle`` / ``circle``: behaviour of mmedit's GenerateFrameIndiceswithPadding
    (reference mmedit_train/mmedit/datasets/pipelines/augmentation.py:808-883) used by the Vid4 / REDS configs.
    """
    half, last = num_frames // 2, seq_len - 1
    lo, hi = center - half, center + half
    below = {"replicate": lambda i: 0, "reflection": lambda i: -i, "reflection_circle": lambda i: hi - i,
             "circle": lambda i: i + num_frames}
    above = {"replicate": lambda i: last, "reflection": lambda i: 2 * last - i,
             "reflection_circle": lambda i: lo - (i - last), "circle": lambda i: i - num_frames}
    if padding not in below:
        raise ValueError(f"unknown padding mode {padding!r}")
    return [i if 0 <= i <= last else (below[padding](i) if i < 0 else above[padding](i)) for i in range(lo, hi + 1)]
