"""Host-side quality metrics of the reference's evaluation harness (CPU, numpy)."""
from __future__ import annotations

import numpy as np


def psnr(img1: np.ndarray, img2: np.ndarray, crop_border: int = 4) -> float:
    """PSNR on [0,255] images: 20*log10(255/sqrt(mse)), borders cropped first
    (reference CVSR_train/metric/psnr_ssim.py:278-317)."""
    a = np.asarray(img1, dtype=np.float64)
    b = np.asarray(img2, dtype=np.float64)
    if a.shape != b.shape:
        raise ValueError(f"Image shapes are different: {a.shape}, {b.shape}.")
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border, ...]
        b = b[crop_border:-crop_border, crop_border:-crop_border, ...]
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float("inf")
    return float(20.0 * np.log10(255.0 / np.sqrt(mse)))
