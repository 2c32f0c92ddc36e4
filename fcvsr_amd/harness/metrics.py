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


def _gaussian_window(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    """cv2.getGaussianKernel(size, sigma): exp(-(i - (size-1)/2)^2 / (2 sigma^2)), normalised to sum 1."""
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


def _filter_valid(img: np.ndarray, g: np.ndarray) -> np.ndarray:
    """Separable 'valid' correlation with the (symmetric) window outer(g, g): what cv2.filter2D(...)[5:-5, 5:-5] keeps."""
    k = g.size
    H, W = img.shape
    rows = np.zeros((H - k + 1, W), dtype=np.float64)
    for i in range(k):
        rows += g[i] * img[i:i + H - k + 1, :]
    out = np.zeros((H - k + 1, W - k + 1), dtype=np.float64)
    for j in range(k):
        out += g[j] * rows[:, j:j + W - k + 1]
    return out


def _ssim_plane(a: np.ndarray, b: np.ndarray) -> float:
    """SSIM of one channel on [0,255] (reference CVSR_train/metric/psnr_ssim.py:320-350: 11x11 Gaussian, sigma 1.5,
    C1 = (0.01*255)^2, C2 = (0.03*255)^2, border of 5 pixels dropped, mean of the map)."""
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    g = _gaussian_window()
    mu1, mu2 = _filter_valid(a, g), _filter_valid(b, g)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _filter_valid(a * a, g) - mu1_sq
    s2 = _filter_valid(b * b, g) - mu2_sq
    s12 = _filter_valid(a * b, g) - mu12
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return float(m.mean())


def to_y_channel(img_hwc: np.ndarray) -> np.ndarray:
    """BGR [0,255] -> Y of YCbCr in [16,235] as float (mmedit/core/evaluation/metrics.py to_y_channel via mmcv.bgr2ycbcr):
    Y = (24.966 B + 128.553 G + 65.481 R) / 255 + 16."""
    img = np.asarray(img_hwc, dtype=np.float64) / 255.0
    return img @ np.array([24.966, 128.553, 65.481]) + 16.0


def ssim(img1: np.ndarray, img2: np.ndarray, crop_border: int = 4, input_order: str = "HWC", convert_to=None) -> float:
    """Structural similarity on [0,255] images; channels averaged (reference CVSR_train/metric/psnr_ssim.py:353-398 for the
    single-plane Y images of the CVSR harness; mmedit/core/evaluation/metrics.py ssim for HWC / CHW and convert_to='Y').
    Known answers: reference tests/test_metrics/test_metrics.py:79-106 (0.9130623, and 0.9987801 on Y)."""
    a, b = np.asarray(img1), np.asarray(img2)
    if a.shape != b.shape:
        raise ValueError(f"Image shapes are different: {a.shape}, {b.shape}.")
    if input_order not in ("HWC", "CHW"):
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')
    if a.ndim == 2:
        a, b = a[..., None], b[..., None]
    elif input_order == "CHW":
        a, b = a.transpose(1, 2, 0), b.transpose(1, 2, 0)
    if isinstance(convert_to, str) and convert_to.lower() == "y":
        a, b = to_y_channel(a)[..., None], to_y_channel(b)[..., None]
    elif convert_to is not None:
        raise ValueError('Wrong color model. Supported values are "Y" and None')
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border, :]
        b = b[crop_border:-crop_border, crop_border:-crop_border, :]
    return float(np.mean([_ssim_plane(a[..., c], b[..., c]) for c in range(a.shape[2])]))
