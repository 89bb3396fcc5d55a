"""SSIM / NMSE / PSNR as the reference's utils/evaluate.py defines them, without scikit-image.

Two forms: the reference's numpy interface (``mse / nmse / psnr / ssim`` on host arrays, used by mri_module.py:201-209 and
by the tests as the independent check) and ``metrics_device`` / ``ssim_device``: the fused HIP kernel (cine_image_metrics)
on GPU tensors -- center crop, 7x7 window moments, SSIM map, frame means, NMSE and PSNR without leaving the device.

``structural_similarity(gt, pred, data_range=maxval)`` with skimage's defaults is a 7x7
uniform window, sample covariance (N/(N-1)), K1 = 0.01, K2 = 0.03, mean over the window-valid
interior; ``ssim`` averages it over frames (reference evaluate.py:25-42).  Host side (numpy).
"""
from typing import Optional

import numpy as np
from scipy.ndimage import uniform_filter


def mse(gt: np.ndarray, pred: np.ndarray) -> np.ndarray:
    return np.mean((gt - pred) ** 2)


def nmse(gt: np.ndarray, pred: np.ndarray) -> np.ndarray:
    return np.linalg.norm(gt - pred) ** 2 / np.linalg.norm(gt) ** 2


def psnr(gt: np.ndarray, pred: np.ndarray, maxval: Optional[float] = None) -> np.ndarray:
    maxval = gt.max() if maxval is None else maxval
    return 10 * np.log10((maxval ** 2) / np.mean((np.asarray(gt, np.float64) - np.asarray(pred, np.float64)) ** 2))


def _ssim2d(x: np.ndarray, y: np.ndarray, data_range: float, win: int = 7, k1: float = 0.01, k2: float = 0.03) -> float:
    x = x.astype(np.float64); y = y.astype(np.float64)
    npix = win * win
    cov_norm = npix / (npix - 1)
    ux, uy = uniform_filter(x, win), uniform_filter(y, win)
    uxx, uyy, uxy = uniform_filter(x * x, win), uniform_filter(y * y, win), uniform_filter(x * y, win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


def ssim(gt: np.ndarray, pred: np.ndarray, maxval: Optional[float] = None) -> float:
    if not gt.ndim == 3:
        raise ValueError("Unexpected number of dimensions in ground truth.")
    if not gt.ndim == pred.ndim:
        raise ValueError("Ground truth dimensions does not match pred.")
    maxval = gt.max() if maxval is None else maxval
    return sum(_ssim2d(gt[i], pred[i], maxval) for i in range(gt.shape[0])) / gt.shape[0]


def metrics_device(gt, pred, maxval: Optional[float] = None) -> dict:
    """All four metrics of (t, h, w) GPU tensors in one pass; crops both to the smaller size first
    (data/transforms.py:161-183).  Values are 0-d float64 device tensors (no host sync)."""
    from cine_hip import ops
    return ops.image_metrics(gt, pred, maxval=maxval)


def ssim_device(gt, pred, maxval: Optional[float] = None):
    return metrics_device(gt, pred, maxval)["ssim"]


METRIC_FUNCS = dict(MSE=mse, NMSE=nmse, PSNR=psnr, SSIM=ssim)
