"""Image metrics used by the parity gates.

SSIM follows the definition the reference's golden gate uses (reference
tests/_ssim.py:11-83): 11x11 Gaussian window (sigma 1.5, normalised), K1 = 0.01,
K2 = 0.03, zero-padded "same" filtering, mean of the SSIM map, channels averaged.
Implemented here as a separable filter (the Gaussian window is an outer product).
"""
from __future__ import annotations

import numpy as np


def _gauss1d(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-0.5 * (x / sigma) ** 2)
    return g / g.sum()


def _blur(img: np.ndarray, g: np.ndarray) -> np.ndarray:
    # zero-padded separable correlation (the window is symmetric, so == convolution)
    r = len(g) // 2
    p = np.pad(img, ((r, r), (r, r)), mode="constant")
    h, w = img.shape
    tmp = np.zeros((h + 2 * r, w), np.float64)
    for k, gk in enumerate(g):
        tmp += gk * p[:, k:k + w]
    out = np.zeros((h, w), np.float64)
    for k, gk in enumerate(g):
        out += gk * tmp[k:k + h, :]
    return out


def ssim(a: np.ndarray, b: np.ndarray, data_range: float = 255.0) -> float:
    if a.shape != b.shape:
        raise ValueError(f"shape mismatch {a.shape} vs {b.shape}")
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    if a.ndim == 3:
        return float(np.mean([ssim(a[..., c], b[..., c], data_range) for c in range(a.shape[2])]))
    g = _gauss1d()
    c1 = (0.01 * data_range) ** 2
    c2 = (0.03 * data_range) ** 2
    mu_a, mu_b = _blur(a, g), _blur(b, g)
    va = _blur(a * a, g) - mu_a * mu_a
    vb = _blur(b * b, g) - mu_b * mu_b
    cov = _blur(a * b, g) - mu_a * mu_b
    s = ((2 * mu_a * mu_b + c1) * (2 * cov + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (va + vb + c2))
    return float(s.mean())


def mean_abs(a: np.ndarray, b: np.ndarray) -> float:
    return float(np.mean(np.abs(a.astype(np.float32) - b.astype(np.float32))))
