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


# ---- colour difference (the AETHER acceptance gate, reference tests/_deltae.py: sRGB D65 -> CIE L*a*b* -> CIEDE2000) ----
def srgb_to_linear(rgb):
    c = np.asarray(rgb, np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def srgb_to_lab(rgb):
    """sRGB in [0, 1] -> L*a*b* (D65 white 0.95047, 1.0, 1.08883)."""
    lin = srgb_to_linear(rgb)
    m = np.array([[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]])
    xyz = lin @ m.T / np.array([0.95047, 1.0, 1.08883])
    eps, kappa = 216.0 / 24389.0, 24389.0 / 27.0
    f = np.where(xyz > eps, np.cbrt(xyz), (kappa * xyz + 16.0) / 116.0)
    return np.stack([116.0 * f[..., 1] - 16.0, 500.0 * (f[..., 0] - f[..., 1]), 200.0 * (f[..., 1] - f[..., 2])], axis=-1)


def delta_e_2000(lab1, lab2):
    """CIEDE2000 (Sharma, Wu, Dalal 2005), kL = kC = kH = 1."""
    l1, a1, b1 = np.moveaxis(np.asarray(lab1, np.float64), -1, 0)
    l2, a2, b2 = np.moveaxis(np.asarray(lab2, np.float64), -1, 0)
    c_bar = 0.5 * (np.hypot(a1, b1) + np.hypot(a2, b2))
    g = 0.5 * (1.0 - np.sqrt(c_bar ** 7 / (c_bar ** 7 + 25.0 ** 7)))
    a1p, a2p = (1.0 + g) * a1, (1.0 + g) * a2
    c1p, c2p = np.hypot(a1p, b1), np.hypot(a2p, b2)
    h1p = np.degrees(np.arctan2(b1, a1p)) % 360.0
    h2p = np.degrees(np.arctan2(b2, a2p)) % 360.0
    dl, dc = l2 - l1, c2p - c1p
    dh = h2p - h1p
    dh = np.where(dh > 180.0, dh - 360.0, np.where(dh < -180.0, dh + 360.0, dh))
    dh = np.where(c1p * c2p == 0.0, 0.0, dh)
    dhp = 2.0 * np.sqrt(c1p * c2p) * np.sin(np.radians(0.5 * dh))
    l_bar, cp_bar = 0.5 * (l1 + l2), 0.5 * (c1p + c2p)
    h_sum, h_diff = h1p + h2p, np.abs(h1p - h2p)
    h_bar = np.where(h_diff <= 180.0, 0.5 * h_sum, np.where(h_sum < 360.0, 0.5 * (h_sum + 360.0), 0.5 * (h_sum - 360.0)))
    h_bar = np.where(c1p * c2p == 0.0, h_sum, h_bar)
    t = (1.0 - 0.17 * np.cos(np.radians(h_bar - 30.0)) + 0.24 * np.cos(np.radians(2.0 * h_bar)) + 0.32 * np.cos(np.radians(3.0 * h_bar + 6.0))
         - 0.20 * np.cos(np.radians(4.0 * h_bar - 63.0)))
    sl = 1.0 + 0.015 * (l_bar - 50.0) ** 2 / np.sqrt(20.0 + (l_bar - 50.0) ** 2)
    sc, sh = 1.0 + 0.045 * cp_bar, 1.0 + 0.015 * cp_bar * t
    rc = 2.0 * np.sqrt(cp_bar ** 7 / (cp_bar ** 7 + 25.0 ** 7))
    rt = -np.sin(np.radians(60.0 * np.exp(-(((h_bar - 275.0) / 25.0) ** 2)))) * rc
    return np.sqrt((dl / sl) ** 2 + (dc / sc) ** 2 + (dhp / sh) ** 2 + rt * (dc / sc) * (dhp / sh))


def filmic_terrain_srgb(linear_rgb, exposure: float = 1.0):
    """The display transform of the reference's acceptance tests (tests/_aether_quadrature.py:33-53, mirroring
    tonemap_filmic_terrain + linear_to_srgb of its tonemap_common.wgsl): Hable curve normalised at white 11.2, then sRGB."""
    x = np.maximum(np.asarray(linear_rgb, np.float64) * float(exposure), 0.0)
    a, b, c, d, e, f, white = 0.22, 0.30, 0.10, 0.20, 0.01, 0.30, 11.2

    def curve(v):
        v = np.asarray(v, np.float64)
        return (v * (a * v + c * b) + d * e) / (v * (a * v + b) + d * f) - e / f

    lin = np.clip(curve(x) / max(float(curve(white)), 1.0e-6), 0.0, 1.0)
    return np.clip(np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(lin, 1.0 / 2.4) - 0.055), 0.0, 1.0)
