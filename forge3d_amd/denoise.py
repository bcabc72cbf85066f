"""Edge-aware a-trous denoiser on the MI355X -- drop-in for ``forge3d.denoise.atrous_denoise``
(reference python/forge3d/denoise.py:18-127): same signature, validation messages and result
(float32, same shape), computed by ``f3d_atrous_denoise`` in libf3dhip.so.  A post filter over a
finished render, guided by the albedo / normal / depth AOVs ``hybrid_render_terrain_reference`` returns.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _native


def atrous_denoise(color: np.ndarray, *, albedo: Optional[np.ndarray] = None, normal: Optional[np.ndarray] = None,
                   depth: Optional[np.ndarray] = None, iterations: int = 3, sigma_color: float = 0.1,
                   sigma_albedo: float = 0.2, sigma_normal: float = 0.3, sigma_depth: float = 0.5) -> np.ndarray:
    color = np.asarray(color)
    if color.ndim != 3 or color.shape[2] != 3:
        raise ValueError("color must be (H, W, 3)")
    h, w, _ = color.shape
    if albedo is not None and np.shape(albedo) != (h, w, 3):
        raise ValueError("albedo must match color shape (H, W, 3)")
    if normal is not None and np.shape(normal) != (h, w, 3):
        raise ValueError("normal must match color shape (H, W, 3)")
    if depth is not None and np.shape(depth) != (h, w):
        raise ValueError("depth must be (H, W)")
    if h == 0 or w == 0:
        return color.astype(np.float32, copy=True)

    def f32(x):
        return None if x is None else np.ascontiguousarray(x, np.float32)

    c, a, n, d = f32(color), f32(albedo), f32(normal), f32(depth)
    out = np.empty((h, w, 3), np.float32)
    err = C.create_string_buffer(512)
    ptr = lambda x: None if x is None else x.ctypes.data  # noqa: E731
    rc = _native.lib().f3d_atrous_denoise(ptr(c), ptr(a), ptr(n), ptr(d), w, h, int(iterations), float(sigma_color),
                                          float(sigma_albedo), float(sigma_normal), float(sigma_depth), out.ctypes.data,
                                          err, len(err))
    if rc != 0:
        _native.raise_status(rc, err.value.decode("utf-8", "replace"))
    return out
