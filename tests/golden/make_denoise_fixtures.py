#!/usr/bin/env python
"""Golden vectors of the a-trous denoiser, produced by the REFERENCE's own NumPy implementation
(python/forge3d/denoise.py is pure NumPy and importable in the build container; the native module is
not needed).  Run here only -- /root/reference does not exist on the GPU box; the .npz travels.

    PYTHONPATH=/root/reference/python python tests/golden/make_denoise_fixtures.py
"""
import importlib.util
from pathlib import Path

import numpy as np

spec = importlib.util.spec_from_file_location("ref_denoise", "/root/reference/python/forge3d/denoise.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(20260927)
h, w = 40, 52
yy, xx = np.mgrid[0:h, 0:w]
clean = np.stack([0.5 + 0.4 * np.sin(xx / 7.0), 0.5 + 0.4 * np.cos(yy / 5.0), np.where(xx > w // 2, 0.9, 0.2)], -1)
color = np.clip(clean + rng.normal(0, 0.08, clean.shape), 0, 1.5).astype(np.float32)
albedo = np.stack([np.where(xx > w // 2, 0.8, 0.3), np.where(yy > h // 3, 0.6, 0.5), 0.4 + 0.0 * xx], -1).astype(np.float32)
normal = np.stack([np.sin(xx / 9.0), np.cos(yy / 11.0) * 0.5, 1.0 + 0.0 * xx], -1).astype(np.float32)  # not unit length
normal[3, 4] = 0.0  # a zero normal: the eps branch of the normalisation
depth = (10.0 + 0.05 * xx + np.where(yy > h // 2, 3.0, 0.0) + rng.normal(0, 0.01, (h, w))).astype(np.float32)
cases = {
    "color_only_3": dict(iterations=3),
    "color_only_1_wide": dict(iterations=1, sigma_color=0.35),
    "all_guides_3": dict(albedo=albedo, normal=normal, depth=depth, iterations=3),
    "all_guides_4_tight": dict(albedo=albedo, normal=normal, depth=depth, iterations=4, sigma_color=0.05,
                               sigma_albedo=0.1, sigma_normal=0.1, sigma_depth=0.2),
    "normal_depth_2": dict(normal=normal, depth=depth, iterations=2, sigma_normal=0.1, sigma_depth=0.1),
    "albedo_no_extra_term": dict(albedo=albedo, iterations=2, sigma_albedo=0.0),
    "zero_iterations_means_one": dict(iterations=0),
}
out = {"color": color, "albedo": albedo, "normal": normal, "depth": depth}
for name, kw in cases.items():
    out["want_" + name] = ref.atrous_denoise(color, **kw)
np.savez_compressed(Path(__file__).resolve().parent / "atrous_cases.npz", **out)
print({k: v.shape for k, v in out.items()})
