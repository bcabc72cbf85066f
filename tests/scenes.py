"""Shared scene definitions for the parity tests."""
from __future__ import annotations

from pathlib import Path

import numpy as np

GOLDEN_DIR = Path(__file__).resolve().parent / "golden"

# Locked golden scene of the reference (tests/test_hybrid_terrain_pt.py:30-76)
SIZE = 256
SPAN = 100.0
RELIEF = 20.0
CAM = {"origin": (0.0, 35.0, 90.0), "look_at": (0.0, 5.0, 0.0), "up": (0.0, 1.0, 0.0), "fov_y": 45.0,
       "exposure": 1.0}
ALBEDO = (0.55, 0.52, 0.48)


def mini_dem() -> np.ndarray:
    return np.load(GOLDEN_DIR / "mini_dem.npy").astype(np.float32)


def golden_dem(step: int = 2) -> np.ndarray:
    dem = mini_dem()[::step, ::step].astype(np.float32)
    dem -= dem.min()
    dem /= max(float(dem.max()), 1e-6)
    return dem


def scene_kwargs(dem) -> dict:
    spacing = SPAN / (dem.shape[1] - 1)
    return dict(spacing=(spacing, spacing), exaggeration=RELIEF, albedo=ALBEDO, sun_azimuth_deg=225.0,
                sun_elevation_deg=35.0, sun_intensity=2.5, env_intensity=0.35, max_frames=512, min_frames=32,
                variance_threshold=1e-3, seed=7)


def fixed_frames(kw: dict, frames: int, **extra) -> dict:
    return {**kw, "max_frames": frames, "min_frames": frames, "variance_threshold": 1e30, **extra}


def golden_png() -> np.ndarray:
    from PIL import Image

    return np.array(Image.open(GOLDEN_DIR / "mini_dem_reference.png"))


def curvature_fixture() -> np.ndarray:
    """256x256 proof DEM of the reference's traversal KATs
    (src/path_tracing/hybrid_compute/terrain_heightfield.rs:618-629), evaluated in f32."""
    i = np.arange(256 * 256)
    x = (i % 256).astype(np.float32)
    y = (i // 256).astype(np.float32)
    f = np.float32
    h = (f(900.0) + f(180.0) * np.sin(x * f(0.071), dtype=np.float32)
         + f(120.0) * np.cos(y * f(0.047), dtype=np.float32)
         + f(650.0) * np.exp(-((x - f(150.0)) ** 2 + (y - f(126.0)) ** 2) / f(900.0), dtype=np.float32))
    return h.astype(np.float32).reshape(256, 256)


def xorshift_stream(state: int):
    """next_u32 of the reference's KAT ray generator (terrain_heightfield.rs:875-880)."""
    while True:
        state ^= (state << 13) & 0xFFFFFFFF
        state ^= state >> 17
        state ^= (state << 5) & 0xFFFFFFFF
        yield state


def proof_rays(n_random: int = 10_000, mask: bool = True):
    """Ray set of `curvature_descent_is_conservative` (terrain_heightfield.rs:2081-2116):
    n_random xorshift rays (seed 0x48454c49) + the 255x255 grazing shadow mask, as (n,8) f32
    rows (origin, tmin=1e-3, direction, tmax=200000), spacing 500 m, origin (0,0)."""
    f = np.float32
    heights = curvature_fixture()
    spacing = f(500.0)

    def surface(x, z):
        cx, cz = int(np.floor(x)), int(np.floor(z))
        ox, oz = f(x) * spacing, f(z) * spacing
        u = float(ox) / 500.0 - cx
        v = float(oz) / 500.0 - cz
        h00, h10 = float(heights[cz, cx]), float(heights[cz, cx + 1])
        h01, h11 = float(heights[cz + 1, cx]), float(heights[cz + 1, cx + 1])
        return ox, oz, f((h00 * (1 - u) + h10 * u) * (1 - v) + (h01 * (1 - u) + h11 * u) * v)

    def ray(x, z, az, el):
        ox, oz, s = surface(x, z)
        hz = np.cos(f(el), dtype=np.float32)
        return [ox, s + f(1.7), oz, f(1e-3), hz * np.cos(f(az), dtype=np.float32), np.sin(f(el), dtype=np.float32),
                hz * np.sin(f(az), dtype=np.float32), f(200000.0)]

    rays = []
    gen = xorshift_stream(0x48454C49)
    umax = f(4294967295.0)
    for _ in range(n_random):
        x = f(1.0) + f(next(gen) % 253) + (f(next(gen)) / umax) * f(0.999)
        z = f(1.0) + f(next(gen) % 253) + (f(next(gen)) / umax) * f(0.999)
        az = (f(next(gen)) / umax) * f(6.283185307179586)
        el = np.deg2rad(f(0.1) + f(next(gen) % 790) / f(100.0)).astype(np.float32)
        rays.append(ray(x, z, az, el))
    if mask:
        az, el = np.deg2rad(f(37.0)).astype(np.float32), np.deg2rad(f(0.6)).astype(np.float32)
        for z in range(255):
            for x in range(255):
                rays.append(ray(f(x) + f(0.5), f(z) + f(0.5), az, el))
    return heights, np.asarray(rays, dtype=np.float32)
