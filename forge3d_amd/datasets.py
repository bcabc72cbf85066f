"""Synthetic stand-ins for the DEMs BASELINE.json names.

The reference ships Rainier / Shasta / ... as git-LFS objects fetched over the network
(reference python/forge3d/datasets.py:53-61, 287-300); none of that is reachable here, so
the benchmark configuration runs on a procedurally generated proxy with pinned seeds
(input "S2").  Everything produced here is labelled synthetic.
"""
from __future__ import annotations

import math

import numpy as np

RAINIER_PROXY_SEED = 20260926


def _lattice(ix: np.ndarray, iy: np.ndarray, seed: int) -> np.ndarray:
    """Integer lattice hash -> uniform [0, 1) (32-bit avalanche mix)."""
    h = (ix.astype(np.uint64) * np.uint64(0x9E3779B1) + iy.astype(np.uint64) * np.uint64(0x85EBCA77)
         + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return h.astype(np.float64) / 4294967296.0


def _value_noise(u: np.ndarray, v: np.ndarray, seed: int) -> np.ndarray:
    x0, y0 = np.floor(u), np.floor(v)
    fx, fy = u - x0, v - y0
    sx, sy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
    ix, iy = x0.astype(np.int64), y0.astype(np.int64)
    a = _lattice(ix, iy, seed)
    b = _lattice(ix + 1, iy, seed)
    c = _lattice(ix, iy + 1, seed)
    d = _lattice(ix + 1, iy + 1, seed)
    return (a * (1 - sx) + b * sx) * (1 - sy) + (c * (1 - sx) + d * sx) * sy


def rainier_proxy(size: int = 2048, seed: int = RAINIER_PROXY_SEED) -> np.ndarray:
    """"rainier-proxy": a 4392 m cone + 6-octave value-noise fBm (600 m), float32 metres."""
    t = (np.arange(size, dtype=np.float64) + 0.5) / size
    u, v = np.meshgrid(t, t)
    r = np.hypot(u - 0.5, v - 0.5)
    cone = 4392.0 * np.maximum(0.0, 1.0 - r / 0.45) ** 1.3
    fbm = np.zeros_like(cone)
    amp, freq, norm = 1.0, 8.0, 0.0
    for octave in range(6):
        fbm += amp * (_value_noise(u * freq, v * freq, seed + 101 * octave) - 0.5)
        norm += amp
        amp *= 0.5
        freq *= 2.0
    dem = cone + 600.0 * 2.0 * fbm / norm
    dem -= dem.min()
    return dem.astype(np.float32)


def orbit_camera(target, radius: float, phi_deg: float, theta_deg: float, fov_y_deg: float, exposure: float = 1.0):
    """Orbit-camera mapping of the reference viewer (src/viewer/terrain/scene.rs:110-139):
    eye = target + r (sin(theta) cos(phi), cos(theta), sin(theta) sin(phi))."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    eye = (target[0] + radius * math.sin(th) * math.cos(ph), target[1] + radius * math.cos(th),
           target[2] + radius * math.sin(th) * math.sin(ph))
    return {"origin": eye, "look_at": tuple(target), "up": (0.0, 1.0, 0.0), "fov_y": fov_y_deg,
            "exposure": exposure}


def rainier_proxy_scene(size: int = 2048):
    """BASELINE.json config 2 on the proxy DEM: returns (dem, camera, kwargs)."""
    dem = rainier_proxy(size)
    spacing = 10.0 * 2048.0 / size
    span = (size - 1) * spacing
    target = (0.0, 0.5 * float(dem.max()), 0.0)
    cam = orbit_camera(target, 1.25 * span, 28.0, 49.0, 42.0)
    kw = dict(spacing=(spacing, spacing), exaggeration=1.0, albedo=(0.6, 0.6, 0.6), sun_azimuth_deg=302.0,
              sun_elevation_deg=24.0, sun_intensity=2.5, env_intensity=0.35, seed=7)
    return dem, cam, kw


def mini_dem_scene(dem_256: np.ndarray):
    """The reference's locked golden scene (tests/test_hybrid_terrain_pt.py:30-76) from the
    shipped 256x256 mini DEM: returns (dem, camera, kwargs)."""
    dem = np.asarray(dem_256)[::2, ::2].astype(np.float32)
    dem -= dem.min()
    dem /= max(float(dem.max()), 1e-6)
    spacing = 100.0 / (dem.shape[1] - 1)
    cam = {"origin": (0.0, 35.0, 90.0), "look_at": (0.0, 5.0, 0.0), "up": (0.0, 1.0, 0.0), "fov_y": 45.0,
           "exposure": 1.0}
    kw = dict(spacing=(spacing, spacing), exaggeration=20.0, albedo=(0.55, 0.52, 0.48), sun_azimuth_deg=225.0,
              sun_elevation_deg=35.0, sun_intensity=2.5, env_intensity=0.35, max_frames=512, min_frames=32,
              variance_threshold=1e-3, seed=7)
    return dem, cam, kw


def proxy_buildings(dem: np.ndarray, spacing: float, n_boxes: int = 50_000, seed: int = 7):
    """BASELINE.md input "S4": procedurally extruded boxes (12 triangles each) standing on the DEM,
    a stand-in for the Lyon CityGML LOD2 buildings of BASELINE.json config 4.  Terrain is centred on
    the world origin, y up, DEM row = +z.  Returns (vertices (8n, 3) f32, indices (12n, 3) u32)."""
    rng = np.random.default_rng(seed)
    h, w = dem.shape
    ox, oz = -0.5 * (w - 1) * spacing, -0.5 * (h - 1) * spacing
    ci = rng.integers(w // 8, w - w // 8, n_boxes)
    cj = rng.integers(h // 8, h - h // 8, n_boxes)
    half_w = rng.uniform(4.0, 18.0, n_boxes)
    half_d = rng.uniform(4.0, 18.0, n_boxes)
    height = rng.uniform(6.0, 60.0, n_boxes)
    ground = dem[cj, ci].astype(np.float64)
    cx, cz = ox + ci * spacing, oz + cj * spacing
    corners = np.array([(-1, -1), (1, -1), (1, 1), (-1, 1)], np.float64)
    verts = np.empty((n_boxes, 8, 3), np.float64)
    for k, (sx, sz) in enumerate(corners):
        for lvl, y in enumerate((ground - 3.0, ground + height)):
            verts[:, 4 * lvl + k, 0] = cx + sx * half_w
            verts[:, 4 * lvl + k, 1] = y
            verts[:, 4 * lvl + k, 2] = cz + sz * half_d
    quads = ((0, 1, 2, 3), (7, 6, 5, 4), (0, 4, 5, 1), (1, 5, 6, 2), (2, 6, 7, 3), (3, 7, 4, 0))
    local = np.array([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))], np.uint32)
    idx = (local[None, :, :] + (8 * np.arange(n_boxes, dtype=np.uint32))[:, None, None]).reshape(-1, 3)
    return verts.reshape(-1, 3).astype(np.float32), idx.astype(np.uint32)


# ---- real DEMs, when somebody supplies them ---------------------------------------------------------------------------
# The reference's sample DEMs are git-LFS / remote objects (python/forge3d/datasets.py:53-61: `rainier` ->
# assets/tif/dem_rainier.tif, sha256 875b2434...); this package never downloads.  fetch_dem looks where the reference's
# _local_dataset_path looks -- $FORGE3D_REPO_ROOT/assets/tif/ -- and says so when the file is not there.
_LOCAL_DEMS = {"rainier": "assets/tif/dem_rainier.tif", "fuji": "assets/tif/Mount_Fuji_30m.tif"}


def fetch_dem(name: str, cache_dir=None, base_url=None):
    """Path of a sample DEM of the reference (python/forge3d/datasets.py:312 fetch_dem): only from a local checkout
    ($FORGE3D_REPO_ROOT) -- there is no download here."""
    import os
    from pathlib import Path

    if name not in _LOCAL_DEMS:
        raise KeyError(f"Unknown dataset '{name}'. Available datasets: {', '.join(sorted(_LOCAL_DEMS))}")
    roots = [os.environ.get("FORGE3D_REPO_ROOT"), cache_dir]
    for root in roots:
        if root and (Path(root) / _LOCAL_DEMS[name]).is_file():
            return Path(root) / _LOCAL_DEMS[name]
        if root and (Path(root) / Path(_LOCAL_DEMS[name]).name).is_file():
            return Path(root) / Path(_LOCAL_DEMS[name]).name
    raise FileNotFoundError(f"{_LOCAL_DEMS[name]} not found under $FORGE3D_REPO_ROOT (forge3d_amd does not download: the reference's "
                            "sample DEMs are git-LFS objects; point FORGE3D_REPO_ROOT at a checkout that has them)")


def real_dem_scene(path, max_side: int = 8193):
    """BASELINE.json config 2 on a REAL DEM file (GeoTIFF or .npy, forge3d_amd.io.load_heightmap): the same orbit camera
    (phi 28, theta 49, radius 1.25 x span, fov 42), sun and material as rainier_proxy_scene.  Heights are shifted to start
    at 0; the cell spacing is the GeoTIFF's pixel scale when that is in metres (> 0.01 -- a geographic raster's degrees are
    turned into metres at the raster's latitude), else 10 m.  Returns (dem, camera, kwargs, description)."""
    from . import io

    p = str(path)
    dem = io.load_heightmap(p)
    spacing_x = spacing_z = 10.0
    if p.lower().endswith((".tif", ".tiff")):
        _, info = io.read_geotiff(p)
        scale, tie = info.get("pixel_scale"), info.get("tiepoint")
        if scale and scale[0] > 0.0 and scale[1] > 0.0:
            if scale[0] > 0.01:
                spacing_x, spacing_z = float(scale[0]), float(scale[1])
            else:  # degrees: metres per degree at the raster's latitude
                lat = float(tie[4]) if tie else 0.0
                spacing_x = float(scale[0]) * 111_320.0 * max(0.1, math.cos(math.radians(lat)))
                spacing_z = float(scale[1]) * 110_574.0
    step = 1
    while max(dem.shape) // step > max_side:  # the reference's node packing holds 8192 cells a side
        step += 1
    if step > 1:
        dem = np.ascontiguousarray(dem[::step, ::step])
        spacing_x, spacing_z = spacing_x * step, spacing_z * step
    dem = np.ascontiguousarray(dem - dem.min(), np.float32)
    span = max((dem.shape[1] - 1) * spacing_x, (dem.shape[0] - 1) * spacing_z)
    cam = orbit_camera((0.0, 0.5 * float(dem.max()), 0.0), 1.25 * span, 28.0, 49.0, 42.0)
    kw = dict(spacing=(spacing_x, spacing_z), exaggeration=1.0, albedo=(0.6, 0.6, 0.6), sun_azimuth_deg=302.0,
              sun_elevation_deg=24.0, sun_intensity=2.5, env_intensity=0.35, seed=7)
    what = f"real DEM {p} ({dem.shape[1]}x{dem.shape[0]}, spacing {spacing_x:.2f} x {spacing_z:.2f} m, relief {float(dem.max()):.0f} m)"
    return dem, cam, kw, what
