"""ctypes front-end of oracle/aether_ref_oracle.c -- TEST INFRASTRUCTURE ONLY (see the C file's header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SRC, _TERRAIN_SRC, _LIB = _HERE / "aether_ref_oracle.c", _HERE / "f3d_oracle.c", _HERE / "libaether_ref_oracle.so"


class Desc(C.Structure):
    _fields_ = [("dem_width", C.c_uint32), ("dem_height", C.c_uint32), ("heights", C.c_void_p), ("spacing_x", C.c_float), ("spacing_z", C.c_float),
                ("exaggeration", C.c_float), ("cam_origin", C.c_float * 3), ("cam_look_at", C.c_float * 3), ("cam_up", C.c_float * 3),
                ("fov_y_deg", C.c_float), ("sun_azimuth_deg", C.c_float), ("sun_elevation_deg", C.c_float), ("sun_intensity", C.c_float),
                ("turbidity", C.c_float), ("ozone_du", C.c_float), ("mie_g", C.c_float), ("ground_albedo", C.c_float), ("width", C.c_uint32),
                ("height", C.c_uint32), ("seed", C.c_uint32), ("spp", C.c_uint32), ("enabled", C.c_int32), ("variance_threshold", C.c_float)]


def build(force: bool = False) -> Path:
    if force or not _LIB.exists() or _LIB.stat().st_mtime < max(_SRC.stat().st_mtime, _TERRAIN_SRC.stat().st_mtime):
        tmp = _LIB.with_suffix(f".{os.getpid()}.tmp")
        subprocess.run(["gcc", "-O2", "-fopenmp", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-shared", "-D_POSIX_C_SOURCE=200809L", str(_SRC),
                        str(_TERRAIN_SRC), "-o", str(tmp), "-lm"], check=True, capture_output=True)
        os.replace(tmp, _LIB)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB))
        _lib.aether_ref_oracle_render.restype = C.c_int
    return _lib


def fill_desc(d, heightmap, width, height, cam, spacing=(1.0, 1.0), exaggeration=1.0, sun_azimuth_deg=90.0, sun_elevation_deg=10.0,
              sun_intensity=20.0, turbidity=2.0, ozone_du=300.0, mie_g=0.8, ground_albedo=0.3, spp=64, seed=7, enabled=True,
              variance_threshold=1e-3):
    """Fill a descriptor structure (this oracle's or the product's: same field names) the way the reference's Python seam
    does (src/py_functions/path_tracing/aether_reference.rs:14-116: defaults and camera keys); returns the DEM to keep alive."""
    dem = np.ascontiguousarray(heightmap, dtype=np.float32)
    if dem.ndim != 2:
        raise TypeError("heightmap must be a 2-D float32 array")
    d.dem_height, d.dem_width = dem.shape
    d.heights = dem.ctypes.data
    d.spacing_x, d.spacing_z, d.exaggeration = float(spacing[0]), float(spacing[1]), float(exaggeration)
    d.cam_origin = (C.c_float * 3)(*cam.get("origin", (0.0, 1.0, 0.0)))
    d.cam_look_at = (C.c_float * 3)(*cam.get("look_at", (1.0, 1.0, 0.0)))
    d.cam_up = (C.c_float * 3)(*cam.get("up", (0.0, 1.0, 0.0)))
    d.fov_y_deg = float(cam.get("fov_y", 20.0))
    d.sun_azimuth_deg, d.sun_elevation_deg, d.sun_intensity = float(sun_azimuth_deg), float(sun_elevation_deg), float(sun_intensity)
    d.turbidity, d.ozone_du, d.mie_g, d.ground_albedo = float(turbidity), float(ozone_du), float(mie_g), float(ground_albedo)
    d.width, d.height, d.seed, d.spp = int(width), int(height), int(seed) & 0xFFFFFFFF, int(spp)
    d.enabled, d.variance_threshold = 1 if enabled else 0, float(variance_threshold)
    return dem


def render(heightmap, width, height, cam, **kw) -> dict:
    d = Desc()
    keep = fill_desc(d, heightmap, width, height, cam, **kw)
    mean_xyz = np.zeros((height, width, 3), np.float32)
    rgb = np.zeros((height, width, 3), np.float32)
    scalars = (C.c_float * 2)()
    hits = C.c_uint64(0)
    err = C.create_string_buffer(512)
    rc = lib().aether_ref_oracle_render(C.byref(d), C.c_void_p(mean_xyz.ctypes.data), C.c_void_p(rgb.ctypes.data), scalars, C.byref(hits), err, len(err))
    del keep
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return {"mean_xyz": mean_xyz, "linear_rgb": rgb, "variance": float(scalars[0]), "converged": bool(scalars[1]), "seed": int(d.seed), "spp": int(d.spp),
            "terrain_primary_hits": int(hits.value), "environment": "black", "wavelength_count": 11, "max_depth": 6}
