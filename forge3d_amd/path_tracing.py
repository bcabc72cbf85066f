"""Host-side mirror of ``forge3d.path_tracing.hybrid_render_terrain_reference``.

Same name, parameters, defaults, validation order, error types/substrings and return dict
as the reference wrapper (reference python/forge3d/path_tracing.py:893-1095) so callers --
and the reference's own tests (tests/test_hybrid_terrain_pt.py) -- can switch the import.
The render itself runs in ``forge3d_amd._native`` -> libf3dhip.so -> HIP kernels on gfx950.
"""
from __future__ import annotations

from typing import Any, Mapping, Sequence

import numpy as np

from . import _native as _NATIVE_MODULE

# The reference binds the compiled extension module here (`_NATIVE`); tests monkeypatch it.
_NATIVE = _NATIVE_MODULE

_TEXT_TYPES = (str, bytes, bytearray, memoryview)


def _sun_rgb(sun_color):
    """Wrapper-level sun_color checks (reference path_tracing.py:985-1001)."""
    if isinstance(sun_color, _TEXT_TYPES):
        raise ValueError(f"sun_color must be three numbers, got {sun_color!r}")
    try:
        n = len(sun_color)
        if n != 3:
            raise ValueError(f"sun_color must have exactly three components, got {n}")
        rgb = tuple(float(sun_color[i]) for i in range(3))
    except (TypeError, ValueError) as exc:
        raise ValueError(f"sun_color must be a sequence of three numbers: {exc}")
    if any(isinstance(c, _TEXT_TYPES) for c in sun_color):
        raise ValueError(f"sun_color components must be numbers, got {sun_color!r}")
    if not all(bool(np.isfinite(c)) for c in rgb):
        raise ValueError(f"sun_color components must be finite, got {sun_color!r}")
    if any(c < 0.0 for c in rgb):
        raise ValueError(f"sun_color components must be non-negative, got {sun_color!r}")
    return rgb


def hybrid_render_terrain_reference(
    heightmap: "np.ndarray",
    width: int,
    height: int,
    camera: "dict | None" = None,
    *,
    spacing: "tuple[float, float]" = (1.0, 1.0),
    exaggeration: float = 1.0,
    albedo: "tuple[float, float, float]" = (0.6, 0.6, 0.6),
    sun_azimuth_deg: float | None = None,
    sun_elevation_deg: float | None = None,
    solar_time: "object | None" = None,
    sun_intensity: float = 2.5,
    sun_color: "Sequence[float] | np.ndarray" = (1.0, 0.97, 0.92),
    env_map: "np.ndarray | None" = None,
    env_intensity: float = 0.35,
    mesh_vertices: "np.ndarray | None" = None,
    mesh_indices: "np.ndarray | None" = None,
    spp: int = 1,
    max_frames: int = 512,
    min_frames: int = 32,
    variance_threshold: float = 1e-3,
    seed: int = 7,
    certificate: bool | str = False,
    cache: str | None = None,
    observer_latitude_deg: float | None = None,
    observer_longitude_deg: float | None = None,
    earth_model: str = "ellipsoid",
    sphere_radius_m: float = 6_371_008.8,
    refraction_model: str = "bennett",
    refraction_k: float = 0.13,
    pressure_mbar: float | None = None,
    temperature_c: float | None = None,
    atmosphere: "Mapping[str, Any] | Any | None" = None,
) -> dict:
    """Converged GPU path-traced reference of a DEM under sun + IBL (MI355X / HIP backend).

    Drop-in for the reference function of the same name: accumulates frames of ``spp``
    tent-jittered camera samples (min-max quadtree heightfield traversal with the exact
    bilinear-patch leaf solve, sun through the ReSTIR reservoir chain, one cosine-weighted
    IBL ray) until the per-pixel variance of the running-mean luminance over the last
    32-frame window drops below ``variance_threshold``; raises instead of returning an
    unconverged image.  Returns ``rgba`` (H,W,4) uint8, ``albedo``/``normal`` (H,W,3)
    float32, ``depth`` (H,W) float32 (NaN on miss), ``frames``, ``variance``, ``converged``
    and the memory diagnostics, plus ``sun_source`` / ``solar_*_deg``.
    """
    _ = cache
    if _NATIVE is None or not hasattr(_NATIVE, "hybrid_render_terrain_reference"):
        raise RuntimeError(
            "hybrid_render_terrain_reference requires the native forge3d module with GPU support"
        )
    dem = np.ascontiguousarray(heightmap, dtype=np.float32)
    if dem.ndim != 2:
        raise ValueError(f"heightmap must be 2D (H, W), got shape {dem.shape}")
    if min(dem.shape) < 2:
        raise ValueError(
            f"terrain heightfield must be at least 2x2 texels, got {dem.shape[1]}x{dem.shape[0]}"
        )
    if not np.isfinite(dem).all():
        raise ValueError("heightmap contains non-finite samples")
    if int(min_frames) > int(max_frames):
        raise ValueError(f"min_frames ({min_frames}) must be <= max_frames ({max_frames})")
    if not 1 <= int(spp) <= 64:
        raise ValueError(f"spp must be in 1..=64, got {spp}")
    if not (float(spacing[0]) > 0.0 and float(spacing[1]) > 0.0):
        raise ValueError(f"spacing must be > 0, got {spacing}")
    sun_rgb = _sun_rgb(sun_color)

    sun_source = "manual_angles"
    if solar_time is not None:
        manual = (sun_azimuth_deg, sun_elevation_deg, observer_latitude_deg, observer_longitude_deg,
                  pressure_mbar, temperature_c)
        if any(value is not None for value in manual):
            raise ValueError(
                "solar_time cannot be combined with manual sun, observer, pressure, or temperature values"
            )
        # The reference resolves SolarTime through its native NREL-SPA (python/forge3d/geo.py:23-52,
        # src/geo/solar.rs:77) -- an adjacent feature outside the terrain-PT hot path.
        from .geo import resolve_solar_time

        solar = resolve_solar_time(solar_time)
        sun_azimuth_deg = solar["azimuth_deg"]
        sun_elevation_deg = solar[
            "true_elevation_deg" if refraction_model == "none" else "apparent_elevation_deg"
        ]
        observer_latitude_deg = solar["observer_lat"]
        observer_longitude_deg = solar["observer_lon"]
        pressure_mbar = solar["pressure_mbar"]
        temperature_c = solar["temperature_c"]
        sun_source = "solar_time"
    else:
        sun_azimuth_deg = 315.0 if sun_azimuth_deg is None else sun_azimuth_deg
        sun_elevation_deg = 45.0 if sun_elevation_deg is None else sun_elevation_deg
        observer_latitude_deg = 0.0 if observer_latitude_deg is None else observer_latitude_deg
        observer_longitude_deg = 0.0 if observer_longitude_deg is None else observer_longitude_deg
        pressure_mbar = 1013.25 if pressure_mbar is None else pressure_mbar
        temperature_c = 15.0 if temperature_c is None else temperature_c

    cam = dict(camera or {})
    env = None
    if env_map is not None:
        env = np.ascontiguousarray(env_map, dtype=np.float32)
        if env.ndim != 3 or env.shape[2] != 3:
            raise ValueError(f"env_map must be (H, W, 3) float32, got {env.shape}")
    if (mesh_vertices is None) != (mesh_indices is None):
        raise ValueError("mesh_vertices and mesh_indices must be provided together")
    mv = mi = None
    if mesh_vertices is not None:
        mv = np.ascontiguousarray(mesh_vertices, dtype=np.float32)
        mi = np.ascontiguousarray(mesh_indices, dtype=np.uint32)
        if mv.ndim != 2 or mv.shape[1] != 3:
            raise ValueError(f"mesh_vertices must be (N, 3), got {mv.shape}")
        if mi.ndim != 2 or mi.shape[1] != 3:
            raise ValueError(f"mesh_indices must be (M, 3), got {mi.shape}")

    result = _NATIVE.hybrid_render_terrain_reference(
        dem,
        int(width),
        int(height),
        cam,
        spacing=(float(spacing[0]), float(spacing[1])),
        exaggeration=float(exaggeration),
        albedo=(float(albedo[0]), float(albedo[1]), float(albedo[2])),
        sun_azimuth_deg=float(sun_azimuth_deg),
        sun_elevation_deg=float(sun_elevation_deg),
        sun_intensity=float(sun_intensity),
        sun_color=sun_rgb,
        env_map=env,
        env_intensity=float(env_intensity),
        mesh_vertices=mv,
        mesh_indices=mi,
        spp=int(spp),
        max_frames=int(max_frames),
        min_frames=int(min_frames),
        variance_threshold=float(variance_threshold),
        seed=int(seed),
        certificate=certificate,
        observer_latitude_deg=float(observer_latitude_deg),
        observer_longitude_deg=float(observer_longitude_deg),
        earth_model=earth_model,
        sphere_radius_m=float(sphere_radius_m),
        refraction_model=refraction_model,
        refraction_k=float(refraction_k),
        pressure_mbar=float(pressure_mbar),
        temperature_c=float(temperature_c),
        atmosphere=atmosphere,
    )
    result["sun_source"] = sun_source
    result["solar_azimuth_deg"] = float(sun_azimuth_deg)
    result["solar_elevation_deg"] = float(sun_elevation_deg)
    return result
