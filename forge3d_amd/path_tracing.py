"""Public entry point of the terrain path tracer: ``hybrid_render_terrain_reference``.

The CONTRACT is the reference's (python/forge3d/path_tracing.py:893-929 signature; :970-1001 validation
order, exception types and message texts; :1092-1094 the three sun keys added to the result) and is pinned
by tests/golden/wrapper_contract.json, captured from the reference package.  The implementation is this
package's own: the call is normalised into a ``_Request`` (arrays coerced once, options kept by name), an
ordered rule table rejects bad input before any device work, the sun is resolved by ``_SunResolution`` and
the native keyword set is produced from a conversion table.  The render runs in ``forge3d_amd._native`` ->
libf3dhip.so -> HIP kernels on gfx950; there is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Mapping, Sequence

import numpy as np

from . import _native as _NATIVE_MODULE

# The seam the reference's tests monkeypatch (tests/test_hybrid_terrain_pt.py:860-876).
_NATIVE = _NATIVE_MODULE

_TEXTUAL = (str, bytes, bytearray, memoryview)


# ---- sun colour ------------------------------------------------------------------------------------
def _three_floats(value) -> tuple:
    """Length and float conversion of a colour triple; raises the TypeError / ValueError Python itself
    produces (their text is part of the contract: 'object of type 'float' has no len()')."""
    count = len(value)
    if count != 3:
        raise ValueError(f"sun_color must have exactly three components, got {count}")
    return tuple(map(float, (value[0], value[1], value[2])))


def _parse_sun_color(value) -> tuple:
    if isinstance(value, _TEXTUAL):
        raise ValueError(f"sun_color must be three numbers, got {value!r}")
    try:
        rgb = _three_floats(value)
    except (TypeError, ValueError) as problem:
        raise ValueError(f"sun_color must be a sequence of three numbers: {problem}")
    for test, what in ((lambda raw, c: isinstance(raw, _TEXTUAL), "numbers"),
                       (lambda raw, c: not np.isfinite(c), "finite"),
                       (lambda raw, c: c < 0.0, "non-negative")):
        if any(test(raw, c) for raw, c in zip(value, rgb)):
            raise ValueError(f"sun_color components must be {what}, got {value!r}")
    return rgb


# ---- the normalised call ---------------------------------------------------------------------------
@dataclass
class _Request:
    dem: np.ndarray
    width: int
    height: int
    camera: dict
    opt: dict                      # keyword options by name, as passed
    env: "np.ndarray | None" = None
    mesh_v: "np.ndarray | None" = None
    mesh_i: "np.ndarray | None" = None
    sun_rgb: tuple = field(default=(1.0, 0.97, 0.92))


# Ordered rejection rules: (is the request bad?, message).  Evaluated top to bottom, first hit raises.
_RULES: "tuple[tuple[Callable[[_Request], bool], Callable[[_Request], str]], ...]" = (
    (lambda q: q.dem.ndim != 2,
     lambda q: f"heightmap must be 2D (H, W), got shape {q.dem.shape}"),
    (lambda q: min(q.dem.shape) < 2,
     lambda q: f"terrain heightfield must be at least 2x2 texels, got {q.dem.shape[1]}x{q.dem.shape[0]}"),
    # (min / max propagate NaN and keep infinities: two reductions instead of a 4 MB temporary -- 0.9 instead of 3.6 ms for a 2048^2 DEM)
    (lambda q: not (np.isfinite(q.dem.min()) and np.isfinite(q.dem.max())),
     lambda q: "heightmap contains non-finite samples"),
    (lambda q: int(q.opt["min_frames"]) > int(q.opt["max_frames"]),
     lambda q: f"min_frames ({q.opt['min_frames']}) must be <= max_frames ({q.opt['max_frames']})"),
    (lambda q: not 1 <= int(q.opt["spp"]) <= 64,
     lambda q: f"spp must be in 1..=64, got {q.opt['spp']}"),
    (lambda q: not all(float(s) > 0.0 for s in q.opt["spacing"][:2]),
     lambda q: f"spacing must be > 0, got {q.opt['spacing']}"),
)


def _coerce_scene_arrays(q: _Request) -> None:
    """Environment map and mesh: shapes checked after the sun has been resolved, like the reference."""
    env_map, verts, faces = q.opt["env_map"], q.opt["mesh_vertices"], q.opt["mesh_indices"]
    if env_map is not None:
        q.env = np.ascontiguousarray(env_map, dtype=np.float32)
        if q.env.ndim != 3 or q.env.shape[2] != 3:
            raise ValueError(f"env_map must be (H, W, 3) float32, got {q.env.shape}")
    if (verts is None) is not (faces is None):
        raise ValueError("mesh_vertices and mesh_indices must be provided together")
    if verts is None:
        return
    q.mesh_v = np.ascontiguousarray(verts, dtype=np.float32)
    q.mesh_i = np.ascontiguousarray(faces, dtype=np.uint32)
    for arr, label, letter in ((q.mesh_v, "mesh_vertices", "N"), (q.mesh_i, "mesh_indices", "M")):
        if arr.ndim != 2 or arr.shape[1] != 3:
            raise ValueError(f"{label} must be ({letter}, 3), got {arr.shape}")


# ---- where the sun comes from ----------------------------------------------------------------------
_MANUAL_SUN_DEFAULTS = {"sun_azimuth_deg": 315.0, "sun_elevation_deg": 45.0, "observer_latitude_deg": 0.0,
                        "observer_longitude_deg": 0.0, "pressure_mbar": 1013.25, "temperature_c": 15.0}


@dataclass
class _SunResolution:
    source: str
    values: dict  # the six keys of _MANUAL_SUN_DEFAULTS, resolved

    @classmethod
    def of(cls, opt: Mapping[str, Any]) -> "_SunResolution":
        given = {name: opt[name] for name in _MANUAL_SUN_DEFAULTS}
        when = opt["solar_time"]
        if when is None:
            return cls("manual_angles", {name: (default if given[name] is None else given[name])
                                         for name, default in _MANUAL_SUN_DEFAULTS.items()})
        if not all(v is None for v in given.values()):
            raise ValueError(
                "solar_time cannot be combined with manual sun, observer, pressure, or temperature values")
        # The reference evaluates SolarTime with its native NREL SPA (python/forge3d/geo.py:23-52,
        # src/geo/solar.rs:77); here forge3d_amd.geo does (adjacent to the hot path).
        from .geo import resolve_solar_time

        solar = resolve_solar_time(when)
        elevation = "true_elevation_deg" if opt["refraction_model"] == "none" else "apparent_elevation_deg"
        return cls("solar_time", {
            "sun_azimuth_deg": solar["azimuth_deg"], "sun_elevation_deg": solar[elevation],
            "observer_latitude_deg": solar["observer_lat"], "observer_longitude_deg": solar["observer_lon"],
            "pressure_mbar": solar["pressure_mbar"], "temperature_c": solar["temperature_c"]})


# ---- what the native function receives --------------------------------------------------------------
def _floats(n: int) -> Callable[[Any], tuple]:
    return lambda v: tuple(float(v[i]) for i in range(n))


def _as_is(v):
    return v


# keyword -> conversion, in the native function's order (src/py_functions/path_tracing/terrain_reference.rs:224-256)
_NATIVE_KEYWORDS: "tuple[tuple[str, Callable[[Any], Any]], ...]" = (
    ("spacing", _floats(2)), ("exaggeration", float), ("albedo", _floats(3)), ("sun_azimuth_deg", float),
    ("sun_elevation_deg", float), ("sun_intensity", float), ("sun_color", _as_is), ("env_map", _as_is),
    ("env_intensity", float), ("mesh_vertices", _as_is), ("mesh_indices", _as_is), ("spp", int),
    ("max_frames", int), ("min_frames", int), ("variance_threshold", float), ("seed", int),
    ("certificate", _as_is), ("observer_latitude_deg", float), ("observer_longitude_deg", float),
    ("earth_model", _as_is), ("sphere_radius_m", float), ("refraction_model", _as_is), ("refraction_k", float),
    ("pressure_mbar", float), ("temperature_c", float), ("atmosphere", _as_is),
)


def hybrid_render_terrain_reference(
        heightmap: "np.ndarray", width: int, height: int, camera: "dict | None" = None, *,
        spacing: "tuple[float, float]" = (1.0, 1.0), exaggeration: float = 1.0,
        albedo: "tuple[float, float, float]" = (0.6, 0.6, 0.6),
        sun_azimuth_deg: float | None = None, sun_elevation_deg: float | None = None,
        solar_time: "object | None" = None, sun_intensity: float = 2.5,
        sun_color: "Sequence[float] | np.ndarray" = (1.0, 0.97, 0.92),
        env_map: "np.ndarray | None" = None, env_intensity: float = 0.35,
        mesh_vertices: "np.ndarray | None" = None, mesh_indices: "np.ndarray | None" = None,
        spp: int = 1, max_frames: int = 512, min_frames: int = 32, variance_threshold: float = 1e-3, seed: int = 7,
        certificate: bool | str = False, cache: str | None = None,
        observer_latitude_deg: float | None = None, observer_longitude_deg: float | None = None,
        earth_model: str = "ellipsoid", sphere_radius_m: float = 6_371_008.8,
        refraction_model: str = "bennett", refraction_k: float = 0.13,
        pressure_mbar: float | None = None, temperature_c: float | None = None,
        atmosphere: "Mapping[str, Any] | Any | None" = None,
) -> dict:
    """Converged path-traced reference image of a DEM under sun + sky on an MI355X.

    Accumulates frames of ``spp`` tent-jittered camera samples (heightfield traversal with the exact
    bilinear-patch solve, the sun through the ReSTIR reservoir chain, one cosine-weighted sky ray) until the
    per-pixel variance of the running-mean luminance over the last 32-frame window is below
    ``variance_threshold``; an unconverged render raises, it never returns.  Result: ``rgba`` (H,W,4) uint8,
    ``albedo`` / ``normal`` (H,W,3) float32, ``depth`` (H,W) float32 with NaN where nothing was hit,
    ``frames``, ``variance``, ``converged``, the memory diagnostics, and ``sun_source`` /
    ``solar_azimuth_deg`` / ``solar_elevation_deg``.  ``cache`` is accepted and ignored.
    """
    options = dict(locals())
    for positional in ("heightmap", "width", "height", "camera"):
        options.pop(positional)
    if getattr(_NATIVE, "hybrid_render_terrain_reference", None) is None:
        raise RuntimeError("hybrid_render_terrain_reference requires the native forge3d module with GPU support")

    request = _Request(np.ascontiguousarray(heightmap, dtype=np.float32), int(width), int(height),
                       dict(camera or {}), options)
    for is_bad, message in _RULES:
        if is_bad(request):
            raise ValueError(message(request))
    request.sun_rgb = _parse_sun_color(sun_color)
    sun = _SunResolution.of(options)
    _coerce_scene_arrays(request)

    values = {**options, **sun.values, "sun_color": request.sun_rgb, "env_map": request.env,
              "mesh_vertices": request.mesh_v, "mesh_indices": request.mesh_i}
    keywords = {name: convert(values[name]) for name, convert in _NATIVE_KEYWORDS}
    result = _NATIVE.hybrid_render_terrain_reference(request.dem, request.width, request.height, request.camera,
                                                     **keywords)
    result.update(sun_source=sun.source, solar_azimuth_deg=float(sun.values["sun_azimuth_deg"]),
                  solar_elevation_deg=float(sun.values["sun_elevation_deg"]))
    return result
