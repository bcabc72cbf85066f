"""The Python entry point against the reference's own wrapper contract.

tests/golden/wrapper_contract.json was captured from the reference package
(`forge3d.path_tracing.hybrid_render_terrain_reference`, reference
python/forge3d/path_tracing.py:893-1095) by tests/golden/make_fixtures.py: the signature
model the reference's test asserts (tests/test_hybrid_terrain_pt.py:461-500), the
(input -> exception type, message) pairs of its pure-Python validation, and what it forwards
to the native function by default.  No GPU needed: the native seam is monkeypatched the way
the reference's test does it (:860-876).
"""
from __future__ import annotations

import inspect
import json

import numpy as np
import pytest

import scenes

CONTRACT = json.loads((scenes.GOLDEN_DIR / "wrapper_contract.json").read_text())
ARRAYS = {"zeros3d": np.zeros((2, 2, 2), np.float32), "zeros1x1": np.zeros((1, 1), np.float32),
          "nan16": np.full((16, 16), np.nan, np.float32), "zeros3x3": np.zeros((3, 3), np.float32),
          "zeros4x4": np.zeros((4, 4), np.float32), "zeros3x2": np.zeros((3, 2), np.float32),
          "idx1x3": np.zeros((1, 3), np.uint32)}
CASES = {
    "ndim": dict(heightmap="zeros3d"), "tiny": dict(heightmap="zeros1x1"), "nan": dict(heightmap="nan16"),
    "min_gt_max": dict(max_frames=4, min_frames=8), "spp0": dict(spp=0), "spp65": dict(spp=65),
    "spacing0": dict(spacing=(0.0, 1.0)), "sun_nan": dict(sun_color=(1.0, float("nan"), 1.0)),
    "sun_neg": dict(sun_color=(1.0, -0.1, 1.0)), "sun_two": dict(sun_color=(1.0, 1.0)),
    "sun_four": dict(sun_color=(1.0, 1.0, 1.0, 1.0)), "sun_scalar": dict(sun_color=0.5),
    "sun_str": dict(sun_color="abc"), "sun_strs": dict(sun_color=("0.5", "0.9", "0.8")),
    "mesh_alone": dict(mesh_vertices="zeros3x3"), "env_shape": dict(env_map="zeros4x4"),
    "mesh_shape": dict(mesh_vertices="zeros3x2", mesh_indices="idx1x3"),
}


class _Native:
    def __init__(self):
        self.calls = []

    def hybrid_render_terrain_reference(self, *args, **kwargs):
        self.calls.append((args, kwargs))
        return {}


@pytest.fixture()
def patched(monkeypatch):
    import forge3d_amd.path_tracing as pt

    native = _Native()
    monkeypatch.setattr(pt, "_NATIVE", native)
    return pt, native


def test_signature_is_the_reference_signature():
    import forge3d_amd
    from forge3d_amd.path_tracing import hybrid_render_terrain_reference as fn

    assert forge3d_amd.hybrid_render_terrain_reference is fn
    rows = []
    for p in inspect.signature(fn).parameters.values():
        default = "<required>" if p.default is inspect._empty else p.default
        ann = "" if p.annotation is inspect._empty else str(p.annotation)
        if len(ann) >= 2 and ann[0] in "'\"" and ann[-1] == ann[0]:
            ann = ann[1:-1]
        rows.append([p.name, p.kind.name, list(default) if isinstance(default, tuple) else default, ann])
    assert rows == CONTRACT["signature"]


def test_native_seam_has_the_reference_native_order_and_defaults():
    """native signature model of reference tests/test_hybrid_terrain_pt.py:533-565"""
    from forge3d_amd import _native

    got = [(p.name, p.default) for p in inspect.signature(_native.hybrid_render_terrain_reference).parameters.values()]
    E = inspect._empty
    want = [("heightmap", E), ("width", E), ("height", E), ("cam", E), ("spacing", (1.0, 1.0)), ("exaggeration", 1.0),
            ("albedo", (0.6, 0.6, 0.6)), ("sun_azimuth_deg", 315.0), ("sun_elevation_deg", 45.0),
            ("sun_intensity", 2.5), ("env_map", None), ("env_intensity", 0.35), ("mesh_vertices", None),
            ("mesh_indices", None), ("spp", 1), ("max_frames", 512), ("min_frames", 32), ("variance_threshold", 1e-3),
            ("seed", 7), ("certificate", None), ("sun_color", None), ("cache", None), ("observer_latitude_deg", 0.0),
            ("observer_longitude_deg", 0.0), ("earth_model", "ellipsoid"), ("sphere_radius_m", 6371008.8),
            ("refraction_model", "bennett"), ("refraction_k", 0.13), ("pressure_mbar", 1013.25),
            ("temperature_c", 15.0), ("atmosphere", None)]
    assert got == want


@pytest.mark.parametrize("name", sorted(CASES))
def test_validation_errors_match_the_reference(patched, name):
    pt, native = patched
    kw = {k: (ARRAYS[v] if isinstance(v, str) and v in ARRAYS else v) for k, v in CASES[name].items()}
    hm = kw.pop("heightmap", np.zeros((4, 4), np.float32))
    exc_name, message = CONTRACT["errors"][name]
    with pytest.raises(Exception) as info:
        pt.hybrid_render_terrain_reference(hm, 8, 8, scenes.CAM, **kw)
    assert type(info.value).__name__ == exc_name
    assert str(info.value) == message
    assert native.calls == []  # rejected before any native / GPU work


def test_forwarded_defaults_match_the_reference(patched):
    pt, native = patched
    pt.hybrid_render_terrain_reference(np.zeros((4, 4), np.float32), 8, 8, scenes.CAM)
    args, kwargs = native.calls[0]
    assert args[1:3] == (8, 8) and isinstance(args[3], dict)
    got = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kwargs.items()
           if isinstance(v, (int, float, str, tuple, bool, type(None)))}
    assert got == CONTRACT["forwarded_defaults"]


def test_solar_time_is_exclusive_with_manual_angles_and_reports_its_source(patched):
    """reference test_public_wrapper_resolves_solar_time_and_reports_source (:860-930)"""
    pt, native = patched

    class When:
        observer_lat, observer_lon, pressure_mbar, temperature_c = 39.742476, -105.1786, 820.0, 11.0

        def position(self):
            return {"azimuth_deg": 194.34, "true_elevation_deg": 39.87, "apparent_elevation_deg": 39.89}

    dem = np.zeros((2, 2), np.float32)
    result = pt.hybrid_render_terrain_reference(dem, 2, 2, solar_time=When(), min_frames=1, max_frames=1)
    kwargs = native.calls[-1][1]
    assert kwargs["sun_azimuth_deg"] == pytest.approx(194.34)
    assert kwargs["sun_elevation_deg"] == pytest.approx(39.89)
    assert kwargs["pressure_mbar"] == 820.0 and kwargs["observer_latitude_deg"] == pytest.approx(39.742476)
    assert "sun_source" not in kwargs and result["sun_source"] == "solar_time"
    pt.hybrid_render_terrain_reference(dem, 2, 2, solar_time=When(), refraction_model="none", min_frames=1,
                                       max_frames=1)
    assert native.calls[-1][1]["sun_elevation_deg"] == pytest.approx(39.87)
    for extra in (dict(sun_azimuth_deg=123.0), dict(pressure_mbar=900.0)):
        with pytest.raises(ValueError, match="cannot be combined"):
            pt.hybrid_render_terrain_reference(dem, 2, 2, solar_time=When(), min_frames=1, max_frames=1, **extra)
    manual = pt.hybrid_render_terrain_reference(dem, 2, 2, min_frames=1, max_frames=1)
    assert manual["sun_source"] == "manual_angles" and manual["solar_azimuth_deg"] == 315.0


def test_unrelated_native_failures_propagate(monkeypatch):
    """reference test_sun_color_valid_input_does_not_suppress_unrelated_failures (:683-707)"""
    import forge3d_amd.path_tracing as pt

    class Broken:
        @staticmethod
        def hybrid_render_terrain_reference(*a, **k):
            raise RuntimeError("invalid device adapter state should propagate")

    monkeypatch.setattr(pt, "_NATIVE", Broken())
    with pytest.raises(RuntimeError, match="adapter state"):
        pt.hybrid_render_terrain_reference(np.zeros((2, 2), np.float32), 8, 8, scenes.CAM, sun_color=[1.0, 1.0, 1.0],
                                           max_frames=4, min_frames=2, variance_threshold=1e30)
    monkeypatch.setattr(pt, "_NATIVE", None)
    with pytest.raises(RuntimeError, match="requires the native"):
        pt.hybrid_render_terrain_reference(np.zeros((2, 2), np.float32), 8, 8)


def test_native_boundary_rejects_malformed_sun_color_and_unknown_settings():
    """extract_sun_color / atmosphere key validation of the PyO3 seam
    (reference tests/test_hybrid_terrain_pt.py:614-637, terrain_reference.rs:12-93)"""
    from forge3d_amd import _native

    dem = np.zeros((4, 4), np.float32)
    fast = {"max_frames": 4, "min_frames": 2, "variance_threshold": 1e30}
    bad = [(1.0, -1.0, 1.0), (1.0, float("nan"), 1.0), (1.0, float("inf"), 1.0), (1.0, 1.0), (1.0, 1.0, 1.0, 1.0), 0.5,
           "abc", ("0.5", "0.9", "0.8"), np.float64(0.5), np.array([1.0, 1.0]), bytearray([1, 1, 1]),
           memoryview(bytes([1, 1, 1]))]
    for value in bad:
        with pytest.raises(ValueError):
            _native.hybrid_render_terrain_reference(dem, 8, 8, dict(scenes.CAM), sun_color=value, **fast)
    with pytest.raises(ValueError, match="unknown atmosphere setting"):
        _native.hybrid_render_terrain_reference(dem, 8, 8, dict(scenes.CAM), atmosphere={"fog": 1}, **fast)
    with pytest.raises(ValueError, match="earth_model"):
        _native.hybrid_render_terrain_reference(dem, 8, 8, dict(scenes.CAM), earth_model="mean-earth", **fast)
    with pytest.raises(ValueError, match="refraction_model"):
        _native.hybrid_render_terrain_reference(dem, 8, 8, dict(scenes.CAM), refraction_model="standard", **fast)
    with pytest.raises(ValueError, match="together"):
        _native.hybrid_render_terrain_reference(dem, 8, 8, dict(scenes.CAM), mesh_vertices=np.zeros((3, 3), np.float32),
                                                **fast)
