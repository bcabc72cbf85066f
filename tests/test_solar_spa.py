"""NREL SPA behind `solar_time=` (forge3d_amd/geo.py), pinned by the reference's own vectors.

Restates reference tests/test_solar_spa.py:28-146 (worked example of the NREL report, the official rows of
tests/data/spa_reference.csv -- committed here as tests/golden/spa/spa_reference.csv --, the timezone-free SolarTime
contract, rejected inputs) and src/geo/solar.rs:277-299 (Rust unit test of the worked example)."""
from __future__ import annotations

import csv
from pathlib import Path

import numpy as np
import pytest

from forge3d_amd.geo import SolarTime, _coerce_solar_time, solar_position

REFERENCE = Path(__file__).parent / "golden" / "spa" / "spa_reference.csv"
ANGLE_TOLERANCE_DEG = 0.0003


def _angle_error(actual, expected):
    delta = abs(actual - expected) % 360.0
    return min(delta, 360.0 - delta)


def test_spa_worked_example_matches_nrel():
    r = solar_position((2003, 10, 17, 12, 30, 30), 39.742476, -105.1786, 1830.14, tz_offset_hours=-7, delta_t_seconds=67,
                       pressure_mbar=820, temperature_c=11)
    assert abs(r["zenith_deg"] - 50.11162) <= ANGLE_TOLERANCE_DEG
    assert _angle_error(r["azimuth_deg"], 194.34024) <= ANGLE_TOLERANCE_DEG


def test_spa_matches_official_reference_rows():
    with REFERENCE.open(newline="", encoding="utf-8") as stream:
        rows = [{k: float(v) for k, v in row.items()} for row in csv.DictReader(stream)]
    assert len(rows) >= 20 and min(r["lat"] for r in rows) <= -80 and max(r["lat"] for r in rows) >= 80
    assert min(r["year"] for r in rows) <= 1900 and max(r["year"] for r in rows) >= 2100
    for row in rows:
        got = solar_position(tuple(int(row[k]) for k in ("year", "month", "day", "hour", "minute", "second")), row["lat"], row["lon"],
                             row["elev_m"], tz_offset_hours=row["tz_offset_hours"], delta_t_seconds=row["delta_t_seconds"],
                             pressure_mbar=row["pressure_mbar"], temperature_c=row["temperature_c"])
        label = f"{int(row['year'])}-{int(row['month']):02}-{int(row['day']):02}@{row['lat']},{row['lon']}"
        assert abs(got["zenith_deg"] - row["zenith_deg"]) <= ANGLE_TOLERANCE_DEG, label
        assert _angle_error(got["azimuth_deg"], row["azimuth_deg"]) <= ANGLE_TOLERANCE_DEG, label
        assert abs(got["true_elevation_deg"] - row["true_elevation_deg"]) <= ANGLE_TOLERANCE_DEG, label
        assert abs(got["distance_au"] - row["distance_au"]) <= 5.1e-7, label
        assert abs(got["equation_of_time_min"] - row["equation_of_time_min"]) <= 5.1e-7, label


def test_solar_time_is_an_explicit_timezone_free_contract():
    when = SolarTime(utc=(2025, 6, 21, 12, 0, 0), observer_lat=48.2082, observer_lon=16.3738, observer_elev_m=171, tz_offset_hours=2,
                     delta_t_seconds=74.5, pressure_mbar=1000, temperature_c=25)
    assert when.position()["azimuth_deg"] == pytest.approx(150.7305003541, abs=ANGLE_TOLERANCE_DEG)
    assert when.to_native()["utc"] == (2025, 6, 21, 12, 0, 0) and "delta_t" not in when.to_native()
    same = _coerce_solar_time({"utc": (2025, 6, 21, 12, 0, 0), "observer_lat": 48.2082, "observer_lon": 16.3738, "observer_elev_m": 171,
                               "tz_offset_hours": 2, "delta_t": 74.5, "pressure_mbar": 1000, "temperature_c": 25})
    assert same.position() == when.position()
    with pytest.raises(TypeError, match="SolarTime or a mapping"):
        _coerce_solar_time(42)


@pytest.mark.parametrize("field,value", [("latitude", 90.0001), ("longitude", 180.0001), ("pressure_mbar", 0.0),
                                         ("temperature_c", -273.15), ("tz_offset_hours", 19.0)])
def test_spa_rejects_invalid_physical_inputs(field, value):
    kwargs = {"tz_offset_hours": 0.0, "delta_t_seconds": 74.0, "pressure_mbar": 1013.25, "temperature_c": 15.0}
    lat, lon = 45.0, 5.0
    if field == "latitude":
        lat = value
    elif field == "longitude":
        lon = value
    else:
        kwargs[field] = value
    with pytest.raises(ValueError):
        solar_position((2025, 1, 1, 12, 0, 0), lat, lon, 0.0, **kwargs)
    with pytest.raises(ValueError, match="invalid civil date/time"):
        solar_position((2025, 2, 30, 12, 0, 0), 45.0, 5.0)


def test_public_wrapper_resolves_solar_time_and_reports_source(monkeypatch):
    """reference tests/test_hybrid_terrain_pt.py:860-925 with the native seam stubbed out the same way."""
    import forge3d_amd.path_tracing as pt

    captured = {}

    class Native:
        @staticmethod
        def hybrid_render_terrain_reference(*args, **kwargs):
            captured.update(kwargs)
            return {}

    monkeypatch.setattr(pt, "_NATIVE", Native())
    when = SolarTime(utc=(2003, 10, 17, 12, 30, 30), observer_lat=39.742476, observer_lon=-105.1786, observer_elev_m=1830.14,
                     tz_offset_hours=-7.0, delta_t_seconds=67.0, pressure_mbar=820.0, temperature_c=11.0)
    dem = np.zeros((2, 2), np.float32)
    result = pt.hybrid_render_terrain_reference(dem, 2, 2, solar_time=when, min_frames=1, max_frames=1)
    expected = when.position()
    assert captured["sun_azimuth_deg"] == pytest.approx(expected["azimuth_deg"])
    assert captured["sun_elevation_deg"] == pytest.approx(expected["apparent_elevation_deg"])
    assert "sun_source" not in captured and result["sun_source"] == "solar_time"
    assert captured["observer_latitude_deg"] == 39.742476 and captured["pressure_mbar"] == 820.0
    pt.hybrid_render_terrain_reference(dem, 2, 2, solar_time=when, refraction_model="none", min_frames=1, max_frames=1)
    assert captured["sun_elevation_deg"] == pytest.approx(expected["true_elevation_deg"])
    # a mapping works too (reference geo.py:92-96)
    pt.hybrid_render_terrain_reference(dem, 2, 2, solar_time=when.to_native(), min_frames=1, max_frames=1)
    assert captured["sun_azimuth_deg"] == pytest.approx(expected["azimuth_deg"])
    for extra in ({"sun_azimuth_deg": 123.0}, {"pressure_mbar": 900.0}):
        with pytest.raises(ValueError, match="cannot be combined"):
            pt.hybrid_render_terrain_reference(dem, 2, 2, solar_time=when, min_frames=1, max_frames=1, **extra)
