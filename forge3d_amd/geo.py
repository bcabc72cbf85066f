"""SolarTime seam of the terrain path tracer wrapper.

The reference resolves ``solar_time=`` through its native NREL-SPA implementation
(reference python/forge3d/geo.py:23-52 -> src/geo/solar.rs:77).  That ephemeris is an
adjacent feature, not part of the path-tracing hot path: objects that can
already report their own position (``.position()`` returning azimuth / true / apparent
elevation, as the reference's SolarTime does) are accepted; anything else raises.
"""
from __future__ import annotations


def resolve_solar_time(solar_time) -> dict:
    position = getattr(solar_time, "position", None)
    if not callable(position):
        raise RuntimeError(
            "forge3d_amd: solar_time= needs an object with a .position() method (the NREL SPA "
            "ephemeris of the reference is outside the MI355X terrain-PT path); pass manual "
            "sun_azimuth_deg / sun_elevation_deg instead"
        )
    solar = dict(position())
    for key in ("azimuth_deg", "true_elevation_deg", "apparent_elevation_deg"):
        if key not in solar:
            raise RuntimeError(f"solar_time.position() did not report {key!r}")
    solar["observer_lat"] = getattr(solar_time, "observer_lat")
    solar["observer_lon"] = getattr(solar_time, "observer_lon")
    solar["pressure_mbar"] = getattr(solar_time, "pressure_mbar")
    solar["temperature_c"] = getattr(solar_time, "temperature_c")
    return solar
