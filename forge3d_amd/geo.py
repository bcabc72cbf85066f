"""Geodetic solar geometry: the ``solar_time=`` seam of the terrain path tracer's wrapper.

Mirrors reference python/forge3d/geo.py (``SolarTime``, ``SolarVector``, ``solar_position``, ``_coerce_solar_time``);
the ephemeris behind it is the NREL Solar Position Algorithm (Reda & Andreas 2003, NREL/TP-560-34302), which the
reference implements in Rust on the host (src/geo/solar.rs:77-200).  It is host-side f64 arithmetic there and here:
a few hundred cosine terms once per render, nothing for the GPU.  The periodic-term tables of the report's
Appendix A ship as data (``forge3d_amd/data/spa_terms.json``, written by tools/make_spa_tables.py).
Pinned by the official NREL rows the reference's own test holds (tests/golden/spa/spa_reference.csv).
"""
from __future__ import annotations

import json
import math
from collections.abc import Mapping
from dataclasses import asdict, dataclass
from functools import lru_cache
from pathlib import Path
from typing import Dict, Optional, Tuple, Union

UtcTuple = Tuple[int, int, int, int, int, Union[int, float]]
SolarVector = Dict[str, float]

_ABERRATION_ARCSEC = -20.4898
_EARTH_FLATTENING = 0.99664719
_EARTH_RADIUS_M = 6_378_140.0


@lru_cache(maxsize=1)
def _tables():
    return json.loads((Path(__file__).resolve().parent / "data" / "spa_terms.json").read_text())


def _poly(coefficients, x: float) -> float:
    value = 0.0
    for c in reversed(coefficients):
        value = value * x + c
    return value


def _series(jme: float, term_sets) -> float:
    """Sum_i JME^i * Sum_k A cos(B + C JME), / 1e8 (report eq. 9-12)."""
    sums = [sum(a * math.cos(b + c * jme) for a, b, c in terms) for terms in term_sets]
    return _poly(sums, jme) / 1e8


def _validate(year, month, day, hour, minute, second, lat, lon, elev_m, tz, delta_t, pressure, temperature):
    """SolarTime::validate, src/geo/solar.rs:47-74 (same messages)."""
    def days_in_month(y, m):
        if m == 2:
            return 29 if (y % 4 == 0 and (y % 100 != 0 or y % 400 == 0)) else 28
        return 30 if m in (4, 6, 9, 11) else 31

    if not -2000 <= year <= 6000:
        raise ValueError("year must be in [-2000, 6000]")
    if (not 1 <= month <= 12 or day == 0 or day > days_in_month(year, month) or not 0 <= hour <= 23 or not 0 <= minute <= 59
            or not math.isfinite(second) or not 0.0 <= second < 60.0):
        raise ValueError("invalid civil date/time")
    for name, value, lo, hi in (("latitude", lat, -90.0, 90.0), ("longitude", lon, -180.0, 180.0), ("tz_offset_hours", tz, -18.0, 18.0)):
        if not (math.isfinite(value) and lo <= value <= hi):
            raise ValueError(f"{name} must be finite and in [{lo:g}, {hi:g}]")
    if not math.isfinite(delta_t) or not math.isfinite(elev_m):
        raise ValueError("delta_t_seconds and elevation_m must be finite")
    if not math.isfinite(pressure) or pressure <= 0.0:
        raise ValueError("pressure_mbar must be finite and positive")
    if not math.isfinite(temperature) or temperature <= -273.15:
        raise ValueError("temperature_c must be above absolute zero")


def solar_position(utc: UtcTuple, lat: float, lon: float, elev_m: float = 0.0, *, tz_offset_hours: float = 0.0,
                   delta_t_seconds: float = 69.0, pressure_mbar: float = 1013.25, temperature_c: float = 15.0) -> SolarVector:
    """Return the NREL-SPA solar vector.  ``utc`` is a timezone-free civil tuple; the caller supplies its UTC offset
    and delta-T explicitly (reference python/forge3d/geo.py:23-52)."""
    if len(utc) != 6:
        raise ValueError("utc must be (year, month, day, hour, minute, second)")
    year, month, day, hour, minute = (int(v) for v in utc[:5])
    second = float(utc[5])
    lat, lon, elev_m = float(lat), float(lon), float(elev_m)
    tz, delta_t, pressure, temperature = float(tz_offset_hours), float(delta_t_seconds), float(pressure_mbar), float(temperature_c)
    _validate(year, month, day, hour, minute, second, lat, lon, elev_m, tz, delta_t, pressure, temperature)
    T = _tables()

    # Julian day / ephemeris day, century, millennium (report 3.1)
    y, m = (year - 1, month + 12) if month <= 2 else (year, month)
    dayf = day + (hour + minute / 60.0 + second / 3600.0 - tz) / 24.0
    a = math.floor(y / 100.0)
    b = 2.0 - a + math.floor(a / 4.0)
    jd = math.floor(365.25 * (y + 4716.0)) + math.floor(30.6001 * (m + 1.0)) + dayf + b - 1524.5
    jde = jd + delta_t / 86400.0
    jc = (jd - 2451545.0) / 36525.0
    jce = (jde - 2451545.0) / 36525.0
    jme = jce / 10.0

    # Earth heliocentric longitude, latitude, radius (3.2); geocentric (3.3)
    L = math.degrees(_series(jme, T["TERMS_L"])) % 360.0
    B = math.degrees(_series(jme, T["TERMS_B"])) % 360.0
    R = _series(jme, T["TERMS_R"])
    theta = (L + 180.0) % 360.0
    beta = -B

    # nutation (3.4), obliquity (3.5), aberration (3.6), apparent sun longitude (3.7), sidereal time (3.8)
    x = [_poly(c, jce) for c in T["NUTATION_COEFFS"]]
    psi = eps = 0.0
    for yrow, pe in zip(T["TERMS_Y"], T["TERMS_PE"]):
        arg = math.radians(sum(xv * yv for xv, yv in zip(x, yrow)))
        psi += (pe[0] + pe[1] * jce) * math.sin(arg)
        eps += (pe[2] + pe[3] * jce) * math.cos(arg)
    delta_psi, delta_eps = psi / 36_000_000.0, eps / 36_000_000.0
    epsilon = _poly(T["OBLIQUITY_COEFFS"], jme / 10.0) / 3600.0 + delta_eps
    lam = theta + delta_psi + _ABERRATION_ARCSEC / (3600.0 * R)
    nu0 = (280.46061837 + 360.98564736629 * (jd - 2451545.0) + jc * jc * (0.000387933 - jc / 38710000.0)) % 360.0
    nu = nu0 + delta_psi * math.cos(math.radians(epsilon))

    # geocentric right ascension / declination (3.9, 3.10)
    beta_r, eps_r, lam_r = math.radians(beta), math.radians(epsilon), math.radians(lam)
    alpha = math.degrees(math.atan2(math.sin(lam_r) * math.cos(eps_r) - math.tan(beta_r) * math.sin(eps_r), math.cos(lam_r))) % 360.0
    delta = math.degrees(math.asin(math.sin(beta_r) * math.cos(eps_r) + math.cos(beta_r) * math.sin(eps_r) * math.sin(lam_r)))

    # local hour angle (3.11), topocentric place (3.12, 3.13), zenith / azimuth (3.14, 3.15)
    h = (nu + lon - alpha) % 360.0
    xi = math.radians(8.794 / (3600.0 * R))
    phi, delta_r, h_r = math.radians(lat), math.radians(delta), math.radians(h)
    u = math.atan(_EARTH_FLATTENING * math.tan(phi))
    oy = _EARTH_FLATTENING * math.sin(u) + (elev_m / _EARTH_RADIUS_M) * math.sin(phi)
    ox = math.cos(u) + (elev_m / _EARTH_RADIUS_M) * math.cos(phi)
    denom = math.cos(delta_r) - ox * math.sin(xi) * math.cos(h_r)
    delta_alpha = math.degrees(math.atan2(-ox * math.sin(xi) * math.sin(h_r), denom))
    delta_prime = math.atan2((math.sin(delta_r) - oy * math.sin(xi)) * math.cos(math.radians(delta_alpha)), denom)
    h_prime = math.radians(h - delta_alpha)
    cz = math.sin(phi) * math.sin(delta_prime) + math.cos(phi) * math.cos(delta_prime) * math.cos(h_prime)
    true_elevation = 90.0 - math.degrees(math.acos(min(1.0, max(-1.0, cz))))
    if true_elevation >= -0.83337:
        refraction = ((pressure / 1010.0) * (283.0 / (273.0 + temperature)) * 1.02
                      / (60.0 * math.tan(math.radians(true_elevation + 10.3 / (true_elevation + 5.11)))))
    else:
        refraction = 0.0
    apparent = true_elevation + refraction
    azimuth = (180.0 + math.degrees(math.atan2(math.sin(h_prime), math.cos(h_prime) * math.sin(phi) - math.tan(delta_prime) * math.cos(phi)))) % 360.0

    # equation of time (A.1)
    mean_longitude = (280.4664567 + jme * (360007.6982779 + jme * (0.03032028 + jme * (1.0 / 49931.0 + jme * (-1.0 / 15300.0 - jme / 2000000.0))))) % 360.0
    eot = (4.0 * (mean_longitude - 0.0057183 - alpha + delta_psi * math.cos(eps_r))) % 1440.0
    if eot > 20.0:
        eot -= 1440.0
    return {"zenith_deg": 90.0 - apparent, "azimuth_deg": azimuth, "apparent_elevation_deg": apparent,
            "true_elevation_deg": true_elevation, "distance_au": R, "equation_of_time_min": eot}


@dataclass(frozen=True)
class SolarTime:
    """reference python/forge3d/geo.py:55-86."""
    utc: UtcTuple
    observer_lat: float
    observer_lon: float
    observer_elev_m: float = 0.0
    tz_offset_hours: float = 0.0
    delta_t_seconds: float = 69.0
    pressure_mbar: float = 1013.25
    temperature_c: float = 15.0
    delta_t: Optional[float] = None

    def position(self) -> SolarVector:
        return solar_position(self.utc, self.observer_lat, self.observer_lon, self.observer_elev_m,
                              tz_offset_hours=self.tz_offset_hours,
                              delta_t_seconds=(self.delta_t_seconds if self.delta_t is None else self.delta_t),
                              pressure_mbar=self.pressure_mbar, temperature_c=self.temperature_c)

    def to_native(self) -> Dict[str, object]:
        payload = asdict(self)
        payload["delta_t_seconds"] = self.delta_t_seconds if self.delta_t is None else self.delta_t
        payload.pop("delta_t")
        return payload


def _coerce_solar_time(value) -> SolarTime:
    """reference python/forge3d/geo.py:89-97; additionally, any object with the reference SolarTime's duck type
    (``position()`` + observer / weather attributes) is accepted as is."""
    if isinstance(value, SolarTime):
        return value
    if isinstance(value, Mapping):
        fields = dict(value)
        if "delta_t" in fields and "delta_t_seconds" not in fields:
            fields["delta_t_seconds"] = fields.pop("delta_t")
        return SolarTime(**fields)
    if callable(getattr(value, "position", None)) and all(hasattr(value, a) for a in ("observer_lat", "observer_lon", "pressure_mbar", "temperature_c")):
        return value
    raise TypeError("solar_time must be forge3d.geo.SolarTime or a mapping")


def resolve_solar_time(solar_time) -> dict:
    """What the wrapper needs from ``solar_time=`` (reference path_tracing.py:1018-1032)."""
    when = _coerce_solar_time(solar_time)
    solar = dict(when.position())
    for key in ("azimuth_deg", "true_elevation_deg", "apparent_elevation_deg"):
        if key not in solar:
            raise RuntimeError(f"solar_time.position() did not report {key!r}")
    solar["observer_lat"] = when.observer_lat
    solar["observer_lon"] = when.observer_lon
    solar["pressure_mbar"] = when.pressure_mbar
    solar["temperature_c"] = when.temperature_c
    return solar


__all__ = ["SolarTime", "SolarVector", "solar_position"]
