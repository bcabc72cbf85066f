#!/usr/bin/env python
"""Golden vectors for the AETHER post from the REFERENCE'S OWN independent spectral oracle.

`/root/reference/tests/_aether_pt_oracle.py` is pure Python + NumPy and "deliberately imports no forge3d production
module" (its docstring): spectral constants, spherical geometry and a 64-step quadrature of single scattering + ground
bounce.  It can therefore be RUN in the build container (it cannot travel: no reference source is copied).  This script
imports it and writes inputs -> outputs as data:

  * segment transmittance  T_rgb(altitude, mu, distance, turbidity, ozone)  =
        _spectral_to_linear_rgb(_transmittance(_optical_columns(altitude, mu, distance, 64, ozone), turbidity))
    -- the law `aether_eval_segment_transmittance` (evaluation_core.wgsl:238-344) samples with 16 points;
  * sky radiance           L_rgb(view, sun elevation, altitude, turbidity)  = independent_reference_radiance(...)
    -- single scattering + ground bounce ("rgb"), and the same call with the module's GROUND_ALBEDO set to 0
    ("rgb_single": single scattering alone = what the SINGLE-scattering table of the LUT bank tabulates).

Run:  python tests/golden/make_aether_independent_vectors.py   (needs /root/reference; writes
tests/golden/atmosphere/independent_oracle_vectors.json)."""
import json
import math
import sys
from pathlib import Path

import numpy as np

REF_TESTS = Path("/root/reference/tests")
sys.path.insert(0, str(REF_TESTS))
import _aether_pt_oracle as ref  # noqa: E402  (the reference's test-owned oracle; run here, never shipped)

out = {"source": "reference tests/_aether_pt_oracle.py (pure Python, run in the build container)", "transmittance": [], "sky": []}
for turbidity in (2.0, 10.0):
    for altitude in (0.0, 1500.0, 35000.0):
        for mu in (-0.35, -0.05, 0.0, 0.2, 0.9):
            for distance in (1.0e3, 1.0e4, 5.0e4, 1.5e5):
                cols = ref._optical_columns(altitude, mu, distance, 64, 300.0)
                rgb = ref._spectral_to_linear_rgb(ref._transmittance(cols, turbidity))
                out["transmittance"].append({"turbidity": turbidity, "ozone_du": 300.0, "altitude_m": altitude, "mu": mu,
                                             "distance_m": distance, "rgb": [float(v) for v in rgb]})
for turbidity in (2.0, 10.0):
    for sun_el in (5.0, 10.0, 30.0, 60.0):
        for altitude in (1.0, 2000.0, 35000.0):
            for view_el in (2.0, 10.0, 30.0, 60.0, 85.0):
                for rel_az in (0.0, 90.0, 180.0):
                    el, az = math.radians(view_el), math.radians(90.0 + rel_az)  # the oracle's sun sits at azimuth 90
                    view = np.array([math.cos(el) * math.cos(az), math.sin(el), math.cos(el) * math.sin(az)])
                    rgb = ref.independent_reference_radiance(view, sun_el, turbidity=turbidity, ozone_du=300.0, mie_g=0.8,
                                                             observer_altitude_m=altitude, sun_azimuth_deg=90.0)
                    albedo = ref.GROUND_ALBEDO
                    ref.GROUND_ALBEDO = 0.0
                    single = ref.independent_reference_radiance(view, sun_el, turbidity=turbidity, ozone_du=300.0, mie_g=0.8,
                                                                observer_altitude_m=altitude, sun_azimuth_deg=90.0)
                    ref.GROUND_ALBEDO = albedo
                    out["sky"].append({"rgb_single": [float(v) for v in single], "turbidity": turbidity, "sun_elevation_deg": sun_el, "altitude_m": altitude,
                                       "view_elevation_deg": view_el, "relative_azimuth_deg": rel_az, "sun_azimuth_deg": 90.0,
                                       "view": [float(v) for v in view], "rgb": [float(v) for v in rgb]})
# ---- the flat-terrain fixture of the reference's two quantitative terrain gates (tests/test_atmosphere_reference.py:405-500:
# Z-up camera at radius 40 km, phi 180, theta 70, fov 52, 64 x 64 over a 60 km plane at z = 0, sun azimuth 90 / elevation 10):
# for hit pixels at the 10th ... 90th percentile of distance, the independent oracle's segment transmittance and its single
# scattering along the ray down to the ground ("aerial": what a surface colour carried over that segment becomes)
def fixture_hits(size=64, radius=40_000.0, span=60_000.0):
    phi, theta = math.radians(180.0), math.radians(70.0)
    eye = np.array([radius * math.sin(theta) * math.cos(phi), radius * math.sin(theta) * math.sin(phi), radius * math.cos(theta)])
    forward = -eye / np.linalg.norm(eye)
    right = np.cross(forward, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, forward)
    half = math.tan(math.radians(52.0) * 0.5)
    xs = (np.arange(size) + 0.5) / size * 2.0 - 1.0
    ys = 1.0 - (np.arange(size) + 0.5) / size * 2.0
    xx, yy = np.meshgrid(xs, ys)
    rays = forward + xx[..., None] * half * right + yy[..., None] * half * up
    rays /= np.linalg.norm(rays, axis=-1, keepdims=True)
    dist = np.full(rays.shape[:2], np.nan)
    down = rays[..., 2] < -1.0e-8
    dist[down] = -eye[2] / rays[..., 2][down]
    at = eye + rays * dist[..., None]
    hit = np.isfinite(dist) & (dist > 0.0) & (np.abs(at[..., 0]) < span * 0.49) & (np.abs(at[..., 1]) < span * 0.49)
    return eye, rays, dist, hit


eye, rays, dist, hit = fixture_hits()
idx = np.argwhere(hit)
order = np.argsort(dist[hit])
out["aerial"] = {"eye_altitude_m": float(eye[2]), "hit_count": int(hit.sum()), "sun_elevation_deg": 10.0, "sun_azimuth_deg": 90.0, "turbidity": 2.0,
                 "ozone_du": 300.0, "mie_g": 0.8, "cases": []}
for tenth in range(1, 10):
    y, x = idx[order[tenth * len(order) // 10]]
    ray = rays[y, x]
    ray_yup = np.array([ray[0], ray[2], ray[1]])
    distance = float(dist[y, x])
    cols = ref._optical_columns(float(eye[2]), float(ray_yup[1]), distance, 64, 300.0)
    t_rgb = ref._spectral_to_linear_rgb(ref._transmittance(cols, 2.0))
    albedo = ref.GROUND_ALBEDO
    ref.GROUND_ALBEDO = 0.0
    s_rgb = ref.independent_reference_radiance(ray_yup, 10.0, turbidity=2.0, ozone_du=300.0, mie_g=0.8, observer_altitude_m=float(eye[2]),
                                               sun_azimuth_deg=90.0)
    ref.GROUND_ALBEDO = albedo
    out["aerial"]["cases"].append({"percentile": 10 * tenth, "pixel": [int(x), int(y)], "distance_m": distance, "view": [float(v) for v in ray_yup],
                                   "ground_distance_m": float(ref._distance_to_boundary(float(eye[2]), float(ray_yup[1]))),
                                   "transmittance_rgb": [float(v) for v in t_rgb], "inscatter_single_rgb": [float(v) for v in s_rgb]})
dst = Path(__file__).resolve().parent / "atmosphere" / "independent_oracle_vectors.json"
dst.write_text(json.dumps(out, indent=0))
print(f"wrote {len(out['transmittance'])} transmittance, {len(out['sky'])} sky and {len(out['aerial']['cases'])} aerial vectors to {dst}")
