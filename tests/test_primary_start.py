"""Primary-ray certificates (csrc/f3d_cone.h): the cone of a pixel is marched once, its camera rays start where the cone
stops being clear of the terrain.  The certificate must be conservative (no ray of the pixel meets terrain before
t_clear) and the render with certificates must be the render without them, bit for bit."""
from __future__ import annotations

import os

import numpy as np
import pytest

import scenes
from emul import emul
from oracle import oracle


def _render_both(dem, size, cam, kw, **extra):
    outs = []
    for off in ("1", None):
        for name in ("F3D_EMUL_NO_PRIMARY_START", "F3D_EMUL_NO_SUN_CLEAR", "F3D_EMUL_NO_IBL_STOP"):
            if off:
                os.environ[name] = off
            else:
                os.environ.pop(name, None)
        try:
            outs.append(emul.render(dem, size[0], size[1], cam, **kw, **extra))
        finally:
            os.environ.pop("F3D_EMUL_NO_PRIMARY_START", None)
            os.environ.pop("F3D_EMUL_NO_SUN_CLEAR", None)
            os.environ.pop("F3D_EMUL_NO_IBL_STOP", None)
    return outs


def _cliff_dem(n=41):
    y, x = np.mgrid[0:n, 0:n].astype(np.float32)
    dem = np.where(x > n // 2, 6.0, 0.5).astype(np.float32)  # a wall across the middle
    dem += 0.3 * np.sin(0.7 * x) * np.cos(0.5 * y)
    dem[5:9, 5:9] += 9.0  # and a tower
    return dem.astype(np.float32)


CAMERAS = [
    ("above, outside", {"origin": (60.0, 45.0, 70.0), "look_at": (0.0, 2.0, 0.0)}),
    ("low, inside, facing the wall", {"origin": (-12.0, 1.6, 3.0), "look_at": (15.0, 2.0, 0.0)}),
    ("outside and below the top", {"origin": (-45.0, 3.0, -2.0), "look_at": (0.0, 3.0, 0.0)}),
    ("along the footprint's edge", {"origin": (-20.0, 4.0, 26.0), "look_at": (20.0, 1.0, 19.5)}),
    ("straight down", {"origin": (0.5, 60.0, 0.25), "look_at": (0.5, 0.0, 0.0)}),
    ("up at the sky", {"origin": (-15.0, 2.0, 0.0), "look_at": (-5.0, 30.0, 2.0)}),
    ("grazing over the tower", {"origin": (-19.5, 9.6, -19.5), "look_at": (15.0, 9.0, 15.0)}),
]


@pytest.mark.parametrize("name,cam", CAMERAS, ids=[c[0] for c in CAMERAS])
@pytest.mark.parametrize("fov,size", [(35.0, (57, 41)), (110.0, (9, 7))])
def test_renders_with_and_without_certificates_are_the_same_bits(name, cam, fov, size):
    dem = _cliff_dem()
    cam = {**cam, "up": (0.0, 1.0, 0.0), "fov_y": fov, "exposure": 1.0}
    kw = dict(spacing=(1.0, 1.0), exaggeration=1.0, sun_azimuth_deg=200.0, sun_elevation_deg=30.0, spp=4, max_frames=3, min_frames=3,
              variance_threshold=1e30, earth_model="flat", refraction_model="none", seed=11)
    try:
        a, b = _render_both(dem, size, cam, kw)
    except RuntimeError as exc:  # (a camera that sees no lit terrain ends with the reference's render error, both ways)
        assert "reservoir" in str(exc)
        return
    for key in ("rgba", "albedo", "normal", "depth", "accum", "m2", "res"):
        assert np.array_equal(a[key], b[key], equal_nan=True), (name, key)
    want = oracle.render(dem, size[0], size[1], cam, **kw)
    for key in ("rgba", "albedo", "normal", "depth"):
        assert np.array_equal(np.asarray(b[key]).reshape(np.asarray(want[key]).shape), want[key], equal_nan=True), (name, key)


def test_certificates_are_conservative_and_do_something():
    """Rays through the corners and the centre of a pixel's jitter square never meet the terrain before the pixel's
    t_clear (the oracle's closest hit says where they do), and on the golden scene most terrain pixels get one."""
    dem = scenes.golden_dem()
    kw = scenes.scene_kwargs(dem)
    W, H = 96, 64
    cam = scenes.CAM
    pixels = [(x, y) for y in range(1, H, 5) for x in range(1, W, 7)]
    starts = emul.primary_start(dem, W, H, cam, pixels, **{k: kw[k] for k in ("spacing", "exaggeration")})
    origin = np.array(cam["origin"], np.float64)
    fwd = np.array(cam["look_at"], np.float64) - origin
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.array(cam["up"], np.float64))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    half_h = np.tan(np.radians(cam["fov_y"]) / 2)
    half_w = half_h * W / H
    rays, owner = [], []
    for i, (gx, gy) in enumerate(pixels):
        for jx, jy in ((0, 0), (-0.5, -0.5), (0.5, -0.5), (-0.5, 0.5), (0.5, 0.5)):
            v = np.array([(((gx + 0.5 + jx) / W) * 2 - 1) * half_w, ((1 - (gy + 0.5 + jy) / H) * 2 - 1) * half_h, -1.0])
            v /= np.linalg.norm(v)
            d = v[0] * right + v[1] * up - v[2] * fwd
            d /= np.linalg.norm(d)
            rays.append([*origin, 1e-3, *d, 1e30])
            owner.append(i)
    sx, sz = kw["spacing"]
    hit = oracle.terrain_trace_batch(dem, np.array(rays, np.float32), origin=(-0.5 * (dem.shape[1] - 1) * sx, -0.5 * (dem.shape[0] - 1) * sz),
                                     spacing=kw["spacing"], exaggeration=kw["exaggeration"], any_hit=False, apply_curvature=False)
    partial = 0
    for k, i in enumerate(owner):
        t_clear = starts[i][0]
        if hit["hit"][k]:
            assert t_clear < 1e37 and hit["t"][k] >= t_clear, (pixels[i], t_clear, hit["t"][k])
    for (t_clear, level), (gx, gy) in zip(starts, pixels):
        partial += 0.0 < t_clear < 1e37
    assert partial >= len(pixels) // 3, (partial, len(pixels))


def test_sun_certificates_are_conservative_and_do_something():
    """Sun rays from points around the centre sample's origin (within the radius the certificate allows for) meet no
    terrain beyond the pixel's clear_from: the oracle's closest hits along them say where they do."""
    dem = _cliff_dem()
    W, H = 64, 48
    rng = np.random.default_rng(5)
    total = certified = 0
    for (name, cam), (az, el) in zip(CAMERAS[:5], ((200.0, 30.0), (90.0, 6.0), (178.0, 4.3), (300.0, 55.0), (45.0, 15.0))):
        cam = {**cam, "up": (0.0, 1.0, 0.0), "fov_y": 40.0, "exposure": 1.0}
        kw = dict(spacing=(1.0, 1.0), exaggeration=1.0, sun_azimuth_deg=az, sun_elevation_deg=el, earth_model="flat", refraction_model="none")
        pixels = [(x, y) for y in range(2, H, 6) for x in range(2, W, 7)]
        recs = emul.sun_clear(dem, W, H, cam, pixels, **kw)
        half_h = np.tan(np.radians(cam["fov_y"]) / 2)
        plane = np.hypot(half_h * W / H / W, half_h / H)
        delta = 1.01 * plane * (1 + plane * plane)
        rays, owner = [], []
        for i, rec in enumerate(recs):
            if rec["depth"] == 0.0:
                continue
            total += 1
            if rec["clear_from"] > 1e37:
                continue
            certified += 1
            slack = 1.0 * rec["depth"] * delta + 0.5
            rho = slack + (rec["depth"] + slack) * delta
            for _ in range(24):
                off = rng.normal(size=3)
                off *= rng.uniform(0.0, rho) / np.linalg.norm(off)
                rays.append([*(np.array(rec["origin"]) + off), 1e-3, *rec["wi"], 1e30])
                owner.append(i)
        if not rays:
            continue
        n = dem.shape[0]
        hit = oracle.terrain_trace_batch(dem, np.array(rays, np.float32), origin=(-0.5 * (n - 1), -0.5 * (n - 1)), any_hit=False,
                                         apply_curvature=False)
        for k, i in enumerate(owner):
            if hit["hit"][k]:
                assert hit["t"][k] <= recs[i]["clear_from"] + 1e-3, (name, pixels[i], recs[i], float(hit["t"][k]))
    assert certified >= 10, (certified, total)  # (a small footprint and low suns: most cylinders leave it below the top)


def _surface_height(dem, spacing, exaggeration, x, z):
    """Bilinear terrain height at world (x, z) (DEM centred on the origin, row = +z)."""
    h, w = dem.shape
    u = (x + 0.5 * (w - 1) * spacing[0]) / spacing[0]
    v = (z + 0.5 * (h - 1) * spacing[1]) / spacing[1]
    i, j = int(np.clip(np.floor(u), 0, w - 2)), int(np.clip(np.floor(v), 0, h - 2))
    fu, fv = u - i, v - j
    c = dem[j:j + 2, i:i + 2].astype(np.float64) * exaggeration
    return (c[0, 0] * (1 - fu) + c[0, 1] * fu) * (1 - fv) + (c[1, 0] * (1 - fu) + c[1, 1] * fu) * fv


@pytest.mark.parametrize("case", ["golden", "cliff", "ragged"])
def test_ibl_certificates_are_conservative_and_do_something(case):
    """The far-horizon table (f3d_cone.h): a ray that starts ANYWHERE on a block's surface (lifted 1e-3 like the IBL rays)
    and climbs more steeply than the block's horizon of its sector meets no terrain beyond the stop distance -- the
    oracle's closest hit along it says where it does."""
    if case == "golden":
        dem = scenes.golden_dem()
        kw = scenes.scene_kwargs(dem)
        geo = {k: kw[k] for k in ("spacing", "exaggeration")}
    elif case == "cliff":
        dem, geo = _cliff_dem(), dict(spacing=(1.0, 1.0), exaggeration=1.0)
    else:
        rng0 = np.random.default_rng(4)
        dem = (np.cumsum(rng0.normal(size=(37, 53)), axis=1) * 3.0 + 40.0 * rng0.random((37, 53))).astype(np.float32)
        geo = dict(spacing=(2.5, 1.5), exaggeration=1.7)
    h, w = dem.shape
    rng = np.random.default_rng(9)
    cells = [(int(rng.integers(0, w - 1)), int(rng.integers(0, h - 1))) for _ in range(150)]
    recs = emul.horizon_blocks(dem, cells, **geo)
    sx, sz = geo["spacing"]
    rays, meta = [], []
    sectors_with_horizon = 0
    for (cx, cz), rec in zip(cells, recs):
        sectors_with_horizon += sum(1 for hz in rec["far"] if hz < 1e30)
        side = 1 << rec["level"]
        bx0, bz0 = (cx >> rec["level"]) * side, (cz >> rec["level"]) * side
        for _ in range(40):
            az = rng.uniform(0, 2 * np.pi)
            dx, dz = np.cos(az), np.sin(az)
            sector = (1 if dx < 0 else 0) | (2 if dz < 0 else 0) | (4 if abs(dz) > abs(dx) else 0)
            horizon = rec["far"][sector]
            if not horizon < 1e30:
                continue
            slope = max(horizon, 0.0) * 1.001 + 2e-4 + rng.exponential(0.3)
            d = np.array([dx, slope, dz]) / np.sqrt(1 + slope * slope)
            # an origin on the block's surface, lifted along a random unit vector like the 1e-3 normal offset
            u = rng.uniform(bx0, min(bx0 + side, w - 1)), rng.uniform(bz0, min(bz0 + side, h - 1))
            x, z = -0.5 * (w - 1) * sx + u[0] * sx, -0.5 * (h - 1) * sz + u[1] * sz
            y = _surface_height(dem, (sx, sz), geo["exaggeration"], x, z)
            lift = rng.normal(size=3)
            lift *= 1e-3 / np.linalg.norm(lift)
            o = np.array([x, y, z]) + lift
            assert np.hypot(o[0] - rec["centre"][0], o[2] - rec["centre"][1]) <= rec["rho"] and o[1] >= rec["y_lo"]
            rays.append([*o, 1e-3, *d, 1e30])
            meta.append(rec["stop_distance"] / np.hypot(d[0], d[2]))
    hit = oracle.terrain_trace_batch(dem, np.array(rays, np.float32), origin=(-0.5 * (w - 1) * sx, -0.5 * (h - 1) * sz),
                                     spacing=geo["spacing"], exaggeration=geo["exaggeration"], any_hit=False, apply_curvature=False)
    for k, t_stop in enumerate(meta):
        if hit["hit"][k]:
            assert hit["t"][k] <= t_stop, (case, rays[k], float(hit["t"][k]), t_stop)
    assert sectors_with_horizon > 2 * len(recs) and len(rays) > 500


@pytest.mark.gpu
def test_gpu_certificates_do_not_change_the_image():
    """The device with certificates == the oracle, on a camera inside the footprint facing a wall and on the golden scene."""
    import forge3d_amd as f3d

    dem = _cliff_dem()
    kw = dict(spacing=(1.0, 1.0), exaggeration=1.0, sun_azimuth_deg=200.0, sun_elevation_deg=30.0, spp=8, max_frames=4, min_frames=4,
              variance_threshold=1e30, earth_model="flat", refraction_model="none", seed=11)
    for name, cam in CAMERAS:
        cam = {**cam, "up": (0.0, 1.0, 0.0), "fov_y": 40.0, "exposure": 1.0}
        try:
            want = oracle.render(dem, 160, 96, cam, **kw)
        except RuntimeError:
            with pytest.raises(RuntimeError):
                f3d.hybrid_render_terrain_reference(dem, 160, 96, cam, **kw)
            continue
        got = f3d.hybrid_render_terrain_reference(dem, 160, 96, cam, **kw)
        for key in ("rgba", "albedo", "normal", "depth"):
            assert np.array_equal(got[key], want[key], equal_nan=True), (name, key)
