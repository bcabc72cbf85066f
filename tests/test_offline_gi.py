"""BASELINE.json configs[2] -- "atmosphere + GI" over a DEM -- as a composition of two pinned halves (SURVEY.md 8f rows 1
and 3; the reference itself has no such combination: its wavefront tracer has no terrain hook):

  radiance   the PBR path tracer with the DEM as its heightfield primitive   (oracle: wavefront_oracle.c on terrain_trace)
  post       the terrain tracer's resolve + AETHER aerial-perspective post   (oracle: f3d_oracle.c, accum_override hook)

`forge3d_amd.offline.render_terrain_gi` must equal the composition of the two oracles bit for bit."""
from __future__ import annotations

import numpy as np
import pytest

import scenes
from forge3d_amd import atmosphere as atm
from oracle import oracle, wavefront_oracle

from forge3d_amd.atmosphere import INSTALLED_BANK as BANK  # noqa: E402


def _case(turbidity=None):
    dem = scenes.golden_dem()
    kw = scenes.scene_kwargs(dem)
    geo = dict(spacing=kw["spacing"], exaggeration=kw["exaggeration"], albedo=(0.55, 0.52, 0.48), sun_azimuth_deg=225.0,
               sun_elevation_deg=30.0, sun_intensity=3.0)
    handle = None if turbidity is None else atm.AtmosphereLutHandle.load_shipped(atm.AtmosphereConfig(turbidity=turbidity), bank_dir=BANK)
    return dem, scenes.CAM, geo, handle


def _oracle_composition(dem, w, h, cam, geo, handle, spp, seed=7):
    """The two oracles chained exactly as render_terrain_gi chains the two device paths."""
    from forge3d_amd.wavefront import DirectionalLight, Sphere, Terrain, WavefrontScene

    az, el = np.deg2rad(np.float32(geo["sun_azimuth_deg"])), np.deg2rad(np.float32(geo["sun_elevation_deg"]))
    to_sun = (float(np.cos(az) * np.cos(el)), float(np.sin(el)), float(np.sin(az) * np.cos(el)))
    scene = WavefrontScene(
        terrain=Terrain(heights=dem, spacing=geo["spacing"], exaggeration=geo["exaggeration"], material_id=0),
        spheres=[Sphere(center=(0.0, -1.0e9, 0.0), radius=0.0, albedo=geo["albedo"], metallic=0.0, roughness=0.9)],
        dir_lights=[DirectionalLight(tuple(-c for c in to_sun), geo["sun_intensity"], (1.0, 0.97, 0.92), 1.0)],
        object_importance=[1.0], env_ground=(0.40, 0.48, 0.62), env_sky=(0.40, 0.48, 0.62), miss_ground=(0.35, 0.45, 0.70),
        miss_sky=(0.35, 0.45, 0.70), cam_origin=cam["origin"], cam_look_at=cam["look_at"], cam_up=cam["up"], fov_y_deg=cam["fov_y"],
        exposure=cam.get("exposure", 1.0), seed_hi=(0x9E3779B9 ^ seed) & 0xFFFFFFFF, seed_lo=0x85EBCA6B)
    gi = wavefront_oracle.render(scene.as_dict(), w, h, spp)
    sums = gi["accum"].copy()
    sums[..., 3] = float(spp)
    post = oracle.render(dem, w, h, cam, spp=1, max_frames=2, min_frames=2, variance_threshold=1e30, seed=seed, atmosphere=handle,
                         accum_override=sums, **geo)
    return gi, post


def test_accumulation_override_is_neutral_for_the_renders_own_sums():
    """The composition hook of the oracle: feeding a render its OWN accumulation back changes nothing."""
    dem, cam, geo, handle = _case(turbidity=2.0)
    k = dict(geo, spp=2, max_frames=3, min_frames=3, variance_threshold=1e30)
    a = oracle.render(dem, 64, 48, cam, atmosphere=handle, dump_state=True, **k)
    b = oracle.render(dem, 64, 48, cam, atmosphere=handle, accum_override=a["accum"].reshape(48, 64, 4), **k)
    assert np.array_equal(a["rgba"], b["rgba"])
    c = oracle.render(dem, 64, 48, cam, accum_override=a["accum"].reshape(48, 64, 4), **k)
    assert np.array_equal(c["rgba"], oracle.render(dem, 64, 48, cam, **k)["rgba"])


@pytest.mark.parametrize("turbidity", [None, 4.0])
def test_oracle_composition_transports_gi_radiance(turbidity):
    """Multi-bounce radiance differs from the one-bounce terrain tracer's, the post moves hit and sky pixels, AOVs stay."""
    dem, cam, geo, handle = _case(turbidity)
    gi, post = _oracle_composition(dem, 72, 48, cam, geo, handle, spp=8)
    plain = oracle.render(dem, 72, 48, cam, spp=1, max_frames=2, min_frames=2, variance_threshold=1e30, **geo)
    assert np.array_equal(post["depth"], plain["depth"], equal_nan=True) and np.array_equal(post["normal"], plain["normal"])
    hit = np.isfinite(post["depth"])
    assert hit.any() and (~hit).any()
    assert (post["rgba"][..., :3][hit] != plain["rgba"][..., :3][hit]).any()
    if turbidity is not None:
        _, bare = _oracle_composition(dem, 72, 48, cam, geo, None, spp=8)
        delta = np.abs(post["rgba"][..., :3].astype(int) - bare["rgba"][..., :3].astype(int)).max(-1)
        assert (delta[hit] > 0).mean() > 0.5 and (delta[~hit] > 0).mean() > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("turbidity,size,spp", [(None, (96, 64), 6), (2.0, (160, 96), 12), (10.0, (61, 47), 5)])
def test_render_terrain_gi_equals_the_composition_of_the_oracles(turbidity, size, spp):
    from forge3d_amd import offline

    dem, cam, geo, handle = _case(turbidity)
    w, h = size
    got = offline.render_terrain_gi(dem, w, h, cam, spp=spp, atmosphere=handle, **geo)
    gi, want = _oracle_composition(dem, w, h, cam, geo, handle, spp)
    assert np.array_equal(got["hdr"], gi["hdr"])
    for key in ("rgba", "albedo", "normal", "depth"):
        assert np.array_equal(got[key], want[key], equal_nan=True), key
    assert got["frames"] == spp and got["path_vertices"] > w * h * spp

@pytest.mark.gpu
def test_camera_rays_starting_at_the_terrain_certificates_change_nothing(monkeypatch):
    """`render_terrain_gi` hands the terrain tracer's primary-ray certificates (`f3d_session_primary_start`) to the PBR
    tracer, whose camera rays then begin their march where the certificate proved free space ends.  With the hand-over
    switched off the march starts at the box entry: both must produce the same hits, so the same bytes."""
    from forge3d_amd import offline

    dem, cam, geo, handle = _case(2.0)
    on = offline.render_terrain_gi(dem, 200, 120, cam, spp=5, atmosphere=handle, **geo)
    monkeypatch.setenv("F3D_GI_NO_PRIMARY_START", "1")
    off = offline.render_terrain_gi(dem, 200, 120, cam, spp=5, atmosphere=handle, **geo)
    for key in ("hdr", "rgba", "albedo", "normal", "depth"):
        assert np.array_equal(on[key], off[key], equal_nan=True), key
    assert on["path_vertices"] == off["path_vertices"]


@pytest.mark.gpu
def test_config3_gi_at_full_size_equals_the_composition_of_the_oracles():
    """BASELINE.json configs[2] exactly as bench.py times it (`configs.C3_gi`) -- the 2048^2 rainier-proxy DEM as the
    heightfield primitive of the PBR path tracer at 1920 x 1080, the AETHER post at turbidity 2 on its radiance -- 4 paths a
    pixel against the composition of the two CPU oracles (seconds on the GPU box's host cores), every pixel of every output.
    (Round 3 checked the GI leg at <= 160 x 96 only; the 1080p render existed as a bench number compared with nothing.)"""
    from forge3d_amd import datasets, offline

    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    geo = dict(spacing=kw["spacing"], exaggeration=kw["exaggeration"], albedo=kw.get("albedo", (0.6, 0.6, 0.6)),
               sun_azimuth_deg=kw["sun_azimuth_deg"], sun_elevation_deg=kw["sun_elevation_deg"], sun_intensity=kw["sun_intensity"])
    handle = atm.AtmosphereLutHandle.load_shipped(atm.AtmosphereConfig(turbidity=2.0), bank_dir=BANK)
    w, h, spp = 1920, 1080, 4
    got = offline.render_terrain_gi(dem, w, h, cam, spp=spp, atmosphere=handle, memory_budget_bytes=8 << 30, **geo)
    gi, want = _oracle_composition(dem, w, h, cam, geo, handle, spp)
    hit = np.isfinite(want["depth"])
    assert 0.3 < hit.mean() < 0.5  # the benched camera: ~40 % terrain, the rest sky
    assert got["path_vertices"] > w * h * spp * 1.2  # multi-bounce: more than one vertex a path on the terrain pixels
    assert np.array_equal(got["hdr"], gi["hdr"])
    for key in ("rgba", "albedo", "normal", "depth"):
        assert np.array_equal(got[key], want[key], equal_nan=True), key
