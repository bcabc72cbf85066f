"""A-trous denoiser.  The golden vectors in tests/golden/atrous_cases.npz were
written by the REFERENCE's own NumPy implementation (tests/golden/make_denoise_fixtures.py); the
oracle (oracle/denoise_oracle.py) must reproduce them bit for bit, the HIP kernel within 2e-5
absolute (expf / acosf differ from NumPy's by a few ulp)."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

TOL = 2e-5


def _cases():
    z = np.load(ROOT / "tests" / "golden" / "atrous_cases.npz")
    g = {k: z[k] for k in ("albedo", "normal", "depth")}
    cases = {
        "color_only_3": dict(iterations=3),
        "color_only_1_wide": dict(iterations=1, sigma_color=0.35),
        "all_guides_3": dict(**g, iterations=3),
        "all_guides_4_tight": dict(**g, iterations=4, sigma_color=0.05, sigma_albedo=0.1, sigma_normal=0.1,
                                   sigma_depth=0.2),
        "normal_depth_2": dict(normal=g["normal"], depth=g["depth"], iterations=2, sigma_normal=0.1, sigma_depth=0.1),
        "albedo_no_extra_term": dict(albedo=g["albedo"], iterations=2, sigma_albedo=0.0),
        "zero_iterations_means_one": dict(iterations=0),
    }
    return z["color"], {k: (kw, z["want_" + k]) for k, kw in cases.items()}


def test_oracle_reproduces_the_reference_vectors_bit_for_bit():
    from oracle import denoise_oracle

    color, cases = _cases()
    for name, (kw, want) in cases.items():
        got = denoise_oracle.atrous_denoise(color, **kw)
        assert got.dtype == np.float32 and np.array_equal(got, want), name


def test_oracle_has_the_reference_behaviour_its_tests_check():
    """reference tests/test_denoise_settings.py:154-250 (variance reduction, edge preservation)."""
    from oracle import denoise_oracle

    rng = np.random.default_rng(42)
    clean = np.ones((64, 64, 3), np.float32) * 0.5
    noisy = np.clip(clean + rng.normal(0, 0.15, clean.shape).astype(np.float32), 0, 1)
    assert np.var(denoise_oracle.atrous_denoise(noisy, iterations=3, sigma_color=0.15)) < np.var(noisy)
    image = np.zeros((64, 64, 3), np.float32)
    image[:, 32:, :] = 1.0
    noisy = np.clip(image + np.random.default_rng(42).normal(0, 0.05, image.shape).astype(np.float32), 0, 1)
    den = denoise_oracle.atrous_denoise(noisy, iterations=3, sigma_color=0.1)
    assert np.mean(den[:, 34:, :]) - np.mean(den[:, :30, :]) > 0.5


def test_wrapper_validates_like_the_reference():
    """Same ValueError messages as reference denoise.py:46-71, before any device work."""
    from forge3d_amd.denoise import atrous_denoise

    with pytest.raises(ValueError, match="color must be"):
        atrous_denoise(np.zeros((64, 64), np.float32))
    with pytest.raises(ValueError, match="color must be"):
        atrous_denoise(np.zeros((64, 64, 4), np.float32))
    color = np.zeros((64, 64, 3), np.float32)
    with pytest.raises(ValueError, match="albedo must match"):
        atrous_denoise(color, albedo=np.zeros((32, 32, 3), np.float32))
    with pytest.raises(ValueError, match="normal must match"):
        atrous_denoise(color, normal=np.zeros((32, 32, 3), np.float32))
    with pytest.raises(ValueError, match="depth must be"):
        atrous_denoise(color, depth=np.zeros((32, 32), np.float32))


@pytest.mark.gpu
def test_hip_denoiser_matches_the_reference_vectors():
    from forge3d_amd.denoise import atrous_denoise

    color, cases = _cases()
    for name, (kw, want) in cases.items():
        got = atrous_denoise(color, **kw)
        assert got.dtype == np.float32 and got.shape == want.shape
        assert float(np.abs(got - want).max()) <= TOL, (name, float(np.abs(got - want).max()))


@pytest.mark.gpu
def test_hip_denoiser_on_a_rendered_frame_matches_the_oracle():
    """The intended use: denoise a low-spp terrain render with its own AOVs (NaN depth on sky pixels
    is mapped to 0 by the caller, as forge3d's examples do)."""
    import forge3d_amd as f3d
    import scenes
    from forge3d_amd.denoise import atrous_denoise
    from oracle import denoise_oracle

    dem = scenes.golden_dem()
    out = f3d.hybrid_render_terrain_reference(dem, 160, 120, scenes.CAM,
                                              **scenes.fixed_frames(scenes.scene_kwargs(dem), 2, spp=1))
    color = out["rgba"][..., :3].astype(np.float32) / 255.0
    depth = np.nan_to_num(out["depth"], nan=0.0)
    kw = dict(albedo=out["albedo"], normal=out["normal"], depth=depth, iterations=3)
    got, want = atrous_denoise(color, **kw), denoise_oracle.atrous_denoise(color, **kw)
    assert float(np.abs(got - want).max()) <= TOL
    assert np.var(got[60:, :, :] - want[60:, :, :]) < 1e-9
    # 1080p sanity: finite, same shape (size-independent property: constant image stays constant inside)
    flat = np.full((1080, 1920, 3), 0.25, np.float32)
    den = atrous_denoise(flat, iterations=2)
    assert den.shape == flat.shape and np.allclose(den[8:-8, 8:-8], 0.25, atol=1e-6)
