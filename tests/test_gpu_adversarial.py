"""`-m gpu`: the adversarial lattice-aligned sets of tests/test_adversarial_march.py on the device --
64-lane waves, the wave-voted leaf drains and the ray sharing (dealt slices, verdict board), incl. the
sharing threshold forced to 64 lanes so that EVERY any-hit ray is dealt from its first step."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

import scenes
from test_adversarial_march import DEMS, check_ray_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def f3d():
    import forge3d_amd
    from forge3d_amd import _native

    assert _native.device_count() >= 1, "no HIP device: the GPU tests must run on the MI355X box"
    return forge3d_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o

    o.build()
    return o


def device_trace(dem, rays, *, origin, spacing, inv_two_r_prime, curvature_enabled, apply_curvature, any_hit):
    from forge3d_amd import _native

    n = rays.shape[0]
    hit, t, nrm = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    err = C.create_string_buffer(256)
    dem = np.ascontiguousarray(dem, np.float32)
    rc = _native.lib().f3d_terrain_trace_batch(dem.ctypes.data, dem.shape[1], dem.shape[0], origin[0], origin[1],
                                               spacing[0], spacing[1], 1.0, inv_two_r_prime, 1 if curvature_enabled else 0,
                                               rays.ctypes.data, n, int(any_hit), 1 if apply_curvature else 0,
                                               hit.ctypes.data, t.ctypes.data, nrm.ctypes.data, err, len(err))
    assert rc == 0, err.value
    return {"hit": hit, "t": t, "normal": nrm}


# descent (0 closest / 1 any), march (2 any / 3 closest, +4 = start in the origin's cell), and the any-hit march
# with the ray-sharing threshold at 64 / 2 lanes (bits 8..15)
MODES = (0, 1, 2, 6, 3, 7, 2 | (64 << 8), 6 | (64 << 8), 6 | (2 << 8))


@pytest.mark.parametrize("name", sorted(DEMS))
@pytest.mark.parametrize("spacing,curved", [(1.0, False), (0.5, True), (10.0, False), (2.0, False)])
def test_lattice_aligned_rays_match_the_oracle_on_the_device(f3d, oracle, name, spacing, curved):
    check_ray_set(device_trace, name, spacing, curved, MODES)


@pytest.mark.parametrize("share", [0, 64])
def test_ray_sharing_on_divergent_partial_waves(f3d, oracle, share):
    """The proof rays reordered so that every wave mixes short and very long marches, in a batch whose last
    wave is partial: any-hit verdicts with the sharing threshold at its default and at 64 lanes."""
    heights, rays = scenes.proof_rays(n_random=6000, mask=True)
    rng = np.random.default_rng(11)
    rays = rays[rng.permutation(rays.shape[0])][: 64 * 900 + 37].copy()
    base = dict(origin=(0.0, 0.0), spacing=(500.0, 500.0), inv_two_r_prime=0.0, curvature_enabled=False,
                apply_curvature=False)
    want = oracle.terrain_trace_batch(heights, rays, any_hit=True, **base)
    for mode in (2, 6):
        got = device_trace(heights, rays, any_hit=mode | (share << 8), **base)
        assert np.array_equal(got["hit"], want["hit"]), (mode, share)


@pytest.mark.parametrize("case", scenes.adversarial_scenes(), ids=lambda c: c[0])
def test_lattice_aligned_renders_match_the_oracle_on_the_device(f3d, oracle, case):
    from forge3d_amd.session import TerrainSession

    _, dem, size, cam, kw = case
    want = oracle.render(dem, size[0], size[1], cam, **kw)
    for variant in (0, 640000000 + 1000000, 40000000 + 8000000):  # default; share-at-64 with 1 lane; share-at-4 with 8
        with TerrainSession(dem, size[0], size[1], cam, kernel_variant=variant, **kw) as s:
            s.enqueue_frames(0, kw["max_frames"], True)
            m2, bad = s.window_stats()
            got = s.resolve(kw["max_frames"])
        assert not bad
        assert np.float32(max(0.0, m2) / np.float32(kw["max_frames"] - 1)) == np.float32(want["variance"])
        for key in ("rgba", "albedo", "normal", "depth"):
            assert np.array_equal(got[key], want[key], equal_nan=True), (variant, key)


@pytest.mark.parametrize("seed", [100, 104, 111, 117, 123, 129])
def test_random_scenes_with_every_ray_shared(f3d, oracle, seed):
    """The fuzz scenes of test_gpu_parity with the sharing threshold at 64 lanes: the dealing code runs for
    every IBL ray instead of only for the tails."""
    from forge3d_amd.session import TerrainSession

    dem, size, cam, kw = scenes.random_scene(seed)
    want = oracle.render(dem, size[0], size[1], cam, **kw)
    with TerrainSession(dem, size[0], size[1], cam, kernel_variant=640000000, **kw) as s:
        s.enqueue_frames(0, kw["max_frames"], True)
        s.window_stats()
        got = s.resolve(kw["max_frames"])
    for key in ("rgba", "albedo", "normal", "depth"):
        assert np.array_equal(got[key], want[key], equal_nan=True), key
