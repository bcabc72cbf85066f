"""Adversarial, lattice-aligned inputs for the stackless march ("measure-zero" deviations).

Random scenes never produce exact f32 ties between slab parameters; structured ones do all the time: a
camera on the footprint's diagonal looking at its centre, a sun at azimuth 45, rays that start on lattice
points and run along lattice directions.  A ray through a lattice corner touches the two cells beside it in
one point, the reference judges them over a zero-length interval, and an any-hit ray below the surface is
answered there (hybrid_terrain_traversal.wgsl:197-201, :288-297).  The march must reproduce that, from the
root and from the origin's cell, whole and cut into slices.  CPU: the kernel headers host-compiled by
tests/emul; `-m gpu`: the same sets through libf3dhip.so (tests/test_gpu_adversarial.py).
"""
from __future__ import annotations

import numpy as np
import pytest

import scenes
from emul import emul
from oracle import oracle

DEMS = scenes.adversarial_dems(33)


def check_ray_set(trace, name, spacing, curved, modes):
    dem = DEMS[name]
    rays = scenes.adversarial_rays(dem, spacing)
    h, w = dem.shape
    base = dict(origin=(-0.5 * (w - 1) * spacing, -0.5 * (h - 1) * spacing), spacing=(spacing, spacing),
                inv_two_r_prime=float(np.float32(1e-4)) if curved else 0.0, curvature_enabled=curved,
                apply_curvature=curved)
    want_any = oracle.terrain_trace_batch(dem, rays, any_hit=True, **base)
    want_closest = oracle.terrain_trace_batch(dem, rays, any_hit=False, **base)
    assert 0.05 < want_any["hit"].mean() < 0.95  # the set is not trivially all-hit / all-miss
    for mode in modes:
        got = trace(dem, rays, any_hit=mode, **base)
        if mode == 0 or (mode & 3) == 3:
            assert np.array_equal(got["hit"], want_closest["hit"]), (name, spacing, curved, mode)
            assert np.array_equal(got["t"], want_closest["t"]), (name, spacing, curved, mode)
            assert np.array_equal(got["normal"], want_closest["normal"]), (name, spacing, curved, mode)
        else:
            bad = np.nonzero(got["hit"] != want_any["hit"])[0]
            assert bad.size == 0, (name, spacing, curved, mode, bad.size, rays[bad[:2]].tolist())
            if mode == 1:
                assert np.array_equal(got["t"], want_any["t"])


# any-hit / closest through the sorted descent (1 / 0) and the march (2 / 3; +4 = start in the origin's cell);
# the two big numbers are any-hit rays cut into 4 slices started at level 2 from the root, and into 7 slices
# started at level 0 from the origin's cell (ray sharing, f3d_march.h march_shared)
MODES = (0, 1, 2, 6, 3, 7, 2 | (4 << 4) | (2 << 8), 6 | (7 << 4) | (0 << 8))


@pytest.mark.parametrize("name", sorted(DEMS))
@pytest.mark.parametrize("spacing,curved", [(1.0, False), (0.5, True), (10.0, False)])
def test_lattice_aligned_rays_match_the_oracle(name, spacing, curved):
    check_ray_set(emul.terrain_trace_batch, name, spacing, curved, MODES)


@pytest.mark.parametrize("case", scenes.adversarial_scenes()[::5], ids=lambda c: c[0])
def test_lattice_aligned_renders_match_the_oracle(case):
    _, dem, size, cam, kw = case
    want = oracle.render(dem, size[0], size[1], cam, **kw)
    for lanes in (1, 4):
        got = emul.render(dem, size[0], size[1], cam, sample_lanes=lanes, **kw)
        for key in ("rgba", "albedo", "normal"):
            assert np.array_equal(got[key], want[key]), (key, lanes)
        assert np.array_equal(got["depth"], want["depth"], equal_nan=True)
        assert np.float32(got["variance"]) == np.float32(want["variance"])


# ---- the ray sharing on a 64-lane wave (tests/emul Wave: lanes are fibers, ballots / shuffles in lockstep) --------
@pytest.mark.parametrize("share", [0, 64, 3])
def test_ray_sharing_on_the_host_wave_matches_the_oracle(share):
    """march_shared / march_deal / the verdict board executed by 64 cooperating lanes on the CPU: the proof rays
    shuffled so that every wave mixes short and very long marches, the last wave partial; sharing threshold at
    its default, at 64 lanes (every ray dealt from its first step) and at 3."""
    heights, rays = scenes.proof_rays(n_random=5000, mask=True)
    rays = rays[np.random.default_rng(11).permutation(rays.shape[0])][: 64 * 300 + 37].copy()
    base = dict(spacing=(500.0, 500.0), inv_two_r_prime=0.0, curvature_enabled=False, apply_curvature=False)
    want_any = oracle.terrain_trace_batch(heights, rays, any_hit=True, **base)
    want_closest = oracle.terrain_trace_batch(heights, rays, any_hit=False, **base)
    for mode in (2, 6):
        got = emul.terrain_trace_batch_wave(heights, rays, any_hit=mode, share_below=share, **base)
        assert np.array_equal(got["hit"], want_any["hit"]), (mode, share)
        assert got["deals"] > 0  # the dealing code really ran
    for mode in (3, 7):  # closest-hit rays are dealt too (march_shared_closest; the host wave's context asks for it)
        got = emul.terrain_trace_batch_wave(heights, rays, any_hit=mode, share_below=share, **base)
        assert np.array_equal(got["hit"], want_closest["hit"]) and np.array_equal(got["t"], want_closest["t"]), (mode, share)
        assert np.array_equal(got["normal"], want_closest["normal"]) and got["deals"] > 0, (mode, share)


@pytest.mark.parametrize("name", ["diagplane", "ragged", "terrace", "rand_sym_int"])
def test_lattice_aligned_rays_on_the_host_wave(name):
    """The adversarial sets through the wave: corner ties inside dealt slices, slices that begin on lattice planes."""
    dem = DEMS[name]
    for spacing, share in ((1.0, 64), (10.0, 0), (0.5, 5)):
        rays = scenes.adversarial_rays(dem, spacing)
        h, w = dem.shape
        base = dict(origin=(-0.5 * (w - 1) * spacing, -0.5 * (h - 1) * spacing), spacing=(spacing, spacing),
                    inv_two_r_prime=0.0, curvature_enabled=False, apply_curvature=False)
        want = oracle.terrain_trace_batch(dem, rays, any_hit=True, **base)
        for mode in (2, 6):
            got = emul.terrain_trace_batch_wave(dem, rays, any_hit=mode, share_below=share, **base)
            bad = np.nonzero(got["hit"] != want["hit"])[0]
            assert bad.size == 0, (name, spacing, share, mode, bad.size, rays[bad[:2]].tolist())
        want = oracle.terrain_trace_batch(dem, rays, any_hit=False, **base)
        for mode in (3, 7):  # closest-hit rays cut into slices: the first hit in ray order, its parameter and normal
            got = emul.terrain_trace_batch_wave(dem, rays, any_hit=mode, share_below=share, **base)
            bad = np.nonzero((got["hit"] != want["hit"]) | (got["t"] != want["t"]) | (got["normal"] != want["normal"]).any(axis=1))[0]
            assert bad.size == 0, (name, spacing, share, mode, bad.size, rays[bad[:2]].tolist())
