"""CPU-side parity of the product's KERNEL CODE (host-compiled by tests/emul, test
infrastructure) against the oracle: the restructured traversal (push-time culling, fat
leaves, path-coded pending siblings), the packed 16-byte reservoirs and the single fused
pass per frame must reproduce the literal three-pass restatement bit for bit.  The `-m gpu`
tests repeat every comparison on the device through libf3dhip.so.
"""
from __future__ import annotations

import os

import numpy as np
import pytest

import scenes
from emul import emul
from oracle import oracle

QUAD_V = np.array([[-18.0, 22.0, -6.0], [18.0, 22.0, -6.0], [18.0, 40.0, -6.0], [-18.0, 40.0, -6.0]], np.float32)
QUAD_I = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)


def _same(a, b):
    for key in ("rgba", "albedo", "normal"):
        assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(a["depth"], b["depth"], equal_nan=True)
    assert a["frames"] == b["frames"] and np.float32(a["variance"]) == np.float32(b["variance"])


@pytest.mark.parametrize("size,spp,frames,extra", [
    ((64, 64), 1, 3, {}),
    ((96, 64), 3, 5, {}),
    ((80, 80), 2, 4, {"mesh_vertices": QUAD_V, "mesh_indices": QUAD_I}),
    ((128, 96), 2, 34, {}),                       # crosses a Welford window + the M-clamp
    ((72, 56), 2, 4, {"earth_model": "flat", "refraction_model": "none", "sun_color": (0.2, 0.3, 1.5)}),
])
def test_kernel_code_matches_oracle(size, spp, frames, extra):
    dem = scenes.golden_dem()
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), frames, spp=spp, **extra)
    want = oracle.render(dem, size[0], size[1], scenes.CAM, dump_state=True, **kw)
    got = emul.render(dem, size[0], size[1], scenes.CAM, **kw)
    _same(got, want)
    assert np.array_equal(got["accum"][:, :3], want["accum"][:, :3])          # accumulated radiance
    assert np.array_equal(got["accum"][:, 3], want["welford"][:, 0])          # Welford mean
    assert np.array_equal(got["m2"], want["welford"][:, 1])                   # Welford m2


@pytest.mark.parametrize("lanes,size,spp,frames,extra", [
    (2, (96, 64), 3, 5, {}),                      # odd spp: the last round has an idle lane
    (4, (96, 64), 6, 4, {}),
    (8, (96, 80), 8, 3, {}),
    (8, (64, 48), 19, 2, {}),                     # three rounds, the last one ragged
    (4, (80, 80), 4, 4, {"mesh_vertices": QUAD_V, "mesh_indices": QUAD_I}),
    (8, (128, 96), 8, 34, {}),                    # Welford window + M-clamp
    (8, (64, 64), 1, 3, {}),                      # fewer samples than lanes
])
def test_sample_lane_frames_match_oracle(lanes, size, spp, frames, extra):
    """The frame kernel's sample-lane form (S samples of a pixel traced at once, hit flags
    predicted from the G-buffer, contributions replayed in order) reproduces the sequential
    sample loop bit for bit -- including the silhouette pixels where the prediction is wrong."""
    dem = scenes.golden_dem()
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), frames, spp=spp, **extra)
    want = oracle.render(dem, size[0], size[1], scenes.CAM, dump_state=True, **kw)
    got = emul.render(dem, size[0], size[1], scenes.CAM, sample_lanes=lanes, **kw)
    _same(got, want)
    assert np.array_equal(got["accum"][:, :3], want["accum"][:, :3])
    assert np.array_equal(got["accum"][:, 3], want["welford"][:, 0])
    assert np.array_equal(got["m2"], want["welford"][:, 1])
    if spp > 1:
        assert got["retraces"] > 0  # the scene has silhouettes: mispredictions were exercised


@pytest.mark.parametrize("seed,lanes", [(7, 1), (11, 4), (23, 8)])
def test_mesh_bvh_reproduces_the_reference_sweep(seed, lanes):
    """The threaded BVH (f3d_bvh.h) against the oracle's sweep over every triangle
    (hybrid_traversal.wgsl:137-172) on ~1 500 triangles: boxes whose quads are coplanar triangle
    pairs (equal-t ties on the diagonals -> lowest index wins), slivers, a degenerate triangle."""
    dem = scenes.golden_dem()
    v, i = scenes.box_city(seed=seed)
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), 3, spp=max(2, lanes), mesh_vertices=v, mesh_indices=i)
    want = oracle.render(dem, 144, 112, scenes.CAM, dump_state=True, **kw)
    got = emul.render(dem, 144, 112, scenes.CAM, sample_lanes=lanes, **kw)
    _same(got, want)
    assert np.array_equal(got["accum"][:, :3], want["accum"][:, :3])
    assert int((want["albedo"][..., 2] > 0.7).sum()) > 2000  # the mesh is really in view


def test_threaded_bvh_build_equals_the_single_thread_build():
    """Large meshes are built by worker threads (top levels split on the caller, subtrees in their own
    arenas, spliced into preorder): the arrays must be the single-thread build's, byte for byte."""
    from forge3d_amd import datasets

    dem = scenes.golden_dem()
    for boxes in (3, 40, 2500):
        v, i = datasets.proxy_buildings(dem, 1.0, n_boxes=boxes, seed=boxes)
        assert emul.bvh_fingerprint(v, i, True) == emul.bvh_fingerprint(v, i, False)


@pytest.mark.parametrize("seed", list(range(100, 112)) + [1179])
def test_random_scenes_match_the_oracle(seed):
    """Fuzz (the GPU suite runs more seeds): random ragged / terraced DEMs, cameras, suns, models, meshes."""
    dem, size, cam, kw = scenes.random_scene(seed)
    lanes = [1, 2, 4, 8][seed % 4]
    try:
        want = oracle.render(dem, size[0], size[1], cam, **kw)
    except RuntimeError as exc:  # seed 1179: the render itself is an error ("no valid reservoirs ...")
        with pytest.raises(RuntimeError, match="no valid reservoirs"):
            emul.render(dem, size[0], size[1], cam, sample_lanes=lanes, **kw)
        assert "no valid reservoirs" in str(exc)
        return
    _same(emul.render(dem, size[0], size[1], cam, sample_lanes=lanes, **kw), want)


def test_env_map_and_ragged_dem():
    dem = scenes.golden_dem(2)[:37, :100].copy()
    env = np.random.default_rng(5).uniform(0.1, 2.0, size=(16, 32, 3)).astype(np.float32)
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), 4, spp=2, env_map=env, seed=99)
    cam = {**scenes.CAM, "origin": (10.0, 30.0, 60.0), "fov_y": 60.0, "exposure": 1.7}
    _same(emul.render(dem, 80, 72, cam, **kw), oracle.render(dem, 80, 72, cam, **kw))


def test_two_by_two_dem_root_is_the_leaf():
    dem = np.array([[0.0, 1.0], [2.0, 0.5]], np.float32)
    kw = dict(spacing=(40.0, 40.0), exaggeration=10.0, max_frames=4, min_frames=4, variance_threshold=1e30, spp=2)
    _same(emul.render(dem, 48, 40, scenes.CAM, **kw), oracle.render(dem, 48, 40, scenes.CAM, **kw))


def test_converging_render_stops_at_the_same_frame():
    dem = scenes.golden_dem(4)
    kw = {**scenes.scene_kwargs(dem), "variance_threshold": 5e-3, "max_frames": 256}
    a, b = emul.render(dem, 64, 64, scenes.CAM, **kw), oracle.render(dem, 64, 64, scenes.CAM, **kw)
    _same(a, b)
    assert a["frames"] % 32 == 0 and a["frames"] < 256


def test_traversal_matches_oracle_on_the_reference_proof_rays():
    heights, rays = scenes.proof_rays(n_random=4000, mask=True)
    inv2r = float(np.float32(1.0 / 14_650_000.0))
    for any_hit, curv in ((True, True), (False, False), (True, False)):
        kw = dict(spacing=(500.0, 500.0), inv_two_r_prime=inv2r, curvature_enabled=True, any_hit=any_hit,
                  apply_curvature=curv)
        want, got = oracle.terrain_trace_batch(heights, rays, **kw), emul.terrain_trace_batch(heights, rays, **kw)
        assert np.array_equal(got["hit"], want["hit"])
        assert np.array_equal(got["t"], want["t"])
        assert np.array_equal(got["normal"], want["normal"])


@pytest.mark.parametrize("curved", [True, False])
def test_stackless_march_matches_oracle_on_the_proof_rays(curved):
    """The frame kernel's traversal (f3d_march.h) in its four start/answer modes: any hit (2) and
    closest hit (3), from the root or from the origin's cell (+4)."""
    heights, rays = scenes.proof_rays(n_random=6000, mask=True)
    inv2r = float(np.float32(1.0 / 14_650_000.0))
    base = dict(spacing=(500.0, 500.0), inv_two_r_prime=inv2r, curvature_enabled=True, apply_curvature=curved)
    want_any = oracle.terrain_trace_batch(heights, rays, any_hit=True, **base)
    want_closest = oracle.terrain_trace_batch(heights, rays, any_hit=False, **base)
    for mode in (2, 6):
        got = emul.terrain_trace_batch(heights, rays, any_hit=mode, **base)
        assert np.array_equal(got["hit"], want_any["hit"]), mode
    for mode in (3, 7):
        got = emul.terrain_trace_batch(heights, rays, any_hit=mode, **base)
        assert np.array_equal(got["hit"], want_closest["hit"]), mode
        assert np.array_equal(got["t"], want_closest["t"]) and np.array_equal(got["normal"], want_closest["normal"])


@pytest.mark.parametrize("slices,level", [(2, 0), (5, 3), (8, 6), (15, 8), (3, 15)])
def test_any_hit_ray_cut_into_slices_gives_the_same_answer(slices, level):
    """Ray sharing (f3d_march.h march_shared): an any-hit ray walked as geometric parameter slices, each
    started at a node of `level` located from the position, ORs to the oracle's answer on all proof rays
    (the device deals such slices over the lanes of a wave; here they run one after the other)."""
    heights, rays = scenes.proof_rays(n_random=6000, mask=True)
    inv2r = float(np.float32(1.0 / 14_650_000.0))
    for curved in (True, False):
        base = dict(spacing=(500.0, 500.0), inv_two_r_prime=inv2r, curvature_enabled=True, apply_curvature=curved)
        want = oracle.terrain_trace_batch(heights, rays, any_hit=True, **base)
        for start in (0, 4):
            got = emul.terrain_trace_batch(heights, rays, any_hit=2 | start | (slices << 4) | (level << 8), **base)
            assert np.array_equal(got["hit"], want["hit"]), (curved, start)


@pytest.mark.parametrize("shape", [(256, 256), (37, 100), (2, 2), (3, 9), (130, 65), (9, 3)])
def test_table_builder_matches_build_minmax_mips(shape):
    dem = np.random.default_rng(shape[0] * 1000 + shape[1]).normal(1000.0, 300.0, size=shape).astype(np.float32)
    want, _ = oracle.build_minmax_mips(dem)
    got = emul.build_minmax_mips(dem)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.array_equal(a, b)


def test_600k_triangle_bvh_reproduces_the_sweep():
    """BASELINE.json configs[3] stand-in: 600 000 triangles, built by the multi-arena worker-thread path
    (>= 64 subtrees spliced), walked by the kernel code against the oracle's sweep over every triangle; a
    24x24 close-up keeps the sweep at a few seconds (the GPU suite runs 64x64 through the device)."""
    from forge3d_amd import datasets

    dem = datasets.rainier_proxy(512)
    spacing = 40.0
    v, i = datasets.proxy_buildings(dem, spacing)
    centres = v.reshape(-1, 8, 3).mean(1)
    cell = np.floor(centres[:, [0, 2]] / 250.0).astype(np.int64)
    uniq, counts = np.unique(cell, axis=0, return_counts=True)
    spot = (uniq[counts.argmax()] + 0.5) * 250.0
    near = centres[np.hypot(centres[:, 0] - spot[0], centres[:, 2] - spot[1]) < 200.0]
    target = (float(spot[0]), float(near[:, 1].mean()), float(spot[1]))
    cam = {"origin": (target[0] + 190.0, target[1] + 130.0, target[2] + 150.0), "look_at": target,
           "up": (0.0, 1.0, 0.0), "fov_y": 55.0, "exposure": 1.0}
    kw = dict(spacing=(spacing, spacing), exaggeration=1.0, albedo=(0.6, 0.6, 0.6), sun_azimuth_deg=302.0,
              sun_elevation_deg=24.0, spp=1, max_frames=2, min_frames=2, variance_threshold=1e30,
              mesh_vertices=v, mesh_indices=i)
    want = oracle.render(dem, 24, 24, cam, **kw)
    assert float((want["albedo"][..., 2] > 0.75).mean()) > 0.2  # buildings fill a good part of the view
    _same(emul.render(dem, 24, 24, cam, **kw), want)


@pytest.mark.parametrize("size,spp,frames,in_flight,sun,force", [
    ((96, 64), 4, 11, 4, (315.0, 45.0), False),     # batches 2, 2, 4, 3
    ((64, 48), 2, 34, 16, (302.0, 24.0), False),    # crosses a Welford window: batches never do; the headline sun angles
    ((80, 60), 3, 9, 8, (135.0, 12.0), True),       # predictions switched on whatever the bits, plus re-traces for no reason
    ((72, 40), 1, 6, 2, (17.0, 61.0), True),
])
def test_frames_in_flight_structure_matches_the_oracle(monkeypatch, size, spp, frames, in_flight, sun, force):
    """DESIGN.md 4.7 on the host: the per-pixel code of k_trace (records traced for a whole batch of frames first),
    k_merge (head, records through the ordered sums with the real reuse weight, tail) and k_fix (mispredicted pixel-frames
    traced again) -- same batches, same prediction flags as the device path -- reproduces the fused per-frame loop and
    the oracle bit for bit, state buffers included."""
    if force:
        monkeypatch.setenv("F3D_EMUL_FORCE_PREDICTION", "1")
    dem = scenes.golden_dem()
    kw = scenes.fixed_frames(dict(scenes.scene_kwargs(dem), sun_azimuth_deg=sun[0], sun_elevation_deg=sun[1]), frames, spp=spp)
    want = oracle.render(dem, size[0], size[1], scenes.CAM, dump_state=True, **kw)
    got = emul.render(dem, size[0], size[1], scenes.CAM, frames_in_flight=in_flight, **kw)
    _same(got, want)
    assert np.array_equal(got["accum"][:, :3], want["accum"][:, :3])
    assert np.array_equal(got["accum"][:, 3], want["welford"][:, 0])
    assert np.array_equal(got["m2"], want["welford"][:, 1])
    if force:
        assert got["retraced_pixels"] > frames  # the re-trace pass ran in every frame


def _mesh_scene_seeds(first, count):
    seeds, seed = [], first
    while len(seeds) < count:
        if scenes.random_scene(seed)[3].get("mesh_vertices") is not None:
            seeds.append(seed)
        seed += 1
    return seeds


@pytest.mark.parametrize("seed", _mesh_scene_seeds(40000, 16))
@pytest.mark.parametrize("lanes", [1, 4])
def test_mesh_walk_on_random_scenes_matches_the_sweep(seed, lanes):
    """The threaded-BVH walk of csrc/f3d_shade.h (terrain asked first by the occlusion rays, successor as a select, slab test
    as one fma per plane) against the oracle's sweep over all triangles, on random terrain + mesh scenes of the fuzz
    generator (tools/fuzz_emul_mesh.py runs thousands of them): every output the same bits."""
    dem, size, cam, kw = scenes.random_scene(seed)
    kw = dict(kw, max_frames=3, min_frames=3, variance_threshold=1e30)
    try:
        want = oracle.render(dem, size[0], size[1], cam, **kw)
    except RuntimeError as exc:  # (a scene both sides reject: the generator makes a few)
        with pytest.raises(RuntimeError):
            emul.render(dem, size[0], size[1], cam, sample_lanes=lanes, **kw)
        pytest.skip(str(exc)[:60])
    _same(emul.render(dem, size[0], size[1], cam, sample_lanes=lanes, **kw), want)


def test_four_wide_mesh_walk_is_what_runs_and_equals_the_other_forms():
    """Round 4: the mesh BVH is walked four children wide (f3d_shade.h mesh_bvh4 over f3d_bvh.h collapse_bvh4).  The mesh
    tests above run that form (the emulator's default, like the product's); here it is checked that the collapse really
    yields a tree (a too-deep one would silently fall back to the binary walk) and that sweep, binary walk and 4-wide walk
    give the same bits on a scene with coplanar pairs and slivers."""
    from forge3d_amd import datasets

    dem = scenes.golden_dem()
    v, i = datasets.proxy_buildings(dem, 100.0 / (dem.shape[1] - 1), n_boxes=400, seed=3)
    v = v * np.float32([1.0, 0.02, 1.0])  # the proxy's metre-sized boxes on the unit-height golden DEM
    wide, binary = emul.bvh4_nodes(v, i)
    assert 0 < wide < binary // 3  # four children a record: far fewer records than binary nodes
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), 2, spp=2, mesh_vertices=v, mesh_indices=i)
    images = []
    try:
        for form in (0, 1, 2):
            emul.set_mesh_walk(form)
            images.append(emul.render(dem, 96, 72, scenes.CAM, **kw))
    finally:
        emul.set_mesh_walk(2)
    assert int((images[0]["albedo"][..., 2] > 0.7).sum()) > 100  # the mesh is in view
    for other in images[1:]:
        for key in ("rgba", "albedo", "normal", "depth"):
            assert np.array_equal(images[0][key], other[key], equal_nan=True), key


@pytest.mark.parametrize("flags", ["-DF3D_MESH_FUSED", "-DF3D_CLOSEST_TERRAIN_FIRST", "-DF3D_BVH4_ANY_SLOT_ORDER"])
def test_build_switches_of_the_mesh_path_equal_the_oracle(flags):
    """The mesh path's A/B builds (round 6: the mesh as a second band of the terrain's pyramid fused into the occlusion rays'
    march, terrain before mesh for camera rays, occlusion rays walking their children in slot order -- all measured slower and
    not in the shipped library) stay bit-identical to the oracle's sweep: the emulator compiled with the switch (a library of its
    own per set of switches, tests/emul/emul.py) on meshes inside the DEM's footprint, in a process of its own."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, F3D_EMUL_CXXFLAGS=flags)
    out = subprocess.run([sys.executable, str(root / "tools" / "fuzz_emul_mesh.py"), "7000", "24", "inside"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-400:]
    assert "24 mesh scenes from seed 7000: mismatches []" in out.stdout, out.stdout[-300:]
