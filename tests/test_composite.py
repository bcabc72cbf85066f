"""Smoke over terrain (BASELINE.json configs[4]): the per-pixel composites the reference's smoke sequence example does
with numpy and Pillow (examples/california_cigar_smoke_demo.py:8527-8544, :3367-3380, Image.alpha_composite), as one
device pass (f3d_smoke_composite).

Pins, in order: the oracle (oracle/composite_oracle.c) against outputs of the reference's own functions
(tests/golden/smoke/composite_vectors.npz, made by tests/golden/make_composite_vectors.py); the reference's three
composite tests (tests/test_california_cigar_smoke_hybrid.py:235-283) restated on the oracle; the device code compiled
for the host against the oracle; and, on the GPU, the HIP pass against the oracle at fixture size and at 1080p, and
the whole config end to end: emitters -> solver -> ray-marcher -> composite over a path-traced terrain frame.
"""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from oracle import smoke_oracle as so

VEC = Path(__file__).resolve().parent / "golden" / "smoke" / "composite_vectors.npz"
# the oracle takes x^0.9 and e^x from fixed polynomials, numpy from its own loops: a value may land on the other side
# of a truncation.  Bound: one code value, on at most 1 channel value in 10 000 (measured on these vectors: none).
ATM_MAX_DIFF, ATM_MAX_RATE = 1, 1.0e-4


@pytest.fixture(scope="module")
def vec():
    return np.load(VEC)


def random_rgba(rng, h, w):
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    band = rng.integers(0, 5, (h, w))
    img[..., 3][band == 0] = 0
    img[..., 3][band == 2] = 255
    return img


@pytest.mark.parametrize("name", ["atm", "atm_sweep", "atm_test"])
def test_oracle_atmospheric_matches_the_reference_function(vec, name):
    got = so.composite_atmospheric(vec[name + "_base"], vec[name + "_smoke"])
    want = vec[name + "_out"]
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert int(diff.max()) <= ATM_MAX_DIFF
    assert np.count_nonzero(diff) <= ATM_MAX_RATE * diff.size
    assert np.all(got[..., 3] == 255)


def test_oracle_smoke_maps_match_the_reference_function_bit_for_bit(vec):
    a, p = vec["maps_atmospheric"], vec["maps_physical"]
    assert np.array_equal(so.composite_smoke_maps(a, p), vec["maps_out"])
    assert np.array_equal(so.composite_smoke_maps(a, None), vec["maps_out_none"])
    assert np.array_equal(so.composite_smoke_maps(a, p, 0.68, 0.58), vec["maps_out_scaled"])
    assert np.array_equal(so.composite_smoke_maps(vec["maps_test_atmospheric"], vec["maps_test_physical"]), vec["maps_test_out"])


def test_oracle_over_matches_pillow_bit_for_bit(vec):
    assert np.array_equal(so.composite_over(vec["over_base"], vec["over_layer"]), vec["over_out"])
    for n in "abc":
        off = tuple(int(c) for c in vec["over_small_offset_" + n])
        assert np.array_equal(so.composite_over(vec["over_base"], vec["over_small"], off), vec["over_small_out_" + n])


def test_reference_composite_tests_hold_on_the_oracle(vec):
    # test_atmospheric_composite_has_visible_smoke_lift, :235-243
    base = np.empty((24, 24, 4), np.uint8)
    base[:] = (34, 37, 38, 255)
    smoke = np.empty((24, 24, 4), np.uint8)
    smoke[:] = (214, 218, 214, 104)
    out = so.composite_atmospheric(base, smoke)[..., :3].astype(np.float32)
    assert float(np.mean(out - base[..., :3].astype(np.float32))) > 40.0
    # test_main_smoke_compositor_keeps_atmospheric_blanket_with_physical_detail, :267-283
    atmospheric, physical = vec["maps_test_atmospheric"], vec["maps_test_physical"]
    combined = so.composite_smoke_maps(atmospheric, physical)
    alpha = combined[..., 3]
    assert combined.shape == atmospheric.shape
    assert int(alpha.max()) <= int(vec["max_alpha"])
    assert np.count_nonzero(alpha > 0) > np.count_nonzero(physical[..., 3] > 0) * 3
    assert int(alpha[12, 10]) > 20
    assert int(alpha[22, 28]) > int(physical[22, 28, 3] * 0.80)


def test_a_clear_layer_leaves_the_terrain_and_an_outside_layer_the_base():
    rng = np.random.default_rng(5)
    base = random_rgba(rng, 20, 30)
    clear = np.zeros_like(base)
    out = so.composite_atmospheric(base, clear)
    assert np.array_equal(out[..., :3], base[..., :3])  # optical depth 0: transmittance 1, nothing added
    layer = random_rgba(rng, 8, 8)
    for off in ((30, 0), (0, 20), (-8, 0), (0, -8), (1000, 1000)):
        assert np.array_equal(so.composite_over(base, layer, off), base)
    partly = so.composite_over(base, layer, (-3, 15))  # clipped on two sides
    assert np.array_equal(partly[:15], base[:15]) and np.array_equal(partly[:, 5:], base[:, 5:])
    assert not np.array_equal(partly, base)


CASES = [(96, 128), (1, 1), (7, 13), (33, 250), (64, 1027)]


@pytest.mark.parametrize("h,w", CASES)
def test_device_code_on_the_host_equals_the_oracle(h, w):
    from tests.emul import emul

    rng = np.random.default_rng(100 + h * w)
    base, layer = random_rgba(rng, h, w), random_rgba(rng, h, w)
    opaque = base.copy()
    opaque[..., 3] = 255
    assert np.array_equal(emul.composite(0, opaque, layer), so.composite_atmospheric(opaque, layer))
    assert np.array_equal(emul.composite(1, base, layer, base_alpha=0.42, layer_alpha=0.92, max_alpha=168), so.composite_smoke_maps(base, layer))
    assert np.array_equal(emul.composite(1, base, None, base_alpha=0.68, layer_alpha=0.0, max_alpha=168), so.composite_smoke_maps(base, None, 0.68))
    assert np.array_equal(emul.composite(2, base, layer), so.composite_over(base, layer))
    small = random_rgba(rng, max(1, h // 2), max(1, w // 3))
    for off in ((0, 0), (w // 4, h // 3), (-2, -1), (w - 1, h - 1)):
        assert np.array_equal(emul.composite(2, base, small, offset=off), so.composite_over(base, small, off))


def test_device_code_on_the_host_equals_the_reference_vectors(vec):
    from tests.emul import emul

    assert np.array_equal(emul.composite(0, vec["atm_sweep_base"], vec["atm_sweep_smoke"]), so.composite_atmospheric(vec["atm_sweep_base"], vec["atm_sweep_smoke"]))
    assert np.array_equal(emul.composite(1, vec["maps_atmospheric"], vec["maps_physical"], base_alpha=0.42, layer_alpha=0.92, max_alpha=168), vec["maps_out"])
    assert np.array_equal(emul.composite(2, vec["over_base"], vec["over_layer"]), vec["over_out"])


def test_python_surface_without_a_gpu():
    import torch

    from forge3d_amd import smoke

    base = np.zeros((4, 4, 4), np.uint8)
    with pytest.raises(ValueError):
        smoke.composite_atmospheric_smoke(base[..., :3], base)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            smoke.composite_atmospheric_smoke(base, base)


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("h,w", CASES + [(1080, 1920)])
def test_hip_equals_the_oracle(h, w):
    from forge3d_amd import smoke

    rng = np.random.default_rng(200 + h * w)
    base, layer = random_rgba(rng, h, w), random_rgba(rng, h, w)
    opaque = base.copy()
    opaque[..., 3] = 255
    assert np.array_equal(smoke.composite_atmospheric_smoke(opaque, layer), so.composite_atmospheric(opaque, layer))
    assert np.array_equal(smoke.composite_main_smoke_maps(base, layer), so.composite_smoke_maps(base, layer))
    assert np.array_equal(smoke.composite_main_smoke_maps(base, None, atmospheric_alpha=0.68), so.composite_smoke_maps(base, None, 0.68))
    assert np.array_equal(smoke.alpha_composite(base, layer), so.composite_over(base, layer))
    small = random_rgba(rng, max(1, h // 2), max(1, w // 3))
    for off in ((0, 0), (w // 4, h // 3), (-2, -1), (w - 1, h - 1), (w, h)):
        assert np.array_equal(smoke.alpha_composite(base, small, off), so.composite_over(base, small, off))


@pytest.mark.gpu
def test_hip_equals_the_reference_vectors(vec):
    from forge3d_amd import smoke

    for name in ("atm", "atm_sweep", "atm_test"):
        got = smoke.composite_atmospheric_smoke(vec[name + "_base"], vec[name + "_smoke"])
        diff = np.abs(got.astype(np.int16) - vec[name + "_out"].astype(np.int16))
        assert int(diff.max()) <= ATM_MAX_DIFF and np.count_nonzero(diff) <= ATM_MAX_RATE * diff.size
    assert np.array_equal(smoke.composite_main_smoke_maps(vec["maps_atmospheric"], vec["maps_physical"]), vec["maps_out"])
    assert np.array_equal(smoke.alpha_composite(vec["over_base"], vec["over_layer"]), vec["over_out"])


@pytest.mark.gpu
def test_device_pointers_and_errors():
    import torch

    from forge3d_amd import _native, smoke

    rng = np.random.default_rng(9)
    base, layer = random_rgba(rng, 37, 101), random_rgba(rng, 37, 101)
    tb, tl = torch.from_numpy(base).cuda(), torch.from_numpy(layer).cuda()
    out = torch.empty_like(tb)
    hb, hl = np.ascontiguousarray(base), np.ascontiguousarray(layer)
    desc = smoke.composite_desc(smoke.COMPOSITE_OVER, hb, hl)
    desc.base, desc.layer = tb.data_ptr(), tl.data_ptr()
    err = C.create_string_buffer(256)
    torch.cuda.synchronize()
    assert _native.lib().f3d_smoke_composite(C.byref(desc), C.c_void_p(out.data_ptr()), None, err, len(err)) == 0, err.value
    assert np.array_equal(out.cpu().numpy(), so.composite_over(base, layer))
    desc.struct_size = 8
    assert _native.lib().f3d_smoke_composite(C.byref(desc), C.c_void_p(out.data_ptr()), None, err, len(err)) == _native.STATUS_VALUE
    assert b"struct_size" in err.value
    with pytest.raises(ValueError, match="must match"):
        smoke.composite_atmospheric_smoke(base, layer[:10])


@pytest.mark.gpu
def test_config5_end_to_end_over_a_terrain_frame():
    """emitters -> f3d_smoke_step -> f3d_smoke_render -> f3d_smoke_composite over hybrid_render_terrain_reference, and
    the same chain on the four oracles: identical frames."""
    import forge3d_amd as f3d
    from forge3d_amd import smoke
    from oracle import oracle as terrain_oracle
    from tests import scenes

    w, h = 160, 96
    dem = scenes.golden_dem(2)
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), 3, spp=2)
    terrain = f3d.hybrid_render_terrain_reference(dem, w, h, scenes.CAM, **kw)["rgba"]
    want_terrain = terrain_oracle.render(dem, w, h, scenes.CAM, **kw)["rgba"]
    assert terrain.shape == (h, w, 4) and terrain.dtype == np.uint8 and np.array_equal(terrain, want_terrain)

    dims = (24, 16, 20)
    dom, st = smoke.SmokeDomain(dims), so.new_state(dims)
    emitters = [dict(center=(6.0, 3.0, 10.0), radius=2.5, density_rate=6.0, temperature_rate=3.0, soot_rate=0.3, emission_rate=2.0, velocity=(3.0, 0.4, 0.0))]
    settings = dict(dt=0.1, turbulence_strength=0.5, turbulence_seed=7, wind=(1.5, 0.0, -0.2), pressure_iterations=8)
    cam = dict(camera_pos=(12.0, 10.0, 46.0), target=(12.0, 6.0, 10.0))
    frames = smoke.simulate_over_terrain(terrain, dom, smoke.SmokeStepSettings(**settings), [smoke.SmokeEmitter(**e) for e in emitters], 3,
                                         steps_per_frame=2, **cam)
    assert sorted(frames) == [0, 1, 2]
    for f in range(3):
        so.step(st, emitters, steps=2, **settings)
        fields = {k: st[k] for k in ("density", "temperature", "soot", "humidity", "emission_rate", "particle_age")}
        layer = so.render_rgba(fields, w, h, frame_index=st["frame_index"], **cam)
        assert np.array_equal(frames[f], so.composite_atmospheric(want_terrain, layer)), f
    assert np.all(frames[2][..., 3] == 255)
    assert np.count_nonzero(np.any(frames[2][..., :3] != terrain[..., :3], axis=-1)) > 50  # the plume is visible
    # ranks of a sequence render disjoint frames of the same state
    dom2 = smoke.SmokeDomain(dims)
    mine = smoke.simulate_over_terrain(terrain, dom2, smoke.SmokeStepSettings(**settings), [smoke.SmokeEmitter(**e) for e in emitters], 3,
                                       steps_per_frame=2, rank=1, world=2, **cam)
    assert sorted(mine) == [1] and np.array_equal(mine[1], frames[1])


@pytest.mark.gpu
def test_two_resident_sequences_driven_in_turns_stay_themselves():
    """Round 5 put the solver and the marcher of a sequence on a stream each, with the library remembering the calling THREAD's
    stream and last march: two sequences advanced alternately by one thread (zip over their frames) must each give the frames
    they give alone -- with and without the overlap."""
    from forge3d_amd import smoke

    w, h = 160, 96
    rng = np.random.default_rng(11)
    terrain = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    terrain[..., 3] = 255
    cam = dict(camera_pos=(12.0, 10.0, 46.0), target=(12.0, 6.0, 10.0))

    def sequence(kind):
        dom = smoke.SmokeDomain((24, 16, 20))
        emitters = [smoke.SmokeEmitter(center=(6.0 + 8.0 * kind, 3.0, 10.0), radius=2.5, density_rate=6.0 - 2.0 * kind, temperature_rate=3.0, soot_rate=0.3,
                                       emission_rate=2.0, velocity=(3.0 - 5.0 * kind, 0.4, 0.0))]
        settings = smoke.SmokeStepSettings(dt=0.1, turbulence_strength=0.5, turbulence_seed=7 + kind, wind=(1.5, 0.0, -0.2), pressure_iterations=8)
        return smoke.SmokeSequence(dom, terrain, **cam), settings, emitters

    alone = []
    for kind in (0, 1):
        seq, settings, emitters = sequence(kind)
        alone.append([f.copy() for f in seq.frames(7, settings, emitters, steps_per_frame=2, overlap=False)])
    assert not np.array_equal(alone[0][-1], alone[1][-1])
    for overlap in (True, False):
        (a, sa, ea), (b, sb, eb) = sequence(0), sequence(1)
        got = [(fa.copy(), fb.copy()) for fa, fb in zip(a.frames(7, sa, ea, steps_per_frame=2, overlap=overlap),
                                                        b.frames(7, sb, eb, steps_per_frame=2, overlap=overlap))]
        for f, (fa, fb) in enumerate(got):
            assert np.array_equal(fa, alone[0][f]) and np.array_equal(fb, alone[1][f]), (overlap, f)


@pytest.mark.gpu
def test_sequence_handle_owns_its_scratch_and_reports_the_shadow_list():
    """ABI 6: the sequence's scratch belongs to its handle (f3d_smoke_seq_*): stats() sees it, close() gives it back, and a closed
    sequence renders the same frames again through a fresh handle.  The marcher's deferred self-shadow list reports its fill."""
    from forge3d_amd import smoke

    w, h = 160, 96
    rng = np.random.default_rng(3)
    terrain = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    terrain[..., 3] = 255
    cam = dict(camera_pos=(12.0, 10.0, 46.0), target=(12.0, 6.0, 10.0))
    emitters = [smoke.SmokeEmitter(center=(6.0, 3.0, 10.0), radius=2.5, density_rate=6.0, temperature_rate=3.0, soot_rate=0.3, emission_rate=2.0, velocity=(3.0, 0.4, 0.0))]
    settings = smoke.SmokeStepSettings(dt=0.1, turbulence_strength=0.5, turbulence_seed=7, wind=(1.5, 0.0, -0.2), pressure_iterations=8)

    def run(seq, overlap):
        return [f.copy() for f in seq.frames(4, settings, emitters, steps_per_frame=2, overlap=overlap)]

    seq = smoke.SmokeSequence(smoke.SmokeDomain((24, 16, 20)), terrain, **cam)
    first = run(seq, True)
    st = seq.stats()  # (of the handle used last: the two-stream schedule)
    assert st["scratch_bytes"] > 0 and st["shadow_list_slots_per_chunk"] == 1024
    assert 0 < st["shadow_list_chunks_used"] <= st["shadow_list_chunks"]  # smoke was marched and the list held it
    assert set(seq._handles) == {"overlap"}
    seq.close()
    assert seq._handles == {}
    other = smoke.SmokeSequence(smoke.SmokeDomain((24, 16, 20)), terrain, **cam)
    again = run(other, False)  # the serial schedule (null stream) through its own handle
    for a, b in zip(first, again):
        assert np.array_equal(a, b)
    other.close()


@pytest.mark.gpu
@pytest.mark.parametrize("knob", ["F3D_SMOKE_SHADOW_OOM", "F3D_SMOKE_SHADOW_MB"])
def test_a_shadow_list_that_cannot_be_had_changes_nothing(knob, monkeypatch):
    """Round-5 advice: the deferred self-shadow list is bounded (F3D_SMOKE_SHADOW_MB) and a list that cannot be allocated sends the
    frame through the one-kernel form instead of failing the call -- same frames either way."""
    from forge3d_amd import smoke

    w, h = 160, 96
    terrain = np.full((h, w, 4), 255, np.uint8)
    cam = dict(camera_pos=(12.0, 10.0, 46.0), target=(12.0, 6.0, 10.0))
    emitters = [smoke.SmokeEmitter(center=(6.0, 3.0, 10.0), radius=2.5, density_rate=6.0, temperature_rate=3.0, soot_rate=0.3, emission_rate=2.0, velocity=(3.0, 0.4, 0.0))]
    settings = smoke.SmokeStepSettings(dt=0.1, turbulence_strength=0.5, turbulence_seed=7, wind=(1.5, 0.0, -0.2), pressure_iterations=8)
    want = [f.copy() for f in smoke.SmokeSequence(smoke.SmokeDomain((24, 16, 20)), terrain, **cam).frames(4, settings, emitters, steps_per_frame=2)]
    monkeypatch.setenv(knob, "1" if knob.endswith("OOM") else "0")  # 0 MB: one chunk, nearly every tile spills
    seq = smoke.SmokeSequence(smoke.SmokeDomain((24, 16, 20)), terrain, **cam)
    got = [f.copy() for f in seq.frames(4, settings, emitters, steps_per_frame=2)]
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    st = seq.stats()
    assert st["shadow_list_chunks"] == (0 if knob.endswith("OOM") else 1)


@pytest.mark.gpu
def test_resident_smoke_sequence_equals_the_host_array_path():
    """Round 4: the sequence with its state, the smoke layer and the terrain frame resident on the GPU (SmokeSequence: only
    the finished RGBA8 frames leave, through two pinned buffers) gives the frames AND the final solver state of the
    host-array path, which crosses the bus with every field for every step and frame."""
    from forge3d_amd import smoke

    w, h = 200, 120
    rng = np.random.default_rng(5)
    terrain = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    terrain[..., 3] = 255
    dims = (24, 16, 20)
    emitters = [smoke.SmokeEmitter(center=(6.0, 3.0, 10.0), radius=2.5, density_rate=6.0, temperature_rate=3.0, soot_rate=0.3, emission_rate=2.0,
                                   velocity=(3.0, 0.4, 0.0))]
    settings = smoke.SmokeStepSettings(dt=0.1, turbulence_strength=0.5, turbulence_seed=7, wind=(1.5, 0.0, -0.2), pressure_iterations=8)
    cam = dict(camera_pos=(12.0, 10.0, 46.0), target=(12.0, 6.0, 10.0))
    host_dom, dev_dom = smoke.SmokeDomain(dims), smoke.SmokeDomain(dims)
    want = smoke.simulate_over_terrain(terrain, host_dom, settings, emitters, 5, steps_per_frame=2, resident=False, **cam)
    seq = smoke.SmokeSequence(dev_dom, terrain, **cam)
    got = [frame.copy() for frame in seq.frames(5, settings, emitters, steps_per_frame=2)]
    assert len(got) == 5
    for f in range(5):
        assert np.array_equal(got[f], want[f]), f
    seq.download()
    assert dev_dom.frame_index == host_dom.frame_index == 10 and dev_dom.time_seconds == host_dom.time_seconds
    for name in ("density", "temperature", "fuel", "soot", "humidity", "emission_rate", "particle_age", "velocity", "pressure"):
        assert np.array_equal(getattr(dev_dom, name), getattr(host_dom, name)), name
    assert all(v == 0.0 for v in seq.kernel_seconds.values())  # untimed calls: the host did not wait for any of them
    for _ in seq.frames(1, settings, emitters, timing=True):
        pass
    assert all(v > 0.0 for v in seq.kernel_seconds.values())
    with pytest.raises(ValueError, match="all host or all device"):
        st = smoke._State()
        for name in smoke._STATE_FIELDS:
            setattr(st, name, seq.state[name].data_ptr())
        st.density = np.zeros(dims[::-1], np.float32).ctypes.data  # one host array among device arrays
        st.dims = (smoke.C.c_uint32 * 3)(*dims)
        st.voxel_size = (smoke.C.c_float * 3)(1.0, 1.0, 1.0)
        st.origin = (smoke.C.c_float * 3)(0.0, 0.0, 0.0)
        err = smoke.C.create_string_buffer(256)
        rc = smoke._native.lib().f3d_smoke_step(smoke.C.byref(st), smoke.C.byref(settings._native()), None, 0, 1, None, err, len(err))
        if rc != 0:
            raise ValueError(err.value.decode())
