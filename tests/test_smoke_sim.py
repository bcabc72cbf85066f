"""Smoke transport solver (SmokeVolume::step, reference src/smoke/sim.rs) -- what advances BASELINE.json configs[4]'s
120-frame sequence.  Pins: the reference's own four solver tests (sim.rs:801-899) restated as known-answer properties on
the oracle; the device's per-voxel code compiled for the host (tests/emul) against the oracle bit for bit; and -- with a
GPU -- the HIP kernels against the oracle bit for bit, over every pass (turbulence, MacCormack, vorticity, diffusion,
emitters with start / end times, terrain collision)."""
from __future__ import annotations

import numpy as np
import pytest

from oracle import smoke_oracle as so

FIELDS = so.STATE_FIELDS


def _small():
    return so.new_state((16, 16, 16))


def _centre_of_mass(state, axis):
    d = state["density"].astype(np.float64)
    idx = np.indices(d.shape)[2 - axis]  # arrays are (z, y, x)
    return float((idx * d).sum() / max(d.sum(), 1e-9))


# ---- the reference's solver tests on the oracle (sim.rs:801-899) ---------------------------------------------------------
def test_emitter_adds_required_fields():
    st = _small()
    so.add_emitter(st, dict(center=(8.0, 8.0, 8.0), radius=3.0, density_rate=2.0, temperature_rate=4.0, fuel_rate=1.0), 0.5)
    assert so.mass(st) > 0.0 and st["temperature"].max() > 0.0 and st["fuel"].max() > 0.0 and st["emission_rate"].max() > 0.0


def test_smoke_advects_with_wind_and_preserves_mass():
    st = _small()
    so.add_emitter(st, dict(center=(5.0, 8.0, 8.0), radius=2.0, density_rate=5.0), 1.0)
    st["velocity"][..., 0] = 1.0
    before_mass, before = so.mass(st), _centre_of_mass(st, 0)
    so.step(st, dt=1.0, density_decay=0.0, temperature_decay=0.0, buoyancy=0.0, vorticity=0.0, diffusion=0.0, pressure_iterations=1)
    assert _centre_of_mass(st, 0) > before
    assert abs(so.mass(st) - before_mass) / before_mass < 0.02


def test_buoyant_plume_rises():
    st = _small()
    so.add_emitter(st, dict(center=(8.0, 4.0, 8.0), radius=2.0, density_rate=4.0, temperature_rate=5.0), 1.0)
    before = _centre_of_mass(st, 1)
    so.step(st, steps=4, dt=0.5, density_decay=0.0, temperature_decay=0.0, buoyancy=1.5, vorticity=0.0, diffusion=0.0, pressure_iterations=8)
    assert _centre_of_mass(st, 1) > before and st["frame_index"] == 4 and st["time_seconds"] == pytest.approx(2.0)


def test_pressure_projection_reduces_divergence():
    st = _small()
    z, y, x = np.indices((16, 16, 16))
    inner = (x >= 1) & (x < 15) & (y >= 1) & (y < 15) & (z >= 1) & (z < 15)
    st["velocity"][..., 0][inner] = (x[inner] * 0.03).astype(np.float32)
    st["velocity"][..., 1][inner] = (y[inner] * -0.02).astype(np.float32)
    before = so.divergence_l2(st)
    so.step(st, dt=0.1, density_decay=0.0, temperature_decay=0.0, buoyancy=0.0, vorticity=0.0, diffusion=0.0, pressure_iterations=30)
    assert so.divergence_l2(st) < before


# ---- configurations that visit every pass ------------------------------------------------------------------------------------
def _cases():
    plume = [dict(center=(9.0, 4.0, 11.0), radius=3.0, density_rate=6.0, temperature_rate=5.0, soot_rate=0.6, humidity_rate=0.3, fuel_rate=0.5,
                  velocity=(0.4, 1.5, 0.1)),
             dict(center=(15.0, 5.0, 9.0), radius=2.5, density_rate=3.0, temperature_rate=2.0, start_time=0.15, end_time=0.35)]
    return {
        "defaults": (dict(dims=(20, 14, 18)), plume[:1], dict(), 5),
        "turbulent_wind": (dict(dims=(24, 16, 20), voxel_size=(1.5, 1.0, 2.0), origin=(-3.0, 0.0, 2.0)), plume,
                           dict(dt=0.1, turbulence_strength=0.8, turbulence_seed=31, wind=(1.2, 0.05, -0.7), velocity_damping=0.05, boundary_damping=0.2), 6),
        "maccormack": (dict(dims=(18, 18, 18)), plume, dict(dt=0.2, mac_cormack=True, diffusion=0.01, vorticity=0.4, pressure_iterations=9), 5),
        # more emitters than travel in the kernel arguments (csrc/f3d_smoke_sim.hip kInlineEmitters = 4): the device-copy path
        "six_emitters": (dict(dims=(22, 12, 16)), [dict(center=(4.0 + 2.5 * k, 3.0 + 0.5 * k, 5.0 + 1.5 * k), radius=1.5 + 0.2 * k, density_rate=2.0 + k,
                                                      temperature_rate=1.0 + 0.5 * k, velocity=(0.1 * k, 1.0, -0.05 * k)) for k in range(6)],
                         dict(dt=0.15, pressure_iterations=6), 3),
        # a row longer than a tile of the K-sweeps-a-launch pressure solve (csrc/f3d_smoke_sim.hip k_jacobi_tiled: 96 voxels): tiles
        # with a K-voxel halo in x too, ragged in every axis; 7 sweeps = a launch of 4 and one of 3
        "wide": (dict(dims=(101, 11, 15)), plume[:1], dict(dt=0.2, pressure_iterations=7), 5),
        "bare": (dict(dims=(12, 10, 14)), plume[:1], dict(dt=0.3, diffusion=0.0, vorticity=0.0, velocity_damping=0.0, mass_conservation=False,
                                                          terrain_collision=False, pressure_iterations=1, turbulence_strength=0.3, wind=(0.0, 0.0, 0.0)), 4),
    }


def _run(stepper, case):
    geo, emitters, settings, steps = _cases()[case]
    st = so.new_state(**geo)
    rng = np.random.default_rng(3)
    st["velocity"][...] = rng.normal(scale=0.2, size=st["velocity"].shape).astype(np.float32)
    st["humidity"][...] = rng.random(st["humidity"].shape).astype(np.float32) * 0.1
    return stepper(st, emitters, steps=steps, **settings)


@pytest.mark.parametrize("case", sorted(_cases()))
def test_emulated_device_code_equals_the_oracle(case):
    from emul import emul

    emul.build()
    a, b = _run(so.step, case), _run(emul.smoke_step, case)
    assert a["frame_index"] == b["frame_index"] and np.float32(a["time_seconds"]) == np.float32(b["time_seconds"])
    assert float(a["density"].max()) > 0.05 and np.isfinite(a["velocity"]).all()
    for name in FIELDS:
        assert np.array_equal(a[name], b[name]), (case, name, float(np.abs(a[name] - b[name]).max()))


def test_python_surface_of_the_solver():
    from forge3d_amd import smoke

    with pytest.raises(ValueError, match="dt must be > 0"):
        smoke.SmokeStepSettings(dt=0.0)
    with pytest.raises(ValueError, match="boundary_damping"):
        smoke.SmokeStepSettings(boundary_damping=1.5)
    s = smoke.SmokeStepSettings()
    assert s.pressure_iterations == 20 and s.mass_conservation and not s.mac_cormack and s.dt == pytest.approx(1.0 / 30.0)
    dom = smoke.SmokeDomain((8, 8, 8))
    assert dom.fuel.shape == (8, 8, 8) and dom.pressure.shape == (8, 8, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["fused", "fused-tiled", "fused2", "fused-row-sums", "persistent", "launches"])
@pytest.mark.parametrize("case", sorted(_cases()))
def test_hip_solver_equals_the_oracle(case, form, monkeypatch):
    """All three drivers of the device solver (csrc/f3d_smoke_sim.hip): the step's phases as one launch each (the default),
    as one persistent cooperative launch with grid barriers between them, and the round-3 form of one launch per pass."""
    from forge3d_amd import smoke

    monkeypatch.setenv("F3D_SMOKE_SOLVER", "fused" if form.startswith("fused") else form)
    if form == "fused-row-sums":  # the grid sums a lane a row from global memory (the default stages each slab in LDS)
        monkeypatch.setenv("F3D_SMOKE_ROW_SUMS", "1")
    if form == "fused-tiled":  # K Jacobi sweeps a launch on LDS tiles (measured slower, opt-in)
        monkeypatch.setenv("F3D_SMOKE_JACOBI", "tiled")
    if form == "fused2":  # two Jacobi sweeps a launch (measured slower, opt-in)
        monkeypatch.setenv("F3D_SMOKE_DOUBLE_SWEEPS", "1")
    geo, emitters, settings, steps = _cases()[case]
    want = _run(so.step, case)
    dom = smoke.SmokeDomain(geo["dims"], geo.get("voxel_size", (1.0, 1.0, 1.0)), geo.get("origin", (0.0, 0.0, 0.0)))
    rng = np.random.default_rng(3)
    dom.velocity = rng.normal(scale=0.2, size=dom.velocity.shape).astype(np.float32)
    dom.humidity = (rng.random(dom.humidity.shape).astype(np.float32) * np.float32(0.1)).astype(np.float32)
    half = steps // 2  # two calls: the state survives the round trip
    em = [smoke.SmokeEmitter(**e) for e in emitters]
    dom.step(smoke.SmokeStepSettings(**settings), em, steps=half)
    dom.step(smoke.SmokeStepSettings(**settings), em, steps=steps - half)
    assert dom.frame_index == want["frame_index"] and np.float32(dom.time_seconds) == np.float32(want["time_seconds"])
    for name in FIELDS:
        assert np.array_equal(getattr(dom, name), want[name]), (case, name)


@pytest.mark.gpu
def test_config5_emitters_to_frames():
    """BASELINE.json configs[4] end to end without the reference package: emitters -> solver -> ray-marcher, a few frames of a
    sequence; every frame's state and image equal the oracles'."""
    from forge3d_amd import smoke

    dims = (48, 32, 64)
    dom = smoke.SmokeDomain(dims)
    st = so.new_state(dims)
    emitters = [dict(center=(24.0, 5.0, 18.0), radius=5.0, density_rate=8.0, temperature_rate=6.0, soot_rate=0.5, velocity=(0.0, 2.0, 0.4))]
    settings = dict(dt=0.2, turbulence_strength=0.5, turbulence_seed=7, wind=(0.3, 0.0, 1.0), buoyancy=1.1)
    cam = dict(camera_pos=(24.0, 40.0, -60.0), target=(24.0, 12.0, 30.0))
    for frame in range(4):
        dom.step(smoke.SmokeStepSettings(**settings), [smoke.SmokeEmitter(**e) for e in emitters], steps=3)
        so.step(st, emitters, steps=3, **settings)
        assert np.array_equal(dom.density, st["density"]) and np.array_equal(dom.particle_age, st["particle_age"])
        got = dom.render_rgba(160, 90, fovy_deg=40.0, **cam)
        fields = {k: st[k] for k in ("density", "temperature", "soot", "humidity", "emission_rate", "particle_age")}
        want = so.render_rgba(fields, 160, 90, frame_index=st["frame_index"], fovy_deg=40.0, **cam)
        assert np.array_equal(got, want), frame
    assert int(got[..., 3].max()) > 40
