"""ctypes front-end of oracle/smoke_oracle.c -- TEST INFRASTRUCTURE ONLY (see the C file's header)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SRC, _LIB = _HERE / "smoke_oracle.c", _HERE / "libsmoke_oracle.so"
_SIM_SRC = _HERE / "smoke_sim_oracle.c"  # the transport solver (SmokeVolume::step)
_COMPOSITE_SRC = _HERE / "composite_oracle.c"  # smoke over terrain (the example's numpy / Pillow composites)


class Volume(C.Structure):
    _fields_ = [("density", C.c_void_p), ("temperature", C.c_void_p), ("soot", C.c_void_p), ("humidity", C.c_void_p),
                ("emission", C.c_void_p), ("age", C.c_void_p), ("dims", C.c_uint32 * 3), ("voxel_size", C.c_float * 3),
                ("origin", C.c_float * 3), ("frame_index", C.c_uint32)]


class Settings(C.Structure):
    _fields_ = [("density_scale", C.c_float), ("extinction", C.c_float), ("scattering", C.c_float),
                ("absorption", C.c_float), ("phase_g", C.c_float), ("step_size", C.c_float), ("max_steps", C.c_uint32),
                ("self_shadow", C.c_int32), ("shadow_steps", C.c_uint32), ("shadow_step_size", C.c_float),
                ("jitter_strength", C.c_float), ("exposure", C.c_float), ("thin_color", C.c_float * 3),
                ("dense_color", C.c_float * 3), ("soot_absorption", C.c_float), ("fire_glow", C.c_float)]


DEFAULTS = dict(density_scale=1.0, extinction=2.6, scattering=0.85, absorption=0.45, phase_g=0.24, step_size=0.0,
                max_steps=256, self_shadow=True, shadow_steps=20, shadow_step_size=0.0, jitter_strength=0.5, exposure=1.0,
                thin_color=(0.50, 0.54, 0.58), dense_color=(0.93, 0.91, 0.82), soot_absorption=0.22, fire_glow=0.35)


def build(force: bool = False) -> Path:
    if force or not _LIB.exists() or _LIB.stat().st_mtime < max(f.stat().st_mtime for f in (_SRC, _SIM_SRC, _COMPOSITE_SRC)):
        import os

        tmp = _LIB.with_suffix(f".{os.getpid()}.tmp")
        subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", str(_SRC), str(_SIM_SRC), str(_COMPOSITE_SRC), "-o", str(tmp), "-lm"],
                       check=True, capture_output=True)
        os.replace(tmp, _LIB)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB))
        _lib.smoke_oracle_sun_transmittance.restype = C.c_float
        _lib.smoke_sim_mass.restype = C.c_float
        _lib.smoke_sim_divergence_l2.restype = C.c_float
    return _lib


def _settings(**kw) -> Settings:
    v = {**DEFAULTS, **kw}
    s = Settings()
    for name, value in v.items():
        if name in ("thin_color", "dense_color"):
            setattr(s, name, (C.c_float * 3)(*value))
        elif name == "self_shadow":
            s.self_shadow = 1 if value else 0
        else:
            setattr(s, name, value)
    return s


def _volume(fields: dict, voxel_size, origin, frame_index):
    """fields: density / temperature / soot / humidity / emission_rate / particle_age, (nz, ny, nx) f32 arrays
    (missing ones default to zeros, age to -1 like SmokeVolume::new)."""
    density = np.ascontiguousarray(fields["density"], np.float32)
    nz, ny, nx = density.shape
    keep = []
    v = Volume()
    for attr, key, fill in (("density", "density", 0.0), ("temperature", "temperature", 0.0), ("soot", "soot", 0.0),
                            ("humidity", "humidity", 0.0), ("emission", "emission_rate", 0.0), ("age", "particle_age", -1.0)):
        arr = np.ascontiguousarray(fields[key], np.float32) if key in fields else np.full(density.shape, fill, np.float32)
        keep.append(arr)
        setattr(v, attr, arr.ctypes.data)
    v.dims = (C.c_uint32 * 3)(nx, ny, nz)
    v.voxel_size = (C.c_float * 3)(*voxel_size)
    v.origin = (C.c_float * 3)(*origin)
    v.frame_index = int(frame_index) & 0xFFFFFFFF
    return v, keep


def render_rgba(fields, width, height, camera_pos, target, up=(0.0, 1.0, 0.0), fovy_deg=45.0, sun_direction=(0.4, 0.8, -0.2),
                voxel_size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), frame_index=0, **settings):
    v, keep = _volume(fields, voxel_size, origin, frame_index)
    s = _settings(**settings)
    out = np.zeros((height, width, 4), np.uint8)
    err = C.create_string_buffer(256)
    f3 = lambda t: (C.c_float * 3)(*t)  # noqa: E731
    rc = lib().smoke_oracle_raymarch_rgba(C.byref(v), C.c_uint32(width), C.c_uint32(height), f3(camera_pos), f3(target), f3(up),
                                          C.c_float(fovy_deg), f3(sun_direction), C.byref(s), C.c_void_p(out.ctypes.data), err,
                                          C.c_size_t(len(err)))
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return out


def render_projection_rgba(fields, width, height, view_direction=(0.0, -1.0, 0.0), sun_direction=(0.4, 0.8, -0.2),
                           voxel_size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), frame_index=0, **settings):
    v, keep = _volume(fields, voxel_size, origin, frame_index)
    s = _settings(**settings)
    out = np.zeros((height, width, 4), np.uint8)
    err = C.create_string_buffer(256)
    f3 = lambda t: (C.c_float * 3)(*t)  # noqa: E731
    rc = lib().smoke_oracle_raymarch_projection_rgba(C.byref(v), C.c_uint32(width), C.c_uint32(height), f3(view_direction),
                                                     f3(sun_direction), C.byref(s), C.c_void_p(out.ctypes.data), err,
                                                     C.c_size_t(len(err)))
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return out


def sun_transmittance(fields, start, sun_dir, step, steps, voxel_size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), **settings):
    v, keep = _volume(fields, voxel_size, origin, 0)
    s = _settings(**settings)
    f3 = lambda t: (C.c_float * 3)(*t)  # noqa: E731
    return float(lib().smoke_oracle_sun_transmittance(C.byref(v), f3(start), f3(sun_dir), C.c_float(step), C.c_uint32(steps),
                                                      C.byref(s)))


# ---- the transport solver (oracle/smoke_sim_oracle.c) --------------------------------------------------------------------
STATE_FIELDS = ("density", "temperature", "fuel", "soot", "humidity", "emission_rate", "particle_age", "velocity", "pressure")
STEP_DEFAULTS = dict(dt=1.0 / 30.0, density_decay=0.015, temperature_decay=0.08, velocity_damping=0.01, diffusion=0.0005, buoyancy=0.7,
                     vorticity=0.12, pressure_iterations=20, turbulence_strength=0.0, turbulence_seed=0, mac_cormack=False,
                     mass_conservation=True, terrain_collision=True, boundary_damping=0.0, wind=(0.0, 0.0, 0.0))  # types.rs:160-180
EMITTER_DEFAULTS = dict(center=(0.0, 0.0, 0.0), radius=1.0, density_rate=1.0, temperature_rate=1.0, fuel_rate=0.0, soot_rate=0.2,
                        humidity_rate=0.0, emission_rate=1.0, velocity=(0.0, 1.0, 0.0), start_time=0.0, end_time=3.4028234663852886e38)


class SimVolume(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in STATE_FIELDS] + [("dims", C.c_uint32 * 3), ("voxel_size", C.c_float * 3), ("origin", C.c_float * 3),
                                                                ("sparse_threshold", C.c_float), ("time_seconds", C.c_float), ("frame_index", C.c_uint32)]


class SimSettings(C.Structure):
    _fields_ = [("dt", C.c_float), ("density_decay", C.c_float), ("temperature_decay", C.c_float), ("velocity_damping", C.c_float),
                ("diffusion", C.c_float), ("buoyancy", C.c_float), ("vorticity", C.c_float), ("pressure_iterations", C.c_uint32),
                ("turbulence_strength", C.c_float), ("turbulence_seed", C.c_uint32), ("mac_cormack", C.c_int32), ("mass_conservation", C.c_int32),
                ("terrain_collision", C.c_int32), ("boundary_damping", C.c_float), ("wind", C.c_float * 3)]


class SimEmitter(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("density_rate", C.c_float), ("temperature_rate", C.c_float), ("fuel_rate", C.c_float),
                ("soot_rate", C.c_float), ("humidity_rate", C.c_float), ("emission_rate", C.c_float), ("velocity", C.c_float * 3),
                ("start_time", C.c_float), ("end_time", C.c_float)]


def new_state(dims, voxel_size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), sparse_threshold=1.0e-5):
    """SmokeVolume::new (types.rs:318-357): zeros, particle age -1; dims = (nx, ny, nz), fields shaped (nz, ny, nx)."""
    nx, ny, nz = (int(d) for d in dims)
    st = {name: np.zeros((nz, ny, nx) + ((3,) if name == "velocity" else ()), np.float32) for name in STATE_FIELDS}
    st["particle_age"][...] = -1.0
    st.update(dims=(nx, ny, nz), voxel_size=tuple(float(v) for v in voxel_size), origin=tuple(float(v) for v in origin),
              sparse_threshold=float(sparse_threshold), time_seconds=0.0, frame_index=0)
    return st


def sim_structs(state, cls=SimVolume):
    v = cls()
    keep = []
    for name in STATE_FIELDS:
        arr = state[name]
        assert arr.dtype == np.float32 and arr.flags["C_CONTIGUOUS"], name
        keep.append(arr)
        setattr(v, name, arr.ctypes.data)
    v.dims = (C.c_uint32 * 3)(*state["dims"])
    v.voxel_size = (C.c_float * 3)(*state["voxel_size"])
    v.origin = (C.c_float * 3)(*state["origin"])
    v.sparse_threshold, v.time_seconds, v.frame_index = state["sparse_threshold"], state["time_seconds"], int(state["frame_index"])
    return v, keep


def settings_struct(cls=SimSettings, **kw):
    vals = {**STEP_DEFAULTS, **kw}
    s = cls()
    for name, value in vals.items():
        if name == "wind":
            s.wind = (C.c_float * 3)(*value)
        elif name in ("mac_cormack", "mass_conservation", "terrain_collision"):
            setattr(s, name, 1 if value else 0)
        else:
            setattr(s, name, value)
    return s


def emitter_array(emitters, cls=SimEmitter):
    arr = (cls * max(1, len(emitters)))()
    for dst, e in zip(arr, emitters):
        vals = {**EMITTER_DEFAULTS, **e}
        for name, value in vals.items():
            setattr(dst, name, (C.c_float * 3)(*value) if name in ("center", "velocity") else value)
    return arr


def step(state, emitters=(), steps=1, **settings):
    """SmokeVolume::step x steps, in place on `state` (a new_state() dict)."""
    v, keep = sim_structs(state)
    s = settings_struct(**settings)
    em = emitter_array(list(emitters))
    for _ in range(int(steps)):
        lib().smoke_sim_step(C.byref(v), C.byref(s), em, C.c_uint32(len(emitters)))
    state["time_seconds"], state["frame_index"] = float(v.time_seconds), int(v.frame_index)
    return state


def add_emitter(state, emitter, dt):
    v, keep = sim_structs(state)
    lib().smoke_sim_add_emitter(C.byref(v), emitter_array([emitter]), C.c_float(dt))


def mass(state):
    v, keep = sim_structs(state)
    return float(lib().smoke_sim_mass(C.byref(v)))


def divergence_l2(state):
    v, keep = sim_structs(state)
    return float(lib().smoke_sim_divergence_l2(C.byref(v)))


# ---- smoke over terrain (composite_oracle.c) ----
def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 4
    return a


def composite_atmospheric(base, smoke) -> np.ndarray:
    base, smoke = _img(base), _img(smoke)
    assert base.shape == smoke.shape
    out = np.empty_like(base)
    lib().composite_oracle_atmospheric(C.c_void_p(base.ctypes.data), C.c_void_p(smoke.ctypes.data), C.c_uint32(base.shape[1]), C.c_uint32(base.shape[0]),
                                       C.c_void_p(out.ctypes.data))
    return out


def composite_smoke_maps(atmospheric, physical=None, atmospheric_alpha=0.42, physical_alpha=0.92, max_alpha=168) -> np.ndarray:
    atmospheric = _img(atmospheric)
    physical = None if physical is None else _img(physical)
    out = np.empty_like(atmospheric)
    lib().composite_oracle_smoke_maps(C.c_void_p(atmospheric.ctypes.data), C.c_void_p(physical.ctypes.data if physical is not None else None),
                                      C.c_uint32(atmospheric.shape[1]), C.c_uint32(atmospheric.shape[0]), C.c_float(atmospheric_alpha),
                                      C.c_float(physical_alpha), C.c_uint32(max_alpha), C.c_void_p(out.ctypes.data))
    return out


def composite_over(base, layer, offset=(0, 0)) -> np.ndarray:
    base, layer = _img(base), _img(layer)
    out = np.empty_like(base)
    lib().composite_oracle_over(C.c_void_p(base.ctypes.data), C.c_uint32(base.shape[1]), C.c_uint32(base.shape[0]), C.c_void_p(layer.ctypes.data),
                                C.c_uint32(layer.shape[1]), C.c_uint32(layer.shape[0]), C.c_int32(int(offset[0])), C.c_int32(int(offset[1])),
                                C.c_void_p(out.ctypes.data))
    return out
