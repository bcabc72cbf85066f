"""ctypes front-end of oracle/smoke_oracle.c -- TEST INFRASTRUCTURE ONLY (see the C file's header)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SRC, _LIB = _HERE / "smoke_oracle.c", _HERE / "libsmoke_oracle.so"


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
    if force or not _LIB.exists() or _LIB.stat().st_mtime < _SRC.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", str(_SRC), "-o", str(_LIB), "-lm"],
                       check=True, capture_output=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB))
        _lib.smoke_oracle_sun_transmittance.restype = C.c_float
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
