"""Smoke volumes rendered on the GPU: the ``forge3d.smoke`` surface of the reference for its ray-marcher.

Reference: ``SmokeDomain`` / ``SmokeRenderSettings`` / ``SmokeEmitter`` of the compiled extension
(src/smoke/py.rs:11-625, re-exported by python/forge3d/smoke.py:62-66) with
``render_rgba`` = ``SmokeVolume::raymarch_rgba`` and ``render_projection_rgba`` =
``SmokeVolume::raymarch_projection_rgba`` (src/smoke/render.rs:7-178).  Same constructor signatures, array
layouts ((z, y, x) NumPy arrays, x fastest), validation messages and exception types; the ray-march itself runs
in libf3dhip.so (csrc/f3d_smoke.hip), one lane per pixel.  The transport solver (``SmokeDomain.step``,
src/smoke/sim.rs) is NOT part of the offline render path this package replaces: it raises.

``render_sequence`` spreads the frames of an animation over the ranks of a torch.distributed job (frames are
independent: BASELINE.json configs[4] "smoke sequence ... x 120 frames, 8 x MI355X" = frame replicas).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, fields
from typing import Iterable, Sequence

import numpy as np

from . import _native

MAX_VOXELS = 256 * 256 * 256  # MAX_CPU_VOXELS, src/smoke/types.rs:4


class _Volume(C.Structure):
    _fields_ = [("density", C.c_void_p), ("temperature", C.c_void_p), ("soot", C.c_void_p), ("humidity", C.c_void_p),
                ("emission", C.c_void_p), ("age", C.c_void_p), ("dims", C.c_uint32 * 3), ("voxel_size", C.c_float * 3),
                ("origin", C.c_float * 3), ("frame_index", C.c_uint32)]


class _View(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("projection", C.c_int32), ("camera_pos", C.c_float * 3),
                ("target", C.c_float * 3), ("up", C.c_float * 3), ("fovy_deg", C.c_float),
                ("view_direction", C.c_float * 3), ("sun_direction", C.c_float * 3)]


class _Settings(C.Structure):
    _fields_ = [("density_scale", C.c_float), ("extinction", C.c_float), ("scattering", C.c_float),
                ("absorption", C.c_float), ("phase_g", C.c_float), ("step_size", C.c_float), ("max_steps", C.c_uint32),
                ("self_shadow", C.c_int32), ("shadow_steps", C.c_uint32), ("shadow_step_size", C.c_float),
                ("jitter_strength", C.c_float), ("exposure", C.c_float), ("thin_color", C.c_float * 3),
                ("dense_color", C.c_float * 3), ("soot_absorption", C.c_float), ("fire_glow", C.c_float)]


def _f3(v, name="value"):
    t = tuple(float(x) for x in v)
    if len(t) != 3:
        raise ValueError(f"{name} must have three components")
    return t


@dataclass
class SmokeRenderSettings:
    """src/smoke/py.rs:276-333 (defaults src/smoke/types.rs:247-268); invalid values raise ValueError."""
    density_scale: float = 1.0
    extinction: float = 2.6
    scattering: float = 0.85
    absorption: float = 0.45
    phase_g: float = 0.24
    step_size: float = 0.0
    max_steps: int = 256
    self_shadow: bool = True
    shadow_steps: int = 20
    shadow_step_size: float = 0.0
    jitter_strength: float = 0.5
    exposure: float = 1.0
    thin_color: tuple = (0.50, 0.54, 0.58)
    dense_color: tuple = (0.93, 0.91, 0.82)
    soot_absorption: float = 0.22
    fire_glow: float = 0.35

    def __post_init__(self):
        self.thin_color, self.dense_color = _f3(self.thin_color, "thin_color"), _f3(self.dense_color, "dense_color")
        problem = self.problem()
        if problem:
            raise ValueError(problem)

    def problem(self) -> "str | None":
        """SmokeRenderSettings::validate (types.rs:271-316): the first violated rule, or None."""
        scalars = ("density_scale", "extinction", "scattering", "absorption", "phase_g", "step_size", "shadow_step_size",
                   "jitter_strength", "exposure", "soot_absorption", "fire_glow")
        for name in scalars:
            if not math.isfinite(float(getattr(self, name))):
                return f"{name} must be finite"
        if min(self.density_scale, self.extinction, self.scattering) < 0.0:
            return "density_scale, extinction, and scattering must be >= 0"
        if min(self.absorption, self.soot_absorption, self.fire_glow) < 0.0:
            return "absorption, soot_absorption, and fire_glow must be >= 0"
        if not -0.99 <= self.phase_g <= 0.99:
            return "phase_g must be in [-0.99, 0.99]"
        if self.step_size < 0.0 or self.shadow_step_size < 0.0:
            return "step sizes must be >= 0"
        if int(self.max_steps) == 0 or int(self.shadow_steps) == 0:
            return "max_steps and shadow_steps must be >= 1"
        if not 0.0 <= self.jitter_strength <= 1.0:
            return "jitter_strength must be in [0, 1]"
        for name in ("thin_color", "dense_color"):
            for axis, value in enumerate(getattr(self, name)):
                if not math.isfinite(value) or value < 0.0:
                    return f"{name}[{axis}] must be finite and >= 0"
        return None

    def _native(self) -> _Settings:
        s = _Settings()
        for f in fields(self):
            v = getattr(self, f.name)
            if f.name in ("thin_color", "dense_color"):
                setattr(s, f.name, (C.c_float * 3)(*v))
            elif f.name in ("max_steps", "shadow_steps"):
                setattr(s, f.name, int(v))
            elif f.name == "self_shadow":
                s.self_shadow = 1 if v else 0
            else:
                setattr(s, f.name, float(v))
        return s

    def __repr__(self):
        return f"SmokeRenderSettings(extinction={self.extinction}, phase_g={self.phase_g}, max_steps={self.max_steps})"


_STATE_FIELDS = ("density", "temperature", "fuel", "soot", "humidity", "emission_rate", "particle_age", "velocity", "pressure")


class _State(C.Structure):
    """f3d_smoke_state"""
    _fields_ = [(name, C.c_void_p) for name in _STATE_FIELDS] + [("dims", C.c_uint32 * 3), ("voxel_size", C.c_float * 3), ("origin", C.c_float * 3),
                                                                 ("sparse_threshold", C.c_float), ("time_seconds", C.c_float), ("frame_index", C.c_uint32)]


class _StepSettings(C.Structure):
    """f3d_smoke_step_settings"""
    _fields_ = [("dt", C.c_float), ("density_decay", C.c_float), ("temperature_decay", C.c_float), ("velocity_damping", C.c_float),
                ("diffusion", C.c_float), ("buoyancy", C.c_float), ("vorticity", C.c_float), ("pressure_iterations", C.c_uint32),
                ("turbulence_strength", C.c_float), ("turbulence_seed", C.c_uint32), ("mac_cormack", C.c_int32), ("mass_conservation", C.c_int32),
                ("terrain_collision", C.c_int32), ("boundary_damping", C.c_float), ("wind", C.c_float * 3)]


class _Emitter(C.Structure):
    """f3d_smoke_emitter"""
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("density_rate", C.c_float), ("temperature_rate", C.c_float), ("fuel_rate", C.c_float),
                ("soot_rate", C.c_float), ("humidity_rate", C.c_float), ("emission_rate", C.c_float), ("velocity", C.c_float * 3),
                ("start_time", C.c_float), ("end_time", C.c_float)]


@dataclass
class SmokeStepSettings:
    """SmokeStepSettings (reference src/smoke/types.rs:142-226, Python class src/smoke/py.rs:195-265): defaults and validation."""
    dt: float = 1.0 / 30.0
    density_decay: float = 0.015
    temperature_decay: float = 0.08
    velocity_damping: float = 0.01
    diffusion: float = 0.0005
    buoyancy: float = 0.7
    vorticity: float = 0.12
    pressure_iterations: int = 20
    turbulence_strength: float = 0.0
    turbulence_seed: int = 0
    mac_cormack: bool = False
    mass_conservation: bool = True
    terrain_collision: bool = True
    boundary_damping: float = 0.0
    wind: tuple = (0.0, 0.0, 0.0)

    def __post_init__(self):
        for name in ("dt", "density_decay", "temperature_decay", "velocity_damping", "diffusion", "buoyancy", "vorticity", "turbulence_strength",
                     "boundary_damping"):
            if not math.isfinite(getattr(self, name)):
                raise ValueError(f"{name} must be finite")
        if self.dt <= 0.0:
            raise ValueError("dt must be > 0")
        if min(self.density_decay, self.temperature_decay, self.velocity_damping, self.diffusion, self.vorticity, self.turbulence_strength) < 0.0:
            raise ValueError("decay, damping, diffusion, vorticity, and turbulence must be >= 0")
        if not 0.0 <= self.boundary_damping <= 1.0:
            raise ValueError("boundary_damping must be in [0, 1]")
        self.wind = _f3(self.wind, "wind")

    def _native(self) -> _StepSettings:
        s = _StepSettings()
        for name, _t in _StepSettings._fields_:
            v = getattr(self, name)
            if name == "wind":
                s.wind = (C.c_float * 3)(*v)
            elif name in ("mac_cormack", "mass_conservation", "terrain_collision"):
                setattr(s, name, 1 if v else 0)
            elif name in ("pressure_iterations", "turbulence_seed"):
                setattr(s, name, int(v))
            else:
                setattr(s, name, float(v))
        return s


@dataclass
class SmokeEmitter:
    """src/smoke/py.rs:20-61 (defaults types.rs:85-101)."""
    center: tuple = (0.0, 0.0, 0.0)
    radius: float = 1.0
    density_rate: float = 1.0
    temperature_rate: float = 1.0
    fuel_rate: float = 0.0
    soot_rate: float = 0.2
    humidity_rate: float = 0.0
    emission_rate: float = 1.0
    velocity: tuple = (0.0, 1.0, 0.0)
    start_time: float = 0.0
    end_time: float = 3.4028234663852886e38

    def __post_init__(self):
        self.center, self.velocity = _f3(self.center, "center"), _f3(self.velocity, "velocity")
        if not (math.isfinite(self.radius) and self.radius > 0.0):
            raise ValueError("radius must be finite and > 0")
        if self.end_time < self.start_time:
            raise ValueError("end_time must be >= start_time")


class SmokeDomain:
    """Dense smoke state + GPU ray-marcher (reference SmokeDomain, src/smoke/py.rs:342-641).

    Fields are float32 arrays of shape (nz, ny, nx) -- ``dims`` is (nx, ny, nz) like the reference's."""

    FIELDS = ("density", "temperature", "soot", "humidity", "emission_rate", "particle_age")

    def __init__(self, dims, voxel_size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), brick_size=(16, 16, 16),
                 sparse_threshold=1.0e-5):
        self._dims = tuple(int(d) for d in dims)
        self._voxel, self._origin = _f3(voxel_size, "voxel_size"), _f3(origin, "origin")
        for axis, d in enumerate(self._dims):
            if d < 2:
                raise ValueError(f"dims[{axis}] must be >= 2")
        n = self._dims[0] * self._dims[1] * self._dims[2]
        if n > MAX_VOXELS:
            raise ValueError(f"smoke domain has {n} voxels, exceeding CPU reference limit {MAX_VOXELS}")
        for axis, v in enumerate(self._voxel):
            if not (math.isfinite(v) and v > 0.0):
                raise ValueError(f"voxel_size[{axis}] must be finite and > 0")
        for axis, v in enumerate(self._origin):
            if not math.isfinite(v):
                raise ValueError(f"origin[{axis}] must be finite")
        if any(int(b) == 0 for b in brick_size):
            raise ValueError(f"brick_size[{[int(b) for b in brick_size].index(0)}] must be >= 1")
        if not (math.isfinite(sparse_threshold) and sparse_threshold >= 0.0):
            raise ValueError("sparse_threshold must be finite and >= 0")
        self.sparse_threshold = float(sparse_threshold)
        shape = (self._dims[2], self._dims[1], self._dims[0])
        self.density = np.zeros(shape, np.float32)
        self.temperature = np.zeros(shape, np.float32)
        self.soot = np.zeros(shape, np.float32)
        self.fuel = np.zeros(shape, np.float32)
        self.pressure = np.zeros(shape, np.float32)
        self.humidity = np.zeros(shape, np.float32)
        self.emission_rate = np.zeros(shape, np.float32)
        self.particle_age = np.full(shape, -1.0, np.float32)
        self.velocity = np.zeros(shape + (3,), np.float32)
        self.time_seconds = 0.0
        self.frame_index = 0
        self.last_kernel_seconds = 0.0

    # -- construction / state ---------------------------------------------------------------------------
    @staticmethod
    def from_density(density, voxel_size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0)) -> "SmokeDomain":
        arr = np.asarray(density)
        if arr.ndim != 3:
            raise ValueError("density must be a 3-D float32 array (z, y, x)")
        dom = SmokeDomain((arr.shape[2], arr.shape[1], arr.shape[0]), voxel_size, origin)
        dom.set_density(arr)
        return dom

    dims = property(lambda self: self._dims)
    voxel_size = property(lambda self: self._voxel)
    origin = property(lambda self: self._origin)

    def _checked(self, value, name):
        arr = np.ascontiguousarray(value, dtype=np.float32)
        if arr.shape != self.density.shape:
            raise ValueError(f"{name} shape must be (z={self._dims[2]}, y={self._dims[1]}, x={self._dims[0]})")
        if not np.isfinite(arr).all():
            raise ValueError(f"{name} contains non-finite values")
        return arr

    def set_density(self, density) -> None:
        """SmokeVolume::set_density (types.rs:489-511): also restarts the age of every occupied voxel."""
        self.density = self._checked(density, "density")
        self.particle_age = np.where(self.density > self.sparse_threshold, 0.0, -1.0).astype(np.float32)

    def set_velocity(self, velocity) -> None:
        arr = np.ascontiguousarray(velocity, dtype=np.float32)
        if arr.shape != self.density.shape + (3,):
            raise ValueError(f"velocity shape must be (z={self._dims[2]}, y={self._dims[1]}, x={self._dims[0]}, 3)")
        if not np.isfinite(arr).all():
            raise ValueError("velocity contains non-finite values")
        self.velocity = arr

    def set_temperature(self, v) -> None:
        self.temperature = self._checked(v, "temperature")

    def set_soot(self, v) -> None:
        self.soot = self._checked(v, "soot")

    def set_humidity(self, v) -> None:
        self.humidity = self._checked(v, "humidity")

    def set_emission(self, v) -> None:
        self.emission_rate = self._checked(v, "emission")

    def set_particle_age(self, v) -> None:
        self.particle_age = self._checked(v, "particle_age")

    def add_emitter(self, emitter: SmokeEmitter, dt: float) -> None:
        """SmokeVolume::add_emitter (src/smoke/sim.rs:7-45): deposit a smooth ball of smoke, f32 arithmetic."""
        if not (math.isfinite(dt) and dt > 0.0):
            raise ValueError("dt must be finite and > 0")
        f = np.float32
        nx, ny, nz = self._dims
        z, y, x = np.meshgrid(np.arange(nz, dtype=np.float32), np.arange(ny, dtype=np.float32),
                              np.arange(nx, dtype=np.float32), indexing="ij")
        px = f(self._origin[0]) + (x + f(0.5)) * f(self._voxel[0])
        py = f(self._origin[1]) + (y + f(0.5)) * f(self._voxel[1])
        pz = f(self._origin[2]) + (z + f(0.5)) * f(self._voxel[2])
        dx, dy, dz = px - f(emitter.center[0]), py - f(emitter.center[1]), pz - f(emitter.center[2])
        d = np.sqrt(dx * dx + dy * dy + dz * dz, dtype=np.float32)
        radius = f(max(emitter.radius, 1.0e-6))
        inside = ~(d > radius)
        t = np.clip(d / np.maximum(radius, f(1.0e-6)), f(0.0), f(1.0)).astype(np.float32)
        falloff = (f(1.0) - t * t * (f(3.0) - f(2.0) * t)).astype(np.float32)
        amount = (f(dt) * falloff).astype(np.float32)
        for name, rate in (("density", emitter.density_rate), ("temperature", emitter.temperature_rate), ("fuel", emitter.fuel_rate),
                           ("soot", emitter.soot_rate), ("humidity", emitter.humidity_rate)):
            field = getattr(self, name)
            field[inside] = np.maximum(field[inside] + f(rate) * amount[inside], f(0.0))
        self.emission_rate[inside] += f(emitter.emission_rate) * falloff[inside]
        self.particle_age[inside] = 0.0
        for c in range(3):
            self.velocity[..., c][inside] += f(emitter.velocity[c]) * amount[inside]

    def step(self, settings: "SmokeStepSettings | None" = None, emitters=None, steps: int = 1) -> None:
        """SmokeVolume::step (reference src/smoke/sim.rs:47-139, Python SmokeDomain.step, py.rs:466-480) x `steps`, on the
        GPU (csrc/f3d_smoke_sim.hip: one launch per pass, a lane per voxel): emitters, forces, advection, diffusion,
        vorticity confinement, pressure projection, boundaries, sub-grid eddies, decay and ageing.  The state is updated
        in place; `last_kernel_seconds` holds the device time."""
        settings = settings or SmokeStepSettings()
        emitters = list(emitters or [])
        if int(steps) < 1 or int(steps) > 1_000_000:  # (a negative count would wrap to ~4e9 solver steps in the C ABI's u32)
            raise ValueError(f"steps must be in 1..=1000000, got {steps}")
        st = _State()
        keep = []
        for name in _STATE_FIELDS:
            arr = np.ascontiguousarray(getattr(self, name), dtype=np.float32)
            setattr(self, name, arr)
            keep.append(arr)
            setattr(st, name, arr.ctypes.data)
        st.dims = (C.c_uint32 * 3)(*self._dims)
        st.voxel_size = (C.c_float * 3)(*self._voxel)
        st.origin = (C.c_float * 3)(*self._origin)
        st.sparse_threshold, st.time_seconds, st.frame_index = float(self.sparse_threshold), float(self.time_seconds), int(self.frame_index) & 0xFFFFFFFF
        em = (_Emitter * max(1, len(emitters)))()
        for dst, e in zip(em, emitters):
            for name, _t in _Emitter._fields_:
                v = getattr(e, name)
                setattr(dst, name, (C.c_float * 3)(*v) if name in ("center", "velocity") else float(v))
        err = C.create_string_buffer(512)
        seconds = C.c_double(0.0)
        rc = _native.lib().f3d_smoke_step(C.byref(st), C.byref(settings._native()), em, C.c_uint32(len(emitters)), C.c_uint32(int(steps)),
                                          C.byref(seconds), err, len(err))
        if rc != 0:
            message = err.value.decode("utf-8", "replace")
            raise (ValueError if rc == _native.STATUS_VALUE else RuntimeError)(message)
        self.time_seconds, self.frame_index = float(st.time_seconds), int(st.frame_index)
        self.last_kernel_seconds = float(seconds.value)

    def mass(self) -> float:
        """SmokeVolume::mass (types.rs:407-409): the f32 sum of the densities in storage order, as the reference forms it
        (`iter().sum::<f32>()`).  (The SOLVER's mass conservation sums rows, then slabs, then the total -- f3d_smoke_sim.h,
        mirrored by oracle/smoke_sim_oracle.c -- so a mass-conserved density equals the oracle's bit for bit but may differ
        from reference sim.rs in the last bits of the scale factor: DESIGN.md 9.5.)"""
        return float(np.cumsum(np.ascontiguousarray(self.density, np.float32).ravel(), dtype=np.float32)[-1]) if self.density.size else 0.0

    def to_density_numpy(self) -> np.ndarray:
        return self.density.copy()

    def to_velocity_numpy(self) -> np.ndarray:
        return self.velocity.copy()

    def to_particle_age_numpy(self) -> np.ndarray:
        return self.particle_age.copy()

    # -- rendering ----------------------------------------------------------------------------------------
    def _render(self, view: _View, settings) -> np.ndarray:
        settings = settings or SmokeRenderSettings()
        problem = settings.problem()
        if problem:
            raise RuntimeError(problem)  # the reference validates again inside the ray-marcher (render.rs:19)
        keep = [np.ascontiguousarray(getattr(self, name), dtype=np.float32) for name in self.FIELDS]
        vol = _Volume()
        vol.density, vol.temperature, vol.soot, vol.humidity, vol.emission, vol.age = (a.ctypes.data for a in keep)
        vol.dims = (C.c_uint32 * 3)(*self._dims)
        vol.voxel_size = (C.c_float * 3)(*self._voxel)
        vol.origin = (C.c_float * 3)(*self._origin)
        vol.frame_index = int(self.frame_index) & 0xFFFFFFFF
        out = np.zeros((int(view.height), int(view.width), 4), np.uint8)
        err = C.create_string_buffer(512)
        seconds = C.c_double(0.0)
        native = settings._native()
        rc = _native.lib().f3d_smoke_render(C.byref(vol), C.byref(view), C.byref(native), out.ctypes.data,
                                            C.byref(seconds), err, len(err))
        if rc != 0:
            message = err.value.decode("utf-8", "replace")
            raise (ValueError if rc == _native.STATUS_VALUE else RuntimeError)(message)
        self.last_kernel_seconds = float(seconds.value)
        return out

    def render_rgba(self, width, height, camera_pos, target, up=(0.0, 1.0, 0.0), fovy_deg=45.0,
                    sun_direction=(0.4, 0.8, -0.2), settings=None, certificate=None, cache=None) -> np.ndarray:
        """Perspective ray-march, (height, width, 4) uint8: straight colour + alpha = 1 - transmittance
        (reference SmokeDomain.render_rgba, src/smoke/py.rs:531-584).  certificate / cache: accepted, ignored."""
        if int(width) < 1 or int(height) < 1:
            raise RuntimeError("width and height must be >= 1")
        view = _View(int(width), int(height), 0)
        view.camera_pos, view.target, view.up = ((C.c_float * 3)(*_f3(v)) for v in (camera_pos, target, up))
        view.fovy_deg = float(fovy_deg)
        view.sun_direction = (C.c_float * 3)(*_f3(sun_direction))
        return self._render(view, settings)

    def render_projection_rgba(self, width, height, view_direction=(0.0, -1.0, 0.0), sun_direction=(0.4, 0.8, -0.2),
                               settings=None, certificate=None, cache=None) -> np.ndarray:
        """Map-aligned parallel projection (reference render_projection_rgba, src/smoke/py.rs:586-625)."""
        if int(width) < 1 or int(height) < 1:
            raise RuntimeError("width and height must be >= 1")
        view = _View(int(width), int(height), 1)
        view.view_direction = (C.c_float * 3)(*_f3(view_direction))
        view.sun_direction = (C.c_float * 3)(*_f3(sun_direction))
        return self._render(view, settings)

    def __repr__(self):
        return f"SmokeDomain(dims={list(self._dims)}, time_seconds={self.time_seconds:.3f}, frame_index={self.frame_index})"


def domain_from_density(density, voxel_size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0)) -> SmokeDomain:
    """reference python/forge3d/smoke.py:571-578"""
    return SmokeDomain.from_density(density, voxel_size, origin)


class _CompositeDesc(C.Structure):
    """f3d_composite_desc"""
    _fields_ = [("struct_size", C.c_uint32), ("mode", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("layer_width", C.c_uint32),
                ("layer_height", C.c_uint32), ("offset_x", C.c_int32), ("offset_y", C.c_int32), ("base", C.c_void_p), ("layer", C.c_void_p),
                ("base_alpha", C.c_float), ("layer_alpha", C.c_float), ("max_alpha", C.c_uint32)]


COMPOSITE_ATMOSPHERIC, COMPOSITE_SMOKE_MAPS, COMPOSITE_OVER = 0, 1, 2
HYBRID_SMOKE_MAX_ALPHA = 168  # reference examples/california_cigar_smoke_demo.py:58


def _rgba8(image, name):
    arr = np.ascontiguousarray(np.asarray(image), dtype=np.uint8)
    if arr.ndim != 3 or arr.shape[2] != 4:
        raise ValueError(f"{name} must be an (H, W, 4) uint8 RGBA image")
    return arr


def composite_desc(mode, base, layer, offset=(0, 0), base_alpha=0.0, layer_alpha=0.0, max_alpha=HYBRID_SMOKE_MAX_ALPHA):
    """f3d_composite_desc over two numpy images (the arrays must outlive the call that uses it)."""
    d = _CompositeDesc()
    d.struct_size, d.mode = C.sizeof(_CompositeDesc), int(mode)
    d.height, d.width = base.shape[:2]
    d.base = base.ctypes.data
    if layer is not None:
        d.layer_height, d.layer_width = layer.shape[:2]
        d.layer = layer.ctypes.data
    d.offset_x, d.offset_y = int(offset[0]), int(offset[1])
    d.base_alpha, d.layer_alpha, d.max_alpha = float(base_alpha), float(layer_alpha), int(max_alpha)
    return d


def _composite(mode, base, layer, **kw):
    base = _rgba8(base, "base")
    layer = None if layer is None else _rgba8(layer, "layer")
    desc = composite_desc(mode, base, layer, **kw)
    out = np.empty_like(base)
    err = C.create_string_buffer(512)
    seconds = C.c_double(0.0)
    rc = _native.lib().f3d_smoke_composite(C.byref(desc), out.ctypes.data, C.byref(seconds), err, len(err))
    if rc != 0:
        message = err.value.decode("utf-8", "replace")
        raise (ValueError if rc == _native.STATUS_VALUE else RuntimeError)(message)
    _composite.last_kernel_seconds = float(seconds.value)
    return out


def composite_atmospheric_smoke(base_rgba, smoke_rgba) -> np.ndarray:
    """A smoke layer (SmokeDomain.render_rgba / render_projection_rgba) as an optical veil over a terrain frame:
    transmittance, back-scatter and warm glow-through per pixel; the result is opaque.  Reference
    examples/california_cigar_smoke_demo.py:8527-8544 (which takes and returns PIL images of the same bytes)."""
    return _composite(COMPOSITE_ATMOSPHERIC, base_rgba, smoke_rgba)


def composite_main_smoke_maps(atmospheric_rgba, physical_rgba=None, *, atmospheric_alpha=0.42, physical_alpha=0.92,
                              max_alpha=HYBRID_SMOKE_MAX_ALPHA) -> np.ndarray:
    """The atmospheric blanket with the physical (solver) detail over it, alpha capped (reference :3367-3380)."""
    return _composite(COMPOSITE_SMOKE_MAPS, atmospheric_rgba, physical_rgba, base_alpha=atmospheric_alpha, layer_alpha=physical_alpha,
                      max_alpha=max_alpha)


def alpha_composite(base_rgba, layer_rgba, offset=(0, 0)) -> np.ndarray:
    """PIL.Image.alpha_composite(base, layer) with the layer's top-left corner at `offset` (clipped to the base), the
    operation composite_volume_detail (:8721-8725) and _shift_rgba (:3335-3349) of the reference example end in."""
    return _composite(COMPOSITE_OVER, base_rgba, layer_rgba, offset=offset)


def render_over_terrain(terrain_rgba, domain: "SmokeDomain", camera_pos, target, **kwargs) -> np.ndarray:
    """One frame of BASELINE.json configs[4]: `domain` ray-marched from the camera of the terrain frame and laid over it
    (terrain_rgba: the (H, W, 4) uint8 image of hybrid_render_terrain_reference for the same camera)."""
    terrain_rgba = _rgba8(terrain_rgba, "terrain_rgba")
    h, w = terrain_rgba.shape[:2]
    return composite_atmospheric_smoke(terrain_rgba, domain.render_rgba(w, h, camera_pos, target, **kwargs))


def simulate_over_terrain(terrain_rgba, domain: "SmokeDomain", settings, emitters, frames, camera_pos, target, *, steps_per_frame=1, rank=0,
                          world=1, **kwargs):
    """configs[4] end to end: `frames` frames of emitters -> solver -> ray-marcher -> composite over one terrain frame, the
    state resident on the GPU (SmokeSequence; resident=False: the round-3 host-array path, same bits).
    The solver is sequential in time, so every rank advances the same state (identical bits on every GPU) and renders
    only frames rank, rank + world, ...: returns {frame index: (H, W, 4) uint8} for this rank."""
    out = {}
    resident = kwargs.pop("resident", True)
    if not resident:  # the host-array path: every field crosses the bus for every step and frame
        for f in range(int(frames)):
            domain.step(settings, emitters, steps=steps_per_frame)
            if f % world == rank:
                out[f] = render_over_terrain(terrain_rgba, domain, camera_pos, target, **kwargs)
        return out
    seq = SmokeSequence(domain, terrain_rgba, camera_pos, target, up=kwargs.pop("up", (0.0, 1.0, 0.0)), fovy_deg=kwargs.pop("fovy_deg", 45.0),
                        sun_direction=kwargs.pop("sun_direction", (0.4, 0.8, -0.2)), render_settings=kwargs.pop("settings", None))
    for name in ("certificate", "cache"):
        kwargs.pop(name, None)
    if kwargs:
        raise TypeError(f"unexpected keyword arguments {sorted(kwargs)}")
    for f in range(int(frames)):
        seq.step(settings, emitters, steps=steps_per_frame)
        if f % world == rank:
            out[f] = seq.render_to_device().cpu().numpy()
    seq.download()
    return out


def render_sequence(frames: "Sequence[SmokeDomain]", width, height, camera_pos, target, *, rank=0, world=1, **kwargs):
    """Frames of an animation are independent: rank r renders frames r, r + world, ...; with an initialised
    torch.distributed process group the images are gathered on rank 0 (list in frame order; None elsewhere)."""
    mine = {i: frames[i].render_rgba(width, height, camera_pos, target, **kwargs) for i in range(rank, len(frames), world)}
    if world == 1:
        return [mine[i] for i in range(len(frames))]
    import torch
    import torch.distributed as dist

    gathered = [None] * world if rank == 0 else None
    dist.gather_object({i: torch.from_numpy(a) for i, a in mine.items()}, gathered, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in gathered:
        merged.update({i: t.numpy() for i, t in part.items()})
    return [merged[i] for i in range(len(frames))]


class SmokeSequence:
    """BASELINE.json configs[4] -- the 120-frame smoke sequence -- with everything resident on the GPU.

    `simulate_over_terrain` moves the nine solver fields to the device and back for every step, the six marcher fields
    again for every frame, and the terrain frame and the smoke layer for every composite: at 1080p on a 96 x 64 x 128
    domain that was 10.9 ms of wall time per frame around 2.2 ms of kernels (round 3).  Here the state is uploaded ONCE
    (torch tensors: device memory is what PyTorch is here for), `f3d_smoke_step` advances it in place,
    `f3d_smoke_render` marches it into a device image, `f3d_smoke_composite` lays that over the device-resident terrain
    frame, and only the finished RGBA8 frame leaves -- through one of two pinned buffers, so the copy of frame f overlaps
    the kernels of frame f + 1.  The arithmetic is the host-array path's, kernel for kernel: frames and state are
    bit-identical (tests/test_smoke.py).

        seq = SmokeSequence(domain, terrain_rgba, camera_pos=..., target=..., fovy_deg=...)
        for frame in seq.frames(120, settings, emitters):   # (H, W, 4) uint8 each
            ...
        seq.download()                                       # the domain's host arrays, brought up to date
    """

    _STATE = _STATE_FIELDS

    def __init__(self, domain: "SmokeDomain", terrain_rgba, camera_pos, target, up=(0.0, 1.0, 0.0), fovy_deg=45.0,
                 sun_direction=(0.4, 0.8, -0.2), render_settings=None, device=None):
        import torch

        self.torch = torch
        self.domain = domain
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        terrain = _rgba8(terrain_rgba, "terrain_rgba")
        self.height, self.width = terrain.shape[:2]
        self.settings = render_settings or SmokeRenderSettings()
        problem = self.settings.problem()
        if problem:
            raise RuntimeError(problem)
        self.view = _View(int(self.width), int(self.height), 0)
        self.view.camera_pos, self.view.target, self.view.up = ((C.c_float * 3)(*_f3(v)) for v in (camera_pos, target, up))
        self.view.fovy_deg = float(fovy_deg)
        self.view.sun_direction = (C.c_float * 3)(*_f3(sun_direction))
        self.state = {name: torch.from_numpy(np.ascontiguousarray(getattr(domain, name), np.float32)).to(self.device) for name in self._STATE}
        self.time_seconds, self.frame_index = float(domain.time_seconds), int(domain.frame_index)
        self.base = torch.from_numpy(terrain).to(self.device)
        self.layer = torch.empty((self.height, self.width, 4), dtype=torch.uint8, device=self.device)
        self.out = [torch.empty_like(self.layer) for _ in range(2)]
        self.pinned = [torch.empty((self.height, self.width, 4), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(self.device)  # the read-back of frame f beside the kernels of frame f + 1
        self.kernel_seconds = {"solver_step": 0.0, "march": 0.0, "composite": 0.0}
        # timing=True makes every library call record and wait for its device time (kernel_seconds); without it a call on
        # resident state returns with its launches enqueued and the host runs ahead of the device (round 5)
        self.timing = False
        self.rendered = [torch.cuda.Event() for _ in range(2)]
        # frames(): the solver and the marcher on a stream each, so that step f + 1 runs beside the march of frame f -- the
        # marcher reads the state only in its first kernels.  The library's sequence HANDLE (f3d_smoke_seq_*, ABI 6) holds the
        # two streams, orders a step behind the last render's reads and a render behind the last step, and owns the scratch.
        self.solver_stream = torch.cuda.Stream(self.device)  # (a high-priority stream measured no different: 0.80-0.81 ms a frame either way)
        self.render_stream = torch.cuda.Stream(self.device)
        self._err = C.create_string_buffer(512)
        # one handle per schedule: "overlap" = (solver stream, marcher stream); "serial" = both on the null stream, which is what
        # step() / render_to_device() called directly use (torch's default stream: the caller's tensor code is ordered with it)
        self._handles = {}
        self._mode = "serial"
        self._turn = 0

    def _handle(self, mode=None):
        mode = mode or self._mode
        self._last_mode = mode
        if mode not in self._handles:
            streams = (self.solver_stream.cuda_stream, self.render_stream.cuda_stream) if mode == "overlap" else (None, None)
            h = C.c_void_p(None)
            with self.torch.cuda.device(self.device):
                self._check(_native.lib().f3d_smoke_seq_create(C.c_void_p(streams[0]), C.c_void_p(streams[1]), C.byref(h), self._err, len(self._err)))
            self._handles[mode] = h
        return self._handles[mode]

    def _march_stream(self):
        return self.render_stream if self._mode == "overlap" else self.torch.cuda.default_stream(self.device)

    def close(self):
        """Give the sequence's device scratch back (the library keeps it per handle: about 0.5 GB at 1080p with self-shadowing)."""
        for h in self._handles.values():
            _native.lib().f3d_smoke_seq_destroy(h)
        self._handles = {}

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def stats(self) -> dict:
        """f3d_smoke_seq_stats of the handle the sequence used last (frames() with overlap: the two-stream one): bytes of scratch
        held, and the fill of the marcher's deferred self-shadow list in its last render (waits for it)."""
        class _Stats(C.Structure):
            _fields_ = [("scratch_bytes", C.c_uint64), ("shadow_list_chunks", C.c_uint32), ("shadow_list_chunks_used", C.c_uint32),
                        ("shadow_list_slots_per_chunk", C.c_uint32), ("reserved", C.c_uint32)]

        st = _Stats()
        self._check(_native.lib().f3d_smoke_seq_stats(self._handle(getattr(self, "_last_mode", None)), C.byref(st), self._err, len(self._err)))
        return {name: int(getattr(st, name)) for name, _ in _Stats._fields_ if name != "reserved"}

    def set_terrain(self, terrain_rgba):
        """Another terrain frame under the same smoke (a moving sun, a camera path rendered frame by frame)."""
        terrain = _rgba8(terrain_rgba, "terrain_rgba")
        if terrain.shape[:2] != (self.height, self.width):
            raise ValueError(f"terrain frame is {terrain.shape[1]}x{terrain.shape[0]}, the sequence renders {self.width}x{self.height}")
        self.base.copy_(self.torch.from_numpy(terrain), non_blocking=False)

    def _check(self, rc):
        if rc != 0:
            message = self._err.value.decode("utf-8", "replace")
            raise (ValueError if rc == _native.STATUS_VALUE else RuntimeError)(message)

    def step(self, settings: "SmokeStepSettings | None" = None, emitters=None, steps: int = 1):
        """SmokeDomain.step on the resident state (no transfer)."""
        settings = settings or SmokeStepSettings()
        emitters = list(emitters or [])
        if int(steps) < 1 or int(steps) > 1_000_000:
            raise ValueError(f"steps must be in 1..=1000000, got {steps}")
        d = self.domain
        st = _State()
        for name in self._STATE:
            setattr(st, name, self.state[name].data_ptr())
        st.dims = (C.c_uint32 * 3)(*d._dims)
        st.voxel_size = (C.c_float * 3)(*d._voxel)
        st.origin = (C.c_float * 3)(*d._origin)
        st.sparse_threshold, st.time_seconds, st.frame_index = float(d.sparse_threshold), float(self.time_seconds), int(self.frame_index) & 0xFFFFFFFF
        em = (_Emitter * max(1, len(emitters)))()
        for dst, e in zip(em, emitters):
            for name, _t in _Emitter._fields_:
                v = getattr(e, name)
                setattr(dst, name, (C.c_float * 3)(*v) if name in ("center", "velocity") else float(v))
        seconds = C.c_double(0.0)
        self._check(_native.lib().f3d_smoke_seq_step(self._handle(), C.byref(st), C.byref(settings._native()), em, C.c_uint32(len(emitters)), C.c_uint32(int(steps)),
                                                     C.byref(seconds) if self.timing else None, self._err, len(self._err)))
        self.time_seconds, self.frame_index = float(st.time_seconds), int(st.frame_index)
        self.kernel_seconds["solver_step"] = float(seconds.value) / int(steps)

    def render_to_device(self):
        """March the resident state and lay it over the terrain frame; returns the device image (a torch uint8 tensor that
        the call after next overwrites)."""
        d = self.domain
        vol = _Volume()
        s = self.state
        vol.density, vol.temperature, vol.soot, vol.humidity, vol.emission, vol.age = (s[n].data_ptr() for n in (
            "density", "temperature", "soot", "humidity", "emission_rate", "particle_age"))
        vol.dims = (C.c_uint32 * 3)(*d._dims)
        vol.voxel_size = (C.c_float * 3)(*d._voxel)
        vol.origin = (C.c_float * 3)(*d._origin)
        vol.frame_index = int(self.frame_index) & 0xFFFFFFFF
        seconds = C.c_double(0.0)
        native = self.settings._native()
        # (the handle orders this march behind the last step, and the next step behind this march's re-pack of the state)
        self._check(_native.lib().f3d_smoke_seq_render(self._handle(), C.byref(vol), C.byref(self.view), C.byref(native), C.c_void_p(self.layer.data_ptr()),
                                                       C.byref(seconds) if self.timing else None, self._err, len(self._err)))
        self.kernel_seconds["march"] = float(seconds.value)
        out = self.out[self._turn]
        # (this image's last read-back -- two frames ago, on the copy stream -- before the composite overwrites it)
        stream = self._march_stream()
        stream.wait_event(self.copied[self._turn])
        desc = _CompositeDesc()
        desc.struct_size = C.sizeof(_CompositeDesc)
        desc.mode, desc.width, desc.height = COMPOSITE_ATMOSPHERIC, self.width, self.height
        desc.layer_width, desc.layer_height = self.width, self.height
        desc.base, desc.layer = self.base.data_ptr(), self.layer.data_ptr()
        desc.max_alpha = HYBRID_SMOKE_MAX_ALPHA
        seconds = C.c_double(0.0)
        self._check(_native.lib().f3d_smoke_seq_composite(self._handle(), C.byref(desc), C.c_void_p(out.data_ptr()), C.byref(seconds) if self.timing else None,
                                                          self._err, len(self._err)))
        self.kernel_seconds["composite"] = float(seconds.value)
        self.rendered[self._turn].record(stream)
        return out

    def frames(self, count: int, settings=None, emitters=None, steps_per_frame: int = 1, timing: bool = False, overlap: bool = True,
               base_provider=None):
        """`count` frames of emitters -> solver -> ray-marcher -> composite; yields (H, W, 4) uint8 host images (each a view of
        a pinned buffer that the frame after next reuses: copy what is to be kept).  The device-to-host copy of a frame
        runs while the next frame's kernels do, and -- unless timing=True asks for kernel_seconds -- the host enqueues a
        frame's launches without waiting for the device.  overlap: the solver on a stream of its own, one step ahead of the
        marcher (neither fills the chip: the solver's phases are 3 000 short workgroups, the marcher's first walk is as long
        as its longest ray); False = everything on the null stream, one kernel after the other.
        base_provider(i, base, stream): called before frame i's march; it enqueues, ON `stream` (the torch stream the frame's
        composite runs on), whatever writes this frame's terrain image into `base` ((H, W, 4) uint8 on the device) -- a terrain
        render under a moving sun, resolved straight into it (TerrainSession(stream=stream.cuda_stream).resolve_device)."""
        torch = self.torch
        pending = None
        self.timing = bool(timing)
        overlap = bool(overlap) and not self.timing
        torch.cuda.synchronize(self.device)  # (the state's upload and whatever the caller did to it on other streams)
        self._mode = "overlap" if overlap else "serial"
        try:
            for index in range(int(count)):
                self.step(settings, emitters, steps=steps_per_frame)  # (on the solver's stream, behind the last march's reads of the state)
                if base_provider is not None:  # (in stream order behind the previous frame's composite, which read the old image)
                    base_provider(index, self.base, self._march_stream())
                image = self.render_to_device()  # (on the marcher's stream, behind the step)
                turn = self._turn
                with torch.cuda.stream(self.copy_stream):
                    self.copy_stream.wait_event(self.rendered[turn])  # the image is complete when the marcher's stream gets there
                    self.pinned[turn].copy_(image, non_blocking=True)
                    self.copied[turn].record()
                self._turn ^= 1
                if pending is not None:
                    self.copied[pending].synchronize()
                    yield self.pinned[pending].numpy()
                pending = turn
            if pending is not None:
                self.copied[pending].synchronize()
                yield self.pinned[pending].numpy()
        finally:
            torch.cuda.synchronize(self.device)
            self._mode = "serial"

    def download(self) -> "SmokeDomain":
        """Bring the domain's host arrays (and its clock) up to date with the resident state."""
        for name in self._STATE:
            setattr(self.domain, name, self.state[name].cpu().numpy())
        self.domain.time_seconds, self.domain.frame_index = self.time_seconds, self.frame_index
        return self.domain
