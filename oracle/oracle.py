"""ctypes front-end of the CPU oracle (libf3d_oracle.so).

TEST INFRASTRUCTURE ONLY (see f3d_oracle.h): imported by tests/, by
``__graft_entry__.smoke()`` and by ``bench.py``'s cpu_baseline leg.  The product
package ``forge3d_amd`` never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libf3d_oracle.so"

EARTH_MODELS = {"flat": 0, "sphere": 1, "ellipsoid": 2, "wgs84": 2}
REFRACTION_MODELS = {"none": 0, "bennett": 1, "saemundsson": 2, "effective_radius": 3}


class Reservoir(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("light_index", C.c_uint32),
        ("direction", C.c_float * 3),
        ("intensity", C.c_float),
        ("light_type", C.c_uint32),
        ("params", C.c_float * 3),
        ("pad", C.c_uint32 * 4),
        ("w_sum", C.c_float),
        ("m", C.c_uint32),
        ("weight", C.c_float),
        ("target_pdf", C.c_float),
    ]


RESERVOIR_DTYPE = np.dtype(
    [
        ("position", "<f4", (3,)),
        ("light_index", "<u4"),
        ("direction", "<f4", (3,)),
        ("intensity", "<f4"),
        ("light_type", "<u4"),
        ("params", "<f4", (3,)),
        ("pad", "<u4", (4,)),
        ("w_sum", "<f4"),
        ("m", "<u4"),
        ("weight", "<f4"),
        ("target_pdf", "<f4"),
    ]
)
assert RESERVOIR_DTYPE.itemsize == 80 and C.sizeof(Reservoir) == 80


class Desc(C.Structure):
    _fields_ = [
        ("heights", C.c_void_p),
        ("dem_w", C.c_uint32),
        ("dem_h", C.c_uint32),
        ("spacing_x", C.c_float),
        ("spacing_z", C.c_float),
        ("exaggeration", C.c_float),
        ("albedo", C.c_float * 3),
        ("cam_origin", C.c_float * 3),
        ("cam_look_at", C.c_float * 3),
        ("cam_up", C.c_float * 3),
        ("fov_y_deg", C.c_float),
        ("exposure", C.c_float),
        ("sun_azimuth_deg", C.c_float),
        ("sun_elevation_deg", C.c_float),
        ("sun_intensity", C.c_float),
        ("sun_color", C.c_float * 3),
        ("observer_lat_deg", C.c_double),
        ("observer_lon_deg", C.c_double),
        ("earth_model", C.c_int32),
        ("refraction_model", C.c_int32),
        ("sphere_radius_m", C.c_double),
        ("refraction_k", C.c_double),
        ("pressure_mbar", C.c_double),
        ("temperature_c", C.c_double),
        ("env_map", C.c_void_p),
        ("env_w", C.c_uint32),
        ("env_h", C.c_uint32),
        ("env_intensity", C.c_float),
        ("mesh_vertices", C.c_void_p),
        ("mesh_vertex_count", C.c_uint32),
        ("mesh_indices", C.c_void_p),
        ("mesh_index_count", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("seed", C.c_uint32),
        ("spp", C.c_uint32),
        ("max_frames", C.c_uint32),
        ("min_frames", C.c_uint32),
        ("variance_threshold", C.c_float),
        ("atmosphere", C.c_void_p),
        ("accum_override", C.c_void_p),
    ]


class Aether(C.Structure):
    """f3do_aether: the three tables the post reads, decoded to f32 RGBA"""
    _fields_ = [("transmittance", C.c_void_p), ("scattering", C.c_void_p), ("aerial", C.c_void_p),
                ("dims", C.c_uint32 * 9), ("turbidity", C.c_float), ("ozone_du", C.c_float), ("mie_g", C.c_float),
                ("bottom_radius_m", C.c_float), ("top_radius_m", C.c_float), ("rayleigh_scale_height_m", C.c_float),
                ("mie_scale_height_m", C.c_float), ("max_aerial_distance_m", C.c_float), ("ground_albedo", C.c_float),
                ("scattering_orders", C.c_uint32)]


def aether_struct(handle):
    """(Aether, keep-alives) from an AtmosphereLutHandle-like object (uint16 RGBA16F tables + config)."""
    a = Aether()
    keep = [np.ascontiguousarray(np.asarray(t, np.uint16).view(np.float16).astype(np.float32))
            for t in (handle.transmittance, handle.accumulated_scattering, handle.aerial_perspective)]
    a.transmittance, a.scattering, a.aerial = (k.ctypes.data for k in keep)
    d, c = handle.config.dimensions, handle.config
    a.dims = (C.c_uint32 * 9)(d.transmittance_mu, d.transmittance_height, d.scattering_mu_view, d.scattering_mu_sun,
                              d.scattering_height, d.scattering_nu, d.aerial_distance, d.aerial_mu_view, d.aerial_height)
    for name in ("turbidity", "ozone_du", "mie_g", "bottom_radius_m", "top_radius_m", "rayleigh_scale_height_m",
                 "mie_scale_height_m", "max_aerial_distance_m", "ground_albedo"):
        setattr(a, name, float(getattr(c, name)))
    a.scattering_orders = int(c.scattering_orders)
    return a, keep


def aether_segment_transmittance(distance_m, altitude_m, mu, turbidity, ozone_du=300.0, bottom_radius_m=6_360_000.0):
    """ae_segment_transmittance of the post (linear sRGB, white-normalised)."""
    out = (C.c_float * 3)()
    lib().f3do_aether_segment_transmittance(distance_m, altitude_m, mu, bottom_radius_m, turbidity, ozone_du, out)
    return np.array(out[:], np.float32)


def aether_sky(handle, altitude_m, view, sun):
    """Sky radiance of the post for unit sun intensity (accumulated-scattering LUT tap), before exposure / Reinhard."""
    a, keep = aether_struct(handle)
    out = (C.c_float * 3)()
    lib().f3do_aether_sky(C.byref(a), float(altitude_m), (C.c_float * 3)(*map(float, view)), (C.c_float * 3)(*map(float, sun)), out)
    return np.array(out[:], np.float32)


def aether_aerial(handle, surface, altitude_m, depth_m, view, sun, sun_intensity=1.0):
    """The post's terrain-hit transport for one ray: surface * T(segment) + inscatter * sun intensity, before exposure / Reinhard."""
    a, keep = aether_struct(handle)
    out = (C.c_float * 3)()
    lib().f3do_aether_aerial(C.byref(a), (C.c_float * 3)(*map(float, surface)), C.c_float(float(altitude_m)), C.c_float(float(depth_m)),
                             (C.c_float * 3)(*map(float, view)), (C.c_float * 3)(*map(float, sun)), C.c_float(float(sun_intensity)), out)
    return np.array(out[:], np.float32)


class Out(C.Structure):
    _fields_ = [
        ("rgba", C.c_void_p),
        ("albedo", C.c_void_p),
        ("normal", C.c_void_p),
        ("depth", C.c_void_p),
        ("accum", C.c_void_p),
        ("welford", C.c_void_p),
        ("reservoir_prev", C.c_void_p),
        ("frames", C.c_uint32),
        ("variance", C.c_float),
        ("converged", C.c_int32),
        ("minmax_pyramid_bytes", C.c_uint64),
        ("n_node", C.c_uint64),
        ("n_leaf", C.c_uint64),
        ("n_hit", C.c_uint64),
        ("n_samples", C.c_uint64),
        ("n_rays", C.c_uint64),
        ("loop_seconds", C.c_double),
    ]


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (a few seconds)."""
    src = _HERE / "f3d_oracle.c"
    hdr = _HERE / "f3d_oracle.h"
    if (
        force
        or not _LIB_PATH.exists()
        or _LIB_PATH.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime)
    ):
        subprocess.run(["make", "-C", str(_HERE), "-B", "libf3d_oracle.so"], check=True,
                       capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        L = C.CDLL(str(_LIB_PATH))
        L.f3do_render.argtypes = [C.POINTER(Desc), C.POINTER(Out), C.c_char_p, C.c_size_t]
        L.f3do_render.restype = C.c_int
        L.f3do_build_minmax_mips.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                             C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
        L.f3do_build_minmax_mips.restype = C.c_int
        L.f3do_terrain_trace_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float,
                                               C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                               C.c_uint32, C.c_void_p, C.c_uint32, C.c_int32,
                                               C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
        L.f3do_terrain_trace_batch.restype = C.c_int
        L.f3do_effective_radius_m.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_int32,
                                              C.c_double, C.c_double, C.c_double, C.c_double,
                                              C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
        L.f3do_effective_radius_m.restype = C.c_int
        L.f3do_aether_segment_transmittance.argtypes = [C.c_float] * 6 + [C.POINTER(C.c_float)]
        L.f3do_aether_segment_transmittance.restype = None
        L.f3do_aether_sky.argtypes = [C.POINTER(Aether), C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.f3do_aether_sky.restype = None
        L.f3do_f16_round.argtypes = [C.c_float]
        L.f3do_f16_round.restype = C.c_float
        L.f3do_sincos_2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.f3do_sincos_2pi.restype = None
        L.f3do_num_threads.restype = C.c_int
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


def _f3(v):
    return (C.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))


def render(heightmap, width, height, camera=None, *, spacing=(1.0, 1.0), exaggeration=1.0,
           albedo=(0.6, 0.6, 0.6), sun_azimuth_deg=315.0, sun_elevation_deg=45.0,
           sun_intensity=2.5, sun_color=(1.0, 0.97, 0.92), env_map=None, env_intensity=0.35,
           mesh_vertices=None, mesh_indices=None, spp=1, max_frames=512, min_frames=32,
           variance_threshold=1e-3, seed=7, observer_latitude_deg=0.0,
           observer_longitude_deg=0.0, earth_model="ellipsoid", sphere_radius_m=6_371_008.8,
           refraction_model="bennett", refraction_k=0.13, pressure_mbar=1013.25,
           temperature_c=15.0, dump_state=False, atmosphere=None, accum_override=None):
    """Run the CPU oracle; returns the reference's result dict (+ counters).  atmosphere: an
    AtmosphereLutHandle-like object (the AETHER aerial-perspective post) or None."""
    L = lib()
    dem = np.ascontiguousarray(heightmap, dtype=np.float32)
    cam = dict(camera or {})
    d = Desc()
    d.heights = dem.ctypes.data
    d.dem_h, d.dem_w = dem.shape
    d.spacing_x, d.spacing_z = float(spacing[0]), float(spacing[1])
    d.exaggeration = float(exaggeration)
    d.albedo = _f3(albedo)
    d.cam_origin = _f3(cam.get("origin", (0.0, 50.0, 120.0)))
    d.cam_look_at = _f3(cam.get("look_at", (0.0, 0.0, 0.0)))
    d.cam_up = _f3(cam.get("up", (0.0, 1.0, 0.0)))
    d.fov_y_deg = float(cam.get("fov_y", 45.0))
    d.exposure = float(cam.get("exposure", 1.0))
    d.sun_azimuth_deg = float(sun_azimuth_deg)
    d.sun_elevation_deg = float(sun_elevation_deg)
    d.sun_intensity = float(sun_intensity)
    d.sun_color = _f3(sun_color)
    d.observer_lat_deg = float(observer_latitude_deg)
    d.observer_lon_deg = float(observer_longitude_deg)
    d.earth_model = EARTH_MODELS[earth_model]
    d.refraction_model = REFRACTION_MODELS[refraction_model]
    d.sphere_radius_m = float(sphere_radius_m)
    d.refraction_k = float(refraction_k)
    d.pressure_mbar = float(pressure_mbar)
    d.temperature_c = float(temperature_c)
    keep = [dem]
    if env_map is not None:
        env = np.ascontiguousarray(env_map, dtype=np.float32)
        keep.append(env)
        d.env_map = env.ctypes.data
        d.env_h, d.env_w = env.shape[0], env.shape[1]
    d.env_intensity = float(env_intensity)
    if mesh_vertices is not None:
        mv = np.ascontiguousarray(mesh_vertices, dtype=np.float32)
        mi = np.ascontiguousarray(mesh_indices, dtype=np.uint32)
        keep += [mv, mi]
        d.mesh_vertices = mv.ctypes.data
        d.mesh_vertex_count = mv.shape[0]
        d.mesh_indices = mi.ctypes.data
        d.mesh_index_count = mi.size
    d.width, d.height = int(width), int(height)
    d.seed, d.spp = int(seed) & 0xFFFFFFFF, int(spp)
    d.max_frames, d.min_frames = int(max_frames), int(min_frames)
    d.variance_threshold = float(variance_threshold)
    if atmosphere is not None:
        aether, aether_keep = aether_struct(atmosphere)
        keep += aether_keep + [aether]
        d.atmosphere = C.addressof(aether)
    if accum_override is not None:  # (H, W, 4) radiance sums + frame count replacing the render's own accumulation
        override = np.ascontiguousarray(accum_override, np.float32).reshape(int(height) * int(width), 4)
        keep.append(override)
        d.accum_override = override.ctypes.data

    P = int(width) * int(height)
    rgba = np.zeros((height, width, 4), np.uint8)
    alb = np.zeros((height, width, 3), np.float32)
    nrm = np.zeros((height, width, 3), np.float32)
    dep = np.zeros((height, width), np.float32)
    o = Out()
    o.rgba, o.albedo, o.normal, o.depth = (rgba.ctypes.data, alb.ctypes.data, nrm.ctypes.data,
                                           dep.ctypes.data)
    state = {}
    if dump_state:
        state["accum"] = np.zeros((P, 4), np.float32)
        state["welford"] = np.zeros((P, 2), np.float32)
        state["reservoir_prev"] = np.zeros(P, RESERVOIR_DTYPE)
        o.accum = state["accum"].ctypes.data
        o.welford = state["welford"].ctypes.data
        o.reservoir_prev = state["reservoir_prev"].ctypes.data
    err = C.create_string_buffer(1024)
    rc = L.f3do_render(C.byref(d), C.byref(o), err, len(err))
    if rc != 0:
        raise OracleError(rc, err.value.decode("utf-8", "replace"))
    out = {
        "rgba": rgba, "albedo": alb, "normal": nrm, "depth": dep,
        "frames": int(o.frames), "variance": float(o.variance), "converged": bool(o.converged),
        "minmax_pyramid_bytes": int(o.minmax_pyramid_bytes),
        "n_node": int(o.n_node), "n_leaf": int(o.n_leaf), "n_hit": int(o.n_hit),
        "n_samples": int(o.n_samples), "n_rays": int(o.n_rays),
        "loop_seconds": float(o.loop_seconds),
    }
    out.update(state)
    return out


def build_minmax_mips(heights):
    """Returns (levels, dims): levels[l] is a (ph, pw, 2) float32 array, finest first."""
    L = lib()
    dem = np.ascontiguousarray(heights, dtype=np.float32)
    h, w = dem.shape
    tot = C.c_uint64(0)
    dims = np.zeros(32, np.uint32)
    n = L.f3do_build_minmax_mips(dem.ctypes.data, w, h, None, dims.ctypes.data, 16, C.byref(tot))
    if n < 0:
        raise OracleError(n, {-10: "terrain heightfield must be at least 2x2 texels",
                              -11: "terrain heightfield contains non-finite samples"}.get(n, "error"))
    flat = np.zeros(tot.value, np.float32)
    L.f3do_build_minmax_mips(dem.ctypes.data, w, h, flat.ctypes.data, dims.ctypes.data, 16,
                             C.byref(tot))
    levels, off = [], 0
    out_dims = []
    for l in range(n):
        pw, ph = int(dims[2 * l]), int(dims[2 * l + 1])
        levels.append(flat[off:off + pw * ph * 2].reshape(ph, pw, 2))
        out_dims.append((pw, ph))
        off += pw * ph * 2
    return levels, out_dims


def terrain_trace_batch(heights, rays, *, origin=(0.0, 0.0), spacing=(1.0, 1.0), exaggeration=1.0,
                        inv_two_r_prime=0.0, curvature_enabled=False, any_hit=True,
                        apply_curvature=True):
    """terrain_trace over rays (n,8): origin xyz, tmin, direction xyz, tmax."""
    L = lib()
    dem = np.ascontiguousarray(heights, dtype=np.float32)
    r = np.ascontiguousarray(rays, dtype=np.float32)
    n = r.shape[0]
    hit = np.zeros(n, np.uint32)
    t = np.zeros(n, np.float32)
    nrm = np.zeros((n, 3), np.float32)
    cnt = np.zeros(3, np.uint64)
    rc = L.f3do_terrain_trace_batch(dem.ctypes.data, dem.shape[1], dem.shape[0], float(origin[0]),
                                    float(origin[1]), float(spacing[0]), float(spacing[1]),
                                    float(exaggeration), float(inv_two_r_prime),
                                    1 if curvature_enabled else 0, r.ctypes.data, n,
                                    1 if any_hit else 0, 1 if apply_curvature else 0,
                                    hit.ctypes.data, t.ctypes.data, nrm.ctypes.data, cnt.ctypes.data)
    if rc != 0:
        raise OracleError(rc, "terrain_trace_batch failed")
    return {"hit": hit, "t": t, "normal": nrm, "n_node": int(cnt[0]), "n_leaf": int(cnt[1]),
            "n_hit": int(cnt[2])}


def effective_radius_m(earth_model, refraction_model, azimuth_deg, *, latitude_deg=0.0,
                       sphere_radius_m=6_371_008.8, pressure_mbar=1013.25, temperature_c=15.0,
                       k=0.13):
    L = lib()
    out = C.c_double(0.0)
    err = C.create_string_buffer(256)
    rc = L.f3do_effective_radius_m(EARTH_MODELS[earth_model], latitude_deg, sphere_radius_m,
                                   REFRACTION_MODELS[refraction_model], pressure_mbar,
                                   temperature_c, k, azimuth_deg, C.byref(out), err, len(err))
    if rc:
        raise OracleError(rc, err.value.decode())
    return out.value


def num_threads() -> int:
    return int(lib().f3do_num_threads())


def set_num_threads(n: int) -> None:
    lib().f3do_set_num_threads(C.c_int(int(n)))
