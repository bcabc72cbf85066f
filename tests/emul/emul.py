"""ctypes front-end of tests/emul/libf3d_emul.so -- TEST INFRASTRUCTURE ONLY.

The emulator compiles the product's kernel headers (forge3d_amd/csrc/f3d_{trace,shade,
build,setup}.h) for the host and runs them one lane at a time, so the kernel LOGIC can be
compared bit-for-bit with the oracle in the GPU-less container.  The `-m gpu` tests repeat
the comparison on the real device through libf3dhip.so.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from forge3d_amd import _native

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "libf3d_emul.so"
_CSRC = _HERE.parent.parent / "forge3d_amd" / "csrc"


def build(force=False):
    global _LIB
    srcs = [_HERE / "f3d_emul.cpp"] + sorted(_CSRC.glob("*.h"))
    extra = os.environ.get("F3D_EMUL_CXXFLAGS", "").split()  # experiment switches (-DF3D_...)
    if extra:  # a library of its own per set of switches: a build with other switches is never taken for this one
        import hashlib
        _LIB = _HERE / ("libf3d_emul_%s.so" % hashlib.sha256(" ".join(extra).encode()).hexdigest()[:10])
    if force or not _LIB.exists() or _LIB.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
        tmp = _LIB.with_suffix(f".{os.getpid()}.tmp")  # parallel test workers may all find the library stale: build aside, then rename
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-march=x86-64-v3",
                        "-ffp-contract=off", "-DF3D_HORIZON_LAZY", *extra, str(_HERE / "f3d_emul.cpp"), "-o", str(tmp)],
                       check=True, capture_output=True)
        os.replace(tmp, _LIB)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB))
        _lib.emul_render.restype = C.c_int
        _lib.emul_trace_batch.restype = C.c_int
        _lib.emul_build_minmax_mips.restype = C.c_int
    return _lib


def render(heightmap, width, height, camera=None, *, spacing=(1.0, 1.0), exaggeration=1.0,
           albedo=(0.6, 0.6, 0.6), sun_azimuth_deg=315.0, sun_elevation_deg=45.0, sun_intensity=2.5,
           sun_color=(1.0, 0.97, 0.92), env_map=None, env_intensity=0.35, mesh_vertices=None,
           mesh_indices=None, spp=1, max_frames=512, min_frames=32, variance_threshold=1e-3, seed=7,
           observer_latitude_deg=0.0, observer_longitude_deg=0.0, earth_model="ellipsoid",
           sphere_radius_m=6_371_008.8, refraction_model="bennett", refraction_k=0.13,
           pressure_mbar=1013.25, temperature_c=15.0, rows=None, sample_lanes=1, frames_in_flight=0):
    """rows=(begin, end): render only that strip of the full image (no halo exchange; statistics runs).
    sample_lanes: 1 = frame_pixel; 2, 4, 8 = CPU mirror of the frame kernel's sample-lane form; the
    result then carries "retraces" = primary rays traced again after a wrong hit-flag prediction.
    frames_in_flight > 1: batches of frames traced first, then the ordered half per frame (the per-pixel
    code of k_trace / k_merge / k_fix); "retraced_pixels" = pixel-frames whose sun-direction prediction failed."""
    d, keep = _native.make_desc(heightmap, width, height, dict(camera or {}), spacing, exaggeration, albedo,
                                sun_azimuth_deg, sun_elevation_deg, sun_intensity, env_map, env_intensity,
                                mesh_vertices, mesh_indices, spp, max_frames, min_frames, variance_threshold,
                                seed, sun_color, observer_latitude_deg, observer_longitude_deg, earth_model,
                                sphere_radius_m, refraction_model, refraction_k, pressure_mbar, temperature_c)
    row_begin, row_end = rows if rows else (0, 0)
    height = (row_end - row_begin) if rows else height
    P = width * height
    rgba = np.zeros((height, width, 4), np.uint8)
    alb = np.zeros((height, width, 3), np.float32)
    nrm = np.zeros((height, width, 3), np.float32)
    dep = np.zeros((height, width), np.float32)
    accum = np.zeros((P, 4), np.float32)
    m2 = np.zeros(P, np.float32)
    res = np.zeros((P, 4), np.uint32)
    frames, conv, var = C.c_uint32(0), C.c_int32(0), C.c_float(0)
    err = C.create_string_buffer(512)
    lib().emul_set_sample_lanes(C.c_uint32(int(sample_lanes)))
    lib().emul_take_retraces.restype = C.c_uint64
    lib().emul_take_retraces()
    lib().emul_take_retraced.restype = C.c_uint64
    lib().emul_take_retraced()
    lib().emul_set_frames_in_flight(C.c_uint32(int(frames_in_flight)))
    rc = lib().emul_render(C.byref(d), C.c_uint32(row_begin), C.c_uint32(row_end), C.c_void_p(rgba.ctypes.data),
                           C.c_void_p(alb.ctypes.data), C.c_void_p(nrm.ctypes.data), C.c_void_p(dep.ctypes.data),
                           C.byref(frames), C.byref(var), C.byref(conv), C.c_void_p(accum.ctypes.data),
                           C.c_void_p(m2.ctypes.data), C.c_void_p(res.ctypes.data), err, C.c_size_t(len(err)))
    retraces = int(lib().emul_take_retraces())
    retraced = int(lib().emul_take_retraced())
    lib().emul_set_frames_in_flight(C.c_uint32(0))
    lib().emul_set_sample_lanes(C.c_uint32(1))
    if rc != 0:
        raise RuntimeError(f"emul status {rc}: {err.value.decode()}")
    return {"retraces": retraces, "retraced_pixels": retraced, "rgba": rgba, "albedo": alb, "normal": nrm, "depth": dep, "frames": frames.value,
            "variance": var.value, "converged": bool(conv.value), "accum": accum, "m2": m2, "res": res}


def terrain_trace_batch(heights, rays, *, origin=(0.0, 0.0), spacing=(1.0, 1.0), exaggeration=1.0,
                        inv_two_r_prime=0.0, curvature_enabled=False, any_hit=True, apply_curvature=True):
    dem = np.ascontiguousarray(heights, np.float32)
    r = np.ascontiguousarray(rays, np.float32)
    n = r.shape[0]
    hit = np.zeros(n, np.uint32)
    t = np.zeros(n, np.float32)
    nrm = np.zeros((n, 3), np.float32)
    rc = lib().emul_trace_batch(C.c_void_p(dem.ctypes.data), C.c_uint32(dem.shape[1]), C.c_uint32(dem.shape[0]),
                                C.c_float(origin[0]), C.c_float(origin[1]), C.c_float(spacing[0]),
                                C.c_float(spacing[1]), C.c_float(exaggeration), C.c_float(inv_two_r_prime),
                                C.c_uint32(1 if curvature_enabled else 0), C.c_void_p(r.ctypes.data), C.c_uint32(n),
                                C.c_int32(int(any_hit)), C.c_int32(1 if apply_curvature else 0),
                                C.c_void_p(hit.ctypes.data), C.c_void_p(t.ctypes.data), C.c_void_p(nrm.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"emul status {rc}")
    return {"hit": hit, "t": t, "normal": nrm}


def terrain_trace_batch_wave(heights, rays, *, origin=(0.0, 0.0), spacing=(1.0, 1.0), exaggeration=1.0, inv_two_r_prime=0.0,
                             curvature_enabled=False, any_hit=2, apply_curvature=True, share_below=0):
    """The stackless march over the batch in 64-lane WAVES with the ray sharing live (lanes are fibers, votes and
    shuffles are exchanged in lockstep): any_hit = 2 any / 3 closest, +4 start in the origin's cell; share_below =
    sharing threshold (0 = default 16, 64 = every any-hit ray without curvature policy is dealt at once)."""
    dem = np.ascontiguousarray(heights, np.float32)
    r = np.ascontiguousarray(rays, np.float32)
    n = r.shape[0]
    hit, t, nrm = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    stats = np.zeros(2, np.uint64)
    rc = lib().emul_trace_batch_wave(C.c_void_p(dem.ctypes.data), C.c_uint32(dem.shape[1]), C.c_uint32(dem.shape[0]),
                                     C.c_float(origin[0]), C.c_float(origin[1]), C.c_float(spacing[0]), C.c_float(spacing[1]),
                                     C.c_float(exaggeration), C.c_float(inv_two_r_prime), C.c_uint32(1 if curvature_enabled else 0),
                                     C.c_void_p(r.ctypes.data), C.c_uint32(n), C.c_int32(int(any_hit)),
                                     C.c_int32(1 if apply_curvature else 0), C.c_uint32(int(share_below)),
                                     C.c_void_p(hit.ctypes.data), C.c_void_p(t.ctypes.data), C.c_void_p(nrm.ctypes.data),
                                     C.c_void_p(stats.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"emul status {rc}")
    return {"hit": hit, "t": t, "normal": nrm, "exchanges": int(stats[0]), "deals": int(stats[1])}


def primary_start(heightmap, width, height, camera, pixels, **kw):
    """(t_clear, level) of the primary-ray certificate (csrc/f3d_cone.h) for each (gx, gy) of `pixels`."""
    defaults = dict(spacing=(1.0, 1.0), exaggeration=1.0, albedo=(0.6, 0.6, 0.6), sun_azimuth_deg=315.0,
                    sun_elevation_deg=45.0, sun_intensity=2.5, env_map=None, env_intensity=0.35,
                    mesh_vertices=None, mesh_indices=None, spp=1, max_frames=512, min_frames=32,
                    variance_threshold=1e-3, seed=7, sun_color=(1.0, 0.97, 0.92), observer_latitude_deg=0.0,
                    observer_longitude_deg=0.0, earth_model="ellipsoid", sphere_radius_m=6_371_008.8,
                    refraction_model="bennett", refraction_k=0.13, pressure_mbar=1013.25, temperature_c=15.0)
    defaults.update(kw)
    d, keep = _native.make_desc(heightmap, width, height, dict(camera or {}), **defaults)
    out = []
    for gx, gy in pixels:
        t, level = C.c_float(0), C.c_uint32(0)
        if lib().emul_primary_start(C.byref(d), C.c_uint32(int(gx)), C.c_uint32(int(gy)), C.byref(t), C.byref(level)) != 0:
            raise RuntimeError("emul_primary_start failed")
        out.append((float(t.value), int(level.value)))
    del keep
    return out


def sun_clear(heightmap, width, height, camera, pixels, **kw):
    """Per (gx, gy): dict(clear_from, depth, origin, wi) of the sun-ray certificate (csrc/f3d_cone.h sun_clear_from);
    clear_from > 1e37 = no certificate (or the centre ray misses)."""
    defaults = dict(spacing=(1.0, 1.0), exaggeration=1.0, albedo=(0.6, 0.6, 0.6), sun_azimuth_deg=315.0,
                    sun_elevation_deg=45.0, sun_intensity=2.5, env_map=None, env_intensity=0.35,
                    mesh_vertices=None, mesh_indices=None, spp=1, max_frames=512, min_frames=32,
                    variance_threshold=1e-3, seed=7, sun_color=(1.0, 0.97, 0.92), observer_latitude_deg=0.0,
                    observer_longitude_deg=0.0, earth_model="ellipsoid", sphere_radius_m=6_371_008.8,
                    refraction_model="bennett", refraction_k=0.13, pressure_mbar=1013.25, temperature_c=15.0)
    defaults.update(kw)
    d, keep = _native.make_desc(heightmap, width, height, dict(camera or {}), **defaults)
    out = []
    for gx, gy in pixels:
        rec = (C.c_float * 8)()
        if lib().emul_sun_clear(C.byref(d), C.c_uint32(int(gx)), C.c_uint32(int(gy)), rec) != 0:
            raise RuntimeError("emul_sun_clear failed")
        out.append({"clear_from": float(rec[0]), "depth": float(rec[1]), "origin": tuple(float(v) for v in rec[2:5]),
                    "wi": tuple(float(v) for v in rec[5:8])})
    del keep
    return out


def horizon_blocks(heightmap, cells, **kw):
    """Per DEM cell (cx, cz): the far-horizon record of the block that holds it (csrc/f3d_cone.h horizon_block_build):
    dict(far[8], rho, stop_distance, level, centre (x, z), y_lo, size (x, z))."""
    defaults = dict(spacing=(1.0, 1.0), exaggeration=1.0, albedo=(0.6, 0.6, 0.6), sun_azimuth_deg=315.0,
                    sun_elevation_deg=45.0, sun_intensity=2.5, env_map=None, env_intensity=0.35,
                    mesh_vertices=None, mesh_indices=None, spp=1, max_frames=512, min_frames=32,
                    variance_threshold=1e-3, seed=7, sun_color=(1.0, 0.97, 0.92), observer_latitude_deg=0.0,
                    observer_longitude_deg=0.0, earth_model="ellipsoid", sphere_radius_m=6_371_008.8,
                    refraction_model="bennett", refraction_k=0.13, pressure_mbar=1013.25, temperature_c=15.0)
    defaults.update(kw)
    d, keep = _native.make_desc(heightmap, 8, 8, {}, **defaults)
    cells = np.ascontiguousarray(cells, np.uint32).reshape(-1, 2)
    rec = np.zeros((cells.shape[0], 16), np.float32)
    if lib().emul_horizon_blocks(C.byref(d), C.c_uint32(cells.shape[0]), cells.ctypes.data_as(C.c_void_p), rec.ctypes.data_as(C.c_void_p)) != 0:
        raise RuntimeError("emul_horizon_blocks failed")
    del keep
    return [{"far": [float(v) for v in r[0:8]], "rho": float(r[8]), "stop_distance": float(r[9]), "level": int(r[10]),
             "centre": (float(r[11]), float(r[12])), "y_lo": float(r[13]), "size": (float(r[14]), float(r[15]))} for r in rec]


def bvh_fingerprint(vertices, indices, threaded: bool):
    """(FNV-1a of the node + triangle arrays, node count) of the mesh BVH, built on one thread or with
    the worker threads of the large-mesh path (forced on for any size)."""
    v = np.ascontiguousarray(vertices, np.float32)
    i = np.ascontiguousarray(indices, np.uint32)
    fn = lib().emul_bvh_fingerprint
    fn.restype = C.c_uint64
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p]
    n = C.c_uint32(0)
    h = fn(v.ctypes.data, v.shape[0], i.ctypes.data, i.size, 1 if threaded else 0, C.addressof(n))
    return int(h), int(n.value)


def build_minmax_mips(heights):
    dem = np.ascontiguousarray(heights, np.float32)
    h, w = dem.shape
    tot = C.c_uint64(0)
    dims = np.zeros(32, np.uint32)
    n = lib().emul_build_minmax_mips(C.c_void_p(dem.ctypes.data), C.c_uint32(w), C.c_uint32(h), None,
                                     C.c_void_p(dims.ctypes.data), C.byref(tot))
    flat = np.zeros(tot.value, np.float32)
    lib().emul_build_minmax_mips(C.c_void_p(dem.ctypes.data), C.c_uint32(w), C.c_uint32(h),
                                 C.c_void_p(flat.ctypes.data), C.c_void_p(dims.ctypes.data), C.byref(tot))
    levels, off = [], 0
    for l in range(n):
        pw, ph = int(dims[2 * l]), int(dims[2 * l + 1])
        levels.append(flat[off:off + pw * ph * 2].reshape(ph, pw, 2))
        off += pw * ph * 2
    return levels


# ---- strip backend for forge3d_amd.distributed.StripRenderer (gloo CPU tests) -----------------
class _EmulStripSession:
    def __init__(self, handle, rows, width, stats):
        self._h, self.rows, self.width, self._stats = handle, rows, width, stats

    def enqueue_frames(self, first, count, collect=False):
        L = lib()
        for i in range(count):
            last = collect and i + 1 == count
            L.emul_session_frame(C.c_void_p(self._h), C.c_uint32(first + i), C.c_int32(1 if last else 0),
                                 C.c_void_p(self._stats.data_ptr()))

    def enqueue_frame_part(self, frame, part, collect=False):
        lib().emul_session_frame_part(C.c_void_p(self._h), C.c_uint32(frame), C.c_uint32(part),
                                      C.c_int32(1 if collect else 0), C.c_void_p(self._stats.data_ptr()))

    def window_stats(self):
        host = self._stats.numpy().astype(np.uint32)
        return float(host[:1].view(np.float32)[0]), bool(host[1])

    def resolve(self, frames):
        rows, w = self.rows, self.width
        rgba = np.zeros((rows, w, 4), np.uint8)
        alb = np.zeros((rows, w, 3), np.float32)
        nrm = np.zeros((rows, w, 3), np.float32)
        dep = np.zeros((rows, w), np.float32)
        valid = C.c_int32(0)
        rc = lib().emul_session_resolve(C.c_void_p(self._h), C.c_uint32(frames), C.c_void_p(rgba.ctypes.data),
                                        C.c_void_p(alb.ctypes.data), C.c_void_p(nrm.ctypes.data),
                                        C.c_void_p(dep.ctypes.data), C.byref(valid))
        if rc != 0:
            raise RuntimeError("emul resolve failed")
        return {"rgba": rgba, "albedo": alb, "normal": nrm, "depth": dep, "any_valid_reservoir": bool(valid.value)}

    def kernel_timing(self, enable):
        return 0.0, 0

    def sample_lanes(self):
        return 1

    def close(self):
        if self._h:
            lib().emul_session_destroy(C.c_void_p(self._h))
            self._h = None


class EmulBackend:
    """CPU stand-in for forge3d_amd.distributed.HipBackend (same kernel code, host-compiled)."""

    def __init__(self):
        import torch

        self.torch = torch

    def empty_bytes(self, n):
        return self.torch.zeros(n, dtype=self.torch.uint8)

    def empty_i32(self, n):
        return self.torch.zeros(n, dtype=self.torch.int32)

    def make_session(self, dem, width, height, cam, row_begin, row_end, res, stats, kw):
        kw = dict(kw)
        for k in ("memory_budget_bytes", "kernel_variant", "device", "stream"):
            kw.pop(k, None)
        defaults = dict(spacing=(1.0, 1.0), exaggeration=1.0, albedo=(0.6, 0.6, 0.6), sun_azimuth_deg=315.0,
                        sun_elevation_deg=45.0, sun_intensity=2.5, env_map=None, env_intensity=0.35,
                        mesh_vertices=None, mesh_indices=None, spp=1, max_frames=512, min_frames=32,
                        variance_threshold=1e-3, seed=7, sun_color=(1.0, 0.97, 0.92), observer_latitude_deg=0.0,
                        observer_longitude_deg=0.0, earth_model="ellipsoid", sphere_radius_m=6_371_008.8,
                        refraction_model="bennett", refraction_k=0.13, pressure_mbar=1013.25, temperature_c=15.0)
        defaults.update(kw)
        d, keep = _native.make_desc(dem, width, height, dict(cam or {}), **defaults)
        L = lib()
        L.emul_session_create.restype = C.c_void_p
        err = C.create_string_buffer(512)
        h = L.emul_session_create(C.byref(d), C.c_uint32(row_begin), C.c_uint32(row_end),
                                  C.c_void_p(res[0].data_ptr()), C.c_void_p(res[1].data_ptr()), err,
                                  C.c_size_t(len(err)))
        if not h:
            raise RuntimeError(err.value.decode())
        return _EmulStripSession(h, row_end - row_begin, width, stats)

    def sync(self):
        pass


def wavefront_render(scene, width, height, frames, first_frame=0, accum=None):
    """The multi-bounce PBR tracer's device path code (csrc/f3d_wf_path.h) compiled for the host:
    adds `frames` frames to `accum` and returns dict(accum, path_vertices)."""
    from forge3d_amd import wavefront as wfm

    if not isinstance(scene, dict):
        scene = scene.as_dict()
    s, keep = wfm._marshal(scene)
    acc = np.zeros((height, width, 4), np.float32) if accum is None else np.ascontiguousarray(accum, np.float32).copy()
    vertices = C.c_uint64(0)
    err = C.create_string_buffer(512)
    L = lib()
    L.emul_wavefront_render.restype = C.c_int
    rc = L.emul_wavefront_render(C.byref(s), C.c_uint32(width), C.c_uint32(height), C.c_uint32(first_frame), C.c_uint32(frames),
                                 acc.ctypes.data_as(C.c_void_p), C.byref(vertices), err, C.c_size_t(len(err)))
    del keep
    if rc != 0:
        _native.raise_status(rc, err.value.decode(errors="replace"))
    return {"accum": acc, "path_vertices": int(vertices.value)}


def smoke_step(state, emitters=(), steps=1, **settings):
    """The smoke solver's device code (csrc/f3d_smoke_sim.h) compiled for the host: `steps` steps in place on a
    smoke_oracle.new_state() dict."""
    from forge3d_amd import smoke as product
    from oracle import smoke_oracle as so

    v, keep = so.sim_structs(state, product._State)
    s = so.settings_struct(product._StepSettings, **settings)
    em = so.emitter_array(list(emitters), product._Emitter)
    L = lib()
    L.emul_smoke_step.restype = C.c_int
    if L.emul_smoke_step(C.byref(v), C.byref(s), em, C.c_uint32(len(emitters)), C.c_uint32(int(steps))) != 0:
        raise RuntimeError("emul_smoke_step failed")
    state["time_seconds"], state["frame_index"] = float(v.time_seconds), int(v.frame_index)
    return state


def composite(mode, base, layer, **kw):
    """The composite pass's per-pixel device code (csrc/f3d_composite.h) compiled for the host."""
    from forge3d_amd import smoke as product

    base = np.ascontiguousarray(base, dtype=np.uint8)
    layer = None if layer is None else np.ascontiguousarray(layer, dtype=np.uint8)
    desc = product.composite_desc(mode, base, layer, **kw)
    out = np.empty_like(base)
    L = lib()
    L.emul_composite.restype = C.c_int
    if L.emul_composite(C.byref(desc), C.c_void_p(out.ctypes.data)) != 0:
        raise RuntimeError("emul_composite failed")
    return out


def aether_reference(heightmap, width, height, cam, **kw) -> dict:
    """The AETHER acceptance reference's device code (csrc/f3d_aether_ref.h) compiled for the host."""
    from forge3d_amd import atmosphere as product
    from oracle import aether_ref_oracle as ao

    d = product._RefDesc()
    d.struct_size = C.sizeof(product._RefDesc)
    keep = ao.fill_desc(d, heightmap, width, height, cam, **kw)
    mean_xyz = np.zeros((height, width, 3), np.float32)
    rgb = np.zeros_like(mean_xyz)
    out = product._RefOut()
    out.mean_xyz, out.linear_rgb = mean_xyz.ctypes.data, rgb.ctypes.data
    err = C.create_string_buffer(512)
    L = lib()
    L.emul_aether_reference.restype = C.c_int
    rc = L.emul_aether_reference(C.byref(d), C.byref(out), err, len(err))
    del keep
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return {"mean_xyz": mean_xyz, "linear_rgb": rgb, "variance": float(out.variance), "converged": bool(out.converged),
            "terrain_primary_hits": int(out.terrain_primary_hits)}


def bvh4_nodes(vertices, indices):
    """(records of the 4-wide form of the mesh BVH -- 0 when the tree is too deep for it --, nodes of the binary tree)."""
    v = np.ascontiguousarray(vertices, np.float32)
    i = np.ascontiguousarray(indices, np.uint32)
    fn = lib().emul_bvh4_nodes
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    n = C.c_uint32(0)
    wide = fn(v.ctypes.data, v.shape[0], i.ctypes.data, i.size, C.addressof(n))
    return int(wide), int(n.value)


def set_mesh_walk(form: int):
    """0: the reference's sweep, 1: the threaded binary walk, 2: four children wide (default)."""
    fn = lib().emul_set_use_bvh
    fn.argtypes = [C.c_int32]
    fn(int(form))
