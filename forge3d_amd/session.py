"""Steppable, device-resident form of the terrain render (C ABI ``f3d_session_*``).

Used by bench.py (inputs resident in HBM before the timed region) and by the row-strip
multi-GPU driver (forge3d_amd/distributed.py).  One TerrainSession owns the image rows
[row_begin, row_end); RNG and all state are keyed by full-image coordinates.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native

HALO_ROWS = 4  # f3d_scene.h kHaloRows: the spatial pass reaches [-3, +4] rows
RESERVOIR_BYTES = 16
WELFORD_WINDOW = 32


class TerrainSession:
    def __init__(self, heightmap, width, height, camera=None, *, row_begin=0, row_end=0, device=-1, stream=0,
                 memory_budget_bytes=0, kernel_variant=0, ext_reservoirs=(None, None), ext_stats=None, bands=0,
                 band_streams=0, mesh_builder=0, frames_in_flight=0,
                 spacing=(1.0, 1.0), exaggeration=1.0, albedo=(0.6, 0.6, 0.6), sun_azimuth_deg=315.0,
                 sun_elevation_deg=45.0, sun_intensity=2.5, sun_color=(1.0, 0.97, 0.92), env_map=None,
                 env_intensity=0.35, mesh_vertices=None, mesh_indices=None, spp=1, max_frames=512, min_frames=32,
                 variance_threshold=1e-3, seed=7, observer_latitude_deg=0.0, observer_longitude_deg=0.0,
                 earth_model="ellipsoid", sphere_radius_m=6_371_008.8, refraction_model="bennett",
                 refraction_k=0.13, pressure_mbar=1013.25, temperature_c=15.0, atmosphere=None):
        self._lib = _native.lib()
        self._handle = C.c_void_p(None)
        desc, keep = _native.make_desc(heightmap, width, height, dict(camera or {}), spacing, exaggeration, albedo,
                                       sun_azimuth_deg, sun_elevation_deg, sun_intensity, env_map, env_intensity,
                                       mesh_vertices, mesh_indices, spp, max_frames, min_frames,
                                       variance_threshold, seed, sun_color, observer_latitude_deg,
                                       observer_longitude_deg, earth_model, sphere_radius_m, refraction_model,
                                       refraction_k, pressure_mbar, temperature_c, atmosphere)
        opts = _native.SessionOpts()
        opts.struct_size = C.sizeof(_native.SessionOpts)
        opts.device = int(device)
        opts.stream = C.c_void_p(int(stream) or None)
        opts.row_begin, opts.row_end = int(row_begin), int(row_end)
        opts.memory_budget_bytes = int(memory_budget_bytes)
        opts.kernel_variant = int(kernel_variant)
        opts.ext_reservoirs[0] = C.c_void_p(ext_reservoirs[0] or None)
        opts.ext_reservoirs[1] = C.c_void_p(ext_reservoirs[1] or None)
        opts.ext_stats = C.c_void_p(ext_stats or None)
        opts.bands, opts.band_streams = int(bands), int(band_streams)
        opts.mesh_builder = int(mesh_builder)
        opts.frames_in_flight = int(frames_in_flight)
        err = C.create_string_buffer(1024)
        rc = self._lib.f3d_session_create(C.byref(desc), C.byref(opts), C.byref(self._handle), err, len(err))
        del keep
        if rc != 0:
            self._handle = C.c_void_p(None)
            _native.raise_status(rc, err.value.decode("utf-8", "replace"))
        self.width, self.height = int(width), int(height)
        self.row_begin = int(row_begin)
        self.row_end = int(row_end) or int(height)
        self.rows = self.row_end - self.row_begin
        self.spp = int(spp)
        self._err = err

    # -- lifecycle --------------------------------------------------------------------
    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.f3d_session_destroy(self._handle)
            self._handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            _native.raise_status(rc, self._err.value.decode("utf-8", "replace"))

    # -- stepping ---------------------------------------------------------------------
    def enqueue_frames(self, first_frame: int, count: int, collect_stats: bool = False):
        """Asynchronously enqueue accumulation frames on the session stream."""
        self._check(self._lib.f3d_session_enqueue_frames(self._handle, int(first_frame), int(count),
                                                         1 if collect_stats else 0, self._err, len(self._err)))

    def enqueue_trace(self, first_frame: int, count: int):
        """Frames in flight: trace frames [first_frame, first_frame + count) in one launch (count <= frames_in_flight())."""
        self._check(self._lib.f3d_session_enqueue_trace(self._handle, int(first_frame), int(count), self._err, len(self._err)))

    def enqueue_merge(self, frame: int, collect_stats: bool = False):
        """Frames in flight: the ordered half (reservoir chain, accumulation) of one traced frame."""
        self._check(self._lib.f3d_session_enqueue_merge(self._handle, int(frame), 1 if collect_stats else 0, self._err, len(self._err)))

    def frames_in_flight(self) -> int:
        """Effective number of frames the session traces per launch (0: every frame is one fused launch)."""
        return int(self._lib.f3d_session_frames_in_flight(self._handle))

    def trace_batch(self, frame: int, remaining: int) -> int:
        """Batch size to trace next at `frame` (short batches first, then frames_in_flight())."""
        return int(self._lib.f3d_session_trace_batch(self._handle, int(frame), int(remaining)))

    def retraced_pixels(self) -> int:
        """Diagnostics (synchronises): pixel-frames whose sun-direction prediction failed and were traced again."""
        total = C.c_uint64(0)
        self._lib.f3d_session_retraced_pixels(self._handle, C.byref(total))
        return int(total.value)

    FINGERPRINT_FIELDS = ("camera", "light", "terrain_scalars", "mesh_scalars", "scalars", "leaf_table", "band_tables",
                          "mesh_vertices", "mesh_indices", "bvh_nodes", "bvh_triangles", "environment", "gbuffer",
                          "reservoirs", "accumulation", "frame_heads")

    def fingerprint(self) -> dict:
        """Diagnostics (synchronises): hashes of everything a frame launch reads, by name."""
        out = (C.c_uint64 * 16)()
        if self._lib.f3d_session_fingerprint(self._handle, out, 16) != 0:
            raise RuntimeError("f3d_session_fingerprint failed")
        return dict(zip(self.FINGERPRINT_FIELDS, (int(v) for v in out)))

    def enqueue_frame_part(self, frame: int, part: int, collect_stats: bool = False):
        """One frame in two launches: part 1 = head + the strip's edge rows (the halo donors), part 2 = interior."""
        self._check(self._lib.f3d_session_enqueue_frame_part(self._handle, int(frame), int(part),
                                                             1 if collect_stats else 0, self._err, len(self._err)))

    # -- peer halos (strips of one node without the host or a collective in the frame chain) -------------
    def halo_export(self) -> bytes:
        """What a neighbouring strip needs to map this strip's reservoirs and its frame counter (f3d_halo_export bytes)."""
        rec = _native.HaloExport()
        self._check(self._lib.f3d_session_halo_export(self._handle, C.byref(rec), self._err, len(self._err)))
        return bytes(rec)

    def halo_connect(self, side: int, export: bytes):
        """side 0: the strip above (smaller rows), 1: the strip below; export = that strip's halo_export()."""
        rec = _native.HaloExport.from_buffer_copy(export)
        self._check(self._lib.f3d_session_halo_connect(self._handle, int(side), C.byref(rec), self._err, len(self._err)))

    def halo_probe_publish(self, nonce: int):
        """Link check, step 1: store `nonce` into this strip's counter block (the store the frame counter uses)."""
        self._check(self._lib.f3d_session_halo_probe(self._handle, 0, int(nonce) & 0xFFFFFFFF, None, self._err, len(self._err)))

    def halo_probe_read(self):
        """Link check, step 2 (after a barrier): the neighbours' words as this device reads them: (above, below)."""
        seen = (C.c_uint32 * 2)()
        self._check(self._lib.f3d_session_halo_probe(self._handle, 1, 0, seen, self._err, len(self._err)))
        return int(seen[0]), int(seen[1])

    def halo_probe_fill(self, nonce: int):
        """Link check with a real block, step 1: fill this strip's two edge blocks of reservoir buffer 0 with the pattern of
        `nonce` (a many-workgroup kernel, as the frame kernels leave their rows) and publish the nonce behind it."""
        self._check(self._lib.f3d_session_halo_probe(self._handle, 2, int(nonce) & 0xFFFFFFFF, None, self._err, len(self._err)))

    def halo_probe_pull(self, nonce_above: int, nonce_below: int):
        """Step 2: wait (on the device) for the neighbours' nonces, pull their blocks with the frame loop's kernel and compare
        the sums with the patterns'; raises if a block is not what its owner wrote."""
        seen = (C.c_uint32 * 2)(int(nonce_above) & 0xFFFFFFFF, int(nonce_below) & 0xFFFFFFFF)
        self._check(self._lib.f3d_session_halo_probe(self._handle, 3, 0, seen, self._err, len(self._err)))
        return int(seen[0]), int(seen[1])

    def halo_probe_clear(self):
        """Step 3, once EVERY strip has pulled (a barrier): reservoir buffer 0 as a new session has it."""
        self._check(self._lib.f3d_session_halo_probe(self._handle, 4, 0, None, self._err, len(self._err)))

    def halo_timeouts(self) -> int:
        """Device-side halo waits of the last enqueue_batch_strip that gave up (a neighbour that stopped); synchronises."""
        n = C.c_uint32(0)
        self._check(self._lib.f3d_session_halo_status(self._handle, C.byref(n), self._err, len(self._err)))
        return int(n.value)

    def halo_stats(self, reset: bool = False) -> dict:
        """How long this strip's pulls stood waiting for its neighbours (device time, ms) since the last reset; synchronises."""
        rec = _native.HaloStats()
        rec.reset = 1 if reset else 0
        self._check(self._lib.f3d_session_halo_stats(self._handle, C.byref(rec), self._err, len(self._err)))
        return {"frames_published": int(rec.frames_published), "timeouts": int(rec.timeouts), "pulls": int(rec.pulls),
                "wait_ms": (float(rec.wait_ms[0]), float(rec.wait_ms[1])), "longest_wait_ms": float(rec.longest_wait_ms),
                "timeout_ms": float(rec.timeout_ms)}

    def enqueue_batch_strip(self, first_frame: int, count: int, collect_stats: bool = False):
        """Frames [first_frame, first_frame + count) of a connected strip in ONE call: per frame its kernels, the frame
        counter, the pull of both neighbours' edge rows -- no host synchronisation, no collective."""
        self._check(self._lib.f3d_session_enqueue_batch_strip(self._handle, int(first_frame), int(count), 1 if collect_stats else 0,
                                                              self._err, len(self._err)))

    def window_stats(self):
        """(max Welford m2 over the owned pixels, saw non-finite) -- synchronises the stream."""
        m2, bad = C.c_float(0.0), C.c_int32(0)
        self._check(self._lib.f3d_session_window_stats(self._handle, C.byref(m2), C.byref(bad), self._err,
                                                       len(self._err)))
        return float(m2.value), bool(bad.value)

    def halo(self, which: int, side: int):
        """(device pointer, bytes) of a halo block (HALO_ROWS rows) of reservoir buffer `which`."""
        ptr, nbytes = C.c_void_p(None), C.c_uint64(0)
        rc = self._lib.f3d_session_halo(self._handle, int(which), int(side), C.byref(ptr), C.byref(nbytes))
        if rc != 0:
            raise ValueError("invalid halo query")
        return int(ptr.value), int(nbytes.value)

    def set_accumulation(self, sums_rgba):
        """Replace the accumulated radiance sums by (rows, width, 4) f32 sums of the caller's (composition hook: the PBR
        tracer's radiance through this session's resolve and AETHER post)."""
        arr = np.ascontiguousarray(sums_rgba, np.float32)
        if arr.shape != (self.rows, self.width, 4):
            raise ValueError(f"accumulation must have shape ({self.rows}, {self.width}, 4)")
        self._check(self._lib.f3d_session_set_accumulation(self._handle, arr.ctypes.data, self._err, len(self._err)))

    def resolve(self, frames: int):
        """Final resolve of the owned rows into host arrays."""
        rows, w = self.rows, self.width
        rgba = np.zeros((rows, w, 4), np.uint8)
        alb = np.zeros((rows, w, 3), np.float32)
        nrm = np.zeros((rows, w, 3), np.float32)
        dep = np.zeros((rows, w), np.float32)
        valid = C.c_int32(0)
        self._check(self._lib.f3d_session_resolve(self._handle, int(frames), rgba.ctypes.data, alb.ctypes.data,
                                                  nrm.ctypes.data, dep.ctypes.data, C.byref(valid), self._err,
                                                  len(self._err)))
        return {"rgba": rgba, "albedo": alb, "normal": nrm, "depth": dep, "any_valid_reservoir": bool(valid.value)}

    def resolve_device(self, frames: int, d_rgba=0, d_albedo=0, d_normal=0, d_depth=0):
        self._check(self._lib.f3d_session_resolve_device(self._handle, int(frames), C.c_void_p(d_rgba or None),
                                                         C.c_void_p(d_albedo or None), C.c_void_p(d_normal or None),
                                                         C.c_void_p(d_depth or None), self._err, len(self._err)))

    def info(self):
        g, p, h = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rows, width = C.c_uint32(0), C.c_uint32(0)
        self._lib.f3d_session_info(self._handle, C.byref(g), C.byref(p), C.byref(h), C.byref(rows), C.byref(width))
        return {"gpu_resource_bytes": int(g.value), "minmax_pyramid_bytes": int(p.value),
                "peak_host_visible_bytes": int(h.value), "rows": int(rows.value), "width": int(width.value)}

    def setup_ms(self) -> dict:
        """Host wall time of the session's creation by phase (ms): what a render pays once before its first frame."""
        out = (C.c_double * 8)()
        self._lib.f3d_session_setup_ms(self._handle, out, 8)
        keys = ("total", "validate", "hash", "upload", "tables", "scene", "alloc", "passes")
        return {k: float(out[i]) for i, k in enumerate(keys)}

    def sample_lanes(self) -> int:
        """Sample lanes per pixel of the frame kernel (1, 2, 4 or 8; chosen from the strip size and spp)."""
        return int(self._lib.f3d_session_sample_lanes(self._handle))

    def row_costs(self) -> np.ndarray:
        """Cost of the last fused frame by image row of this strip (float32[rows], wave time in 100 MHz ticks); synchronises."""
        out = np.zeros(self.rows, np.float32)
        self._check(self._lib.f3d_session_row_costs(self._handle, out.ctypes.data_as(C.POINTER(C.c_float)), self.rows, self._err, len(self._err)))
        return out

    def primary_start_ptr(self) -> int:
        """Device pointer to the session's primary-ray certificates (rows x width records, f3d_cone.h), 0 if it has none; valid
        while the session lives.  The PBR path tracer takes it (WavefrontScene / f3d_wf_scene.primary_start)."""
        return int(self._lib.f3d_session_primary_start(self._handle) or 0)

    def kernel_timing(self, enable: bool):
        """enable=True starts recording a hipEvent pair around every frame launch;
        enable=False stops and returns (average ms per launch, launches)."""
        avg, n = C.c_double(0.0), C.c_uint32(0)
        rc = self._lib.f3d_session_kernel_timing(self._handle, 1 if enable else 0, C.byref(avg), C.byref(n))
        if rc != 0:
            raise RuntimeError("kernel timing failed")
        return float(avg.value), int(n.value)


def kernel_variant(*, sample_lanes: int = 0, waves_per_simd: int = 0, tile_map: int = 0, leaf_quorum: int = 0, share_below: int = 0) -> int:
    """f3d_session_opts.kernel_variant from names (include/f3d_terrain_pt.h lists the decimal fields): every A/B switch of the
    frame kernel that is not a build flag.  All zero = the shipped default."""
    if sample_lanes not in (0, 1, 2, 4, 8):
        raise ValueError("sample_lanes must be 0 (automatic), 1, 2, 4 or 8")
    if waves_per_simd not in (0, 4, 5, 6):
        raise ValueError("waves_per_simd must be 0 / 6 (default), 4 (1 sample lane) or 5 (4 sample lanes)")
    if not (0 <= tile_map <= 4 and 0 <= leaf_quorum <= 64 and 0 <= share_below <= 64):
        raise ValueError("tile_map in 0..4, leaf_quorum and share_below in 0..64")
    budget = 0 if waves_per_simd in (0, 6) else 100 + waves_per_simd
    return budget + 1000 * tile_map + 10000 * leaf_quorum + 1000000 * sample_lanes + 10000000 * share_below


def describe_kernel_variant(v: int) -> dict:
    """The fields of a kernel_variant, by name."""
    v = int(v)
    budget = v % 1000
    return {"waves_per_simd": 6 if budget == 0 else (budget - 100), "tile_map": (v // 1000) % 10, "leaf_quorum": (v // 10000) % 100,
            "sample_lanes": (v // 1000000) % 10, "share_below": (v // 10000000) % 100}


def reservoir_buffer_bytes(rows: int, width: int) -> int:
    return (rows + 2 * HALO_ROWS) * width * RESERVOIR_BYTES
