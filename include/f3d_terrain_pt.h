/* include/f3d_terrain_pt.h -- C ABI of libf3dhip.so
 *
 * MI355X-native (gfx950, HIP) replacement for ONE path of forge3d: the PROMETHEUS
 * terrain path tracer behind `forge3d.hybrid_render_terrain_reference`.  These entry
 * points are what the reference's FFI for this path would bind (INTEGRATION.md shows
 * the Rust `extern "C"` block and the PyO3 shim a forge3d maintainer would add).
 * Plain pointers and sizes only; no torch / HIP types in any signature (streams and
 * device buffers travel as `void*`).  All functions are blocking unless stated and
 * thread-compatible (no hidden process-wide state besides the HIP primary context).
 *
 * Status codes: 0 ok; 1 value error (-> Python ValueError); 2 render error
 * (reference RenderError::Render -> RuntimeError "[Render] Render error: ...");
 * 3 upload error (RenderError::Upload); 4 device error (no GPU / HIP failure --
 * there is NO CPU fallback: without a usable gfx950 device every compute entry
 * point fails with status 4).
 */
#ifndef F3D_TERRAIN_PT_H
#define F3D_TERRAIN_PT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever a struct of this header changes size or layout.  Every struct that has grown
 * (or may grow) carries its own size as its FIRST member; the caller sets it to sizeof(the struct it was compiled
 * against) and the library refuses a size it does not know (status 1) instead of reading past the caller's struct.
 *   1  round 1: f3d_terrain_ref_desc without `atmosphere`, no struct_size members
 *   2  round 2: + f3d_terrain_ref_desc.atmosphere (binary-incompatible, unversioned -- the reason for this scheme)
 *   3  round 3: + struct_size first in f3d_terrain_ref_desc / f3d_session_opts / f3d_wf_scene (and in the new
 *               f3d_composite_desc, f3d_aether_ref_desc), f3d_abi_version(); f3d_wf_scene + terrain, hair, medium;
 *               new entry points: peer halos (f3d_session_halo_*, f3d_session_enqueue_batch_strip),
 *               f3d_session_set_accumulation, f3d_smoke_step, f3d_smoke_composite, f3d_aether_reference_render
 *   4  round 4: + f3d_session_halo_stats / f3d_halo_stats, f3d_session_halo_probe modes 2 and 3 (no existing struct
 *               changed: a caller built against version 3 keeps working)
 *   5  round 5: + f3d_session_row_costs, f3d_session_primary_start, f3d_smoke_set_stream, f3d_smoke_wait_fields_read; the
 *               smoke entry points return without waiting when their results stay on the device and no time is asked for;
 *               f3d_wf_scene + primary_start at its end (no existing signature changed)
 *   6  round 6: + f3d_smoke_seq_* (a smoke sequence behind a handle: its streams, its ordering, its scratch).  The handle-less
 *               smoke entry points are synchronous again, as in ABI 4, unless the thread has called f3d_smoke_set_stream --
 *               which, with f3d_smoke_wait_fields_read, stays for this one revision as a shim over the thread's default
 *               context.  No struct changed: a caller built against version 3, 4 or 5 keeps working. */
#define F3D_ABI_VERSION 6u
#define F3D_STATUS_OK 0
#define F3D_STATUS_VALUE 1
#define F3D_STATUS_RENDER 2
#define F3D_STATUS_UPLOAD 3
#define F3D_STATUS_DEVICE 4

/* EarthModel / RefractionModel of reference src/geo/refraction.rs:15-43 */
#define F3D_EARTH_FLAT 0
#define F3D_EARTH_SPHERE 1
#define F3D_EARTH_ELLIPSOID 2
#define F3D_REFRACTION_NONE 0
#define F3D_REFRACTION_BENNETT 1
#define F3D_REFRACTION_SAEMUNDSSON 2
#define F3D_REFRACTION_EFFECTIVE_RADIUS 3

/* AETHER LUT payload handed to the aerial-perspective post: the reference's AtmosphereLutHandle
 * (src/core/atmosphere/runtime.rs:38-89; payload layout src/core/atmosphere/precomputed.rs:5-25) as plain host
 * pointers.  Tables are RGBA16F bit patterns, x fastest: transmittance [height][mu], accumulated scattering
 * [height * nu_count + nu][mu_sun][mu_view], aerial [height][mu_view][distance] (rgb = 0, a = mean transmittance). */
typedef struct f3d_aether_luts {
    const uint16_t *transmittance, *accumulated_scattering, *aerial;
    uint32_t transmittance_mu, transmittance_height;
    uint32_t scattering_mu_view, scattering_mu_sun, scattering_height, scattering_nu;
    uint32_t aerial_distance, aerial_mu_view, aerial_height;
    /* AtmosphereConfig, src/core/atmosphere/bake.rs:131-162 */
    float turbidity, ozone_du, mie_g, bottom_radius_m, top_radius_m, rayleigh_scale_height_m, mie_scale_height_m,
        max_aerial_distance_m, ground_albedo;
    uint32_t scattering_orders;
} f3d_aether_luts;

/* Mirrors TerrainReferenceDesc (reference
 * src/path_tracing/hybrid_compute/render_terrain.rs:239-282) plus the earth /
 * refraction parameters the PyO3 seam parses (src/py_functions/path_tracing/
 * terrain_reference.rs:295-312).  All pointers are HOST pointers that are only read
 * during the call. */
typedef struct f3d_terrain_ref_desc {
    uint32_t struct_size; /* = sizeof(f3d_terrain_ref_desc) of the caller's header (see F3D_ABI_VERSION) */
    const float *heights; /* (dem_height, dem_width) row-major f32 */
    uint32_t dem_width, dem_height;
    float spacing_x, spacing_z;
    float exaggeration;
    float albedo[3];
    float cam_origin[3], cam_look_at[3], cam_up[3];
    float fov_y_deg, exposure;
    float sun_azimuth_deg, sun_elevation_deg, sun_intensity;
    float sun_color[3];
    double observer_latitude_deg, observer_longitude_deg;
    int32_t earth_model;      /* F3D_EARTH_* */
    int32_t refraction_model; /* F3D_REFRACTION_* */
    double sphere_radius_m, refraction_k, pressure_mbar, temperature_c;
    const float *env_map; /* (env_height, env_width, 3) or NULL */
    uint32_t env_width, env_height;
    float env_intensity;
    const float *mesh_vertices; /* (mesh_vertex_count, 3) or NULL */
    uint32_t mesh_vertex_count;
    const uint32_t *mesh_indices; /* mesh_index_count entries, 3 per triangle, or NULL */
    uint32_t mesh_index_count;
    uint32_t width, height;
    uint32_t seed, spp, max_frames, min_frames;
    float variance_threshold;
    /* TerrainReferenceDesc::atmosphere (render_terrain.rs:265): NULL = no aerial perspective; else the converged
     * accumulation goes through the AETHER post (aether_post.rs, prometheus_aerial.wgsl) before the resolve. */
    const f3d_aether_luts *atmosphere;
} f3d_terrain_ref_desc;

/* Mirrors TerrainReferenceOutput (render_terrain.rs:285-299).  The four image
 * buffers are CALLER-allocated host memory; the library never frees or retains
 * them. */
typedef struct f3d_terrain_ref_out {
    uint8_t *rgba; /* (height, width, 4) */
    float *albedo; /* (height, width, 3) */
    float *normal; /* (height, width, 3) */
    float *depth;  /* (height, width), qNaN on miss */
    uint32_t frames;
    float variance;
    int32_t converged;
    uint64_t peak_host_visible_bytes;
    uint64_t minmax_pyramid_bytes;
    uint64_t gpu_resource_bytes;
    double loop_seconds;     /* accumulation loop only (device time, host clock) */
    double setup_seconds;    /* pyramid build + uploads + G-buffer pass */
    double readback_seconds; /* resolve + copies back */
} f3d_terrain_ref_out;

/* Replaces `_forge3d.hybrid_render_terrain_reference`
 * (reference src/py_functions/path_tracing/terrain_reference.rs:221-457 ->
 * HybridPathTracer::render_terrain_reference, render_terrain.rs:563-1434). */
int f3d_terrain_ref_render(const f3d_terrain_ref_desc *desc, f3d_terrain_ref_out *out, char *err,
                           size_t errlen);

/* ---- session API: the same render, device-resident and steppable -------------
 * Used by bench.py (inputs resident in HBM before the timed region) and by the
 * row-strip multi-GPU driver (forge3d_amd/distributed.py).  A session owns the
 * pixel rows [row_begin, row_end) of the full width x height image; RNG and all
 * state are keyed by full-image coordinates, so any partition reproduces the
 * single-GPU image. */
typedef struct f3d_session f3d_session;

#define F3D_FRAMES_IN_FLIGHT_AUTO 0xFFFFFFFFu
typedef struct f3d_session_opts {
    uint32_t struct_size; /* = sizeof(f3d_session_opts) of the caller's header */
    int32_t device;      /* HIP device ordinal, -1 = current */
    void *stream;        /* hipStream_t to enqueue on, NULL = the null stream */
    uint32_t row_begin;  /* first owned image row */
    uint32_t row_end;    /* one past the last owned row; 0 = height */
    uint64_t memory_budget_bytes; /* 0 = 512 MiB (reference MEMORY_BUDGET_LIMIT) */
    int32_t kernel_variant;       /* 0 = default.  A/B switches as decimal fields (forge3d_amd.session.kernel_variant() builds one
                                   * from names, describe_kernel_variant() reads one back):
                                   *   v % 1000              register budget of the frame kernel: 0 = 6 waves per SIMD (80 VGPRs);
                                   *                         104 = 4 waves (1 sample lane), 105 = 5 waves (4 sample lanes): the two
                                   *                         occupancy A/Bs profiles/README.md still cites; anything else is refused
                                   *   (v / 1000) % 10       tile -> XCD map: 0 / 2 rows dealt round-robin + longest-first dispatch
                                   *                         (default), 1 consecutive tiles, 3 contiguous bands, 4 = 2 without longest-first
                                   *   (v / 10000) % 100     lanes with a queued leaf that trigger a drain (0 = 64: only a full FIFO does)
                                   *   (v / 1000000) % 10    sample lanes per pixel: 1, 2, 4, 8; 0 = automatic (a non-zero register
                                   *                         budget with 0 here selects the 1-lane kernel)
                                   *   (v / 10000000) % 100  ray sharing: deal the IBL rays when at most this many lanes still march
                                   *                         (0 = 16; 64 = from the first step)
                                   * e.g. 4000105 = 4 sample lanes at 5 waves per SIMD; 8001000 = 8 lanes, consecutive tiles */
    /* Optional caller-owned DEVICE buffers (NULL -> the library allocates).  The
     * strip driver allocates these as torch tensors so RCCL can move them.
     *   reservoirs[2]: ping-pong packed reservoirs, each (rows + 8) * width * 16 B,
     *                  local row r holds image row row_begin - 4 + r;
     *   stats:         4 x u32 {max m2 bits, nonfinite flag, any_valid flag, bad_reservoir flag}. */
    void *ext_reservoirs[2];
    void *ext_stats;
    /* Band pipelining: the strip is cut into `bands` horizontal bands (0 = automatic: 1 for strips that fill
     * the chip, several for thin multi-GPU strips) whose launches go round-robin to `band_streams` internal
     * HIP streams (0 = automatic); band b of frame f + 1 waits only for bands b - 1, b, b + 1 of frame f (the
     * spatial reuse reads -3 .. +4 rows), so consecutive frames overlap and a thin strip is no longer bound by the
     * latency of one wave's ray chain.  Results do not depend on either number. */
    uint32_t bands;
    uint32_t band_streams;
    /* Acceleration structure of the optional triangle mesh: 0 = automatic (host SAH unless F3D_MESH_BVH=lbvh is
     * set in the environment), 1 = binned-SAH build on the host (reference accel::build_bvh, src/accel/sah_cpu.rs),
     * 2 = linear BVH built on the GPU (reference src/accel/lbvh_gpu: Morton codes, radix sort, Karras topology,
     * bottom-up refit).  Images do not depend on the choice. */
    uint32_t mesh_builder;
    /* Frames in flight: 0 / 1 = every frame is one fused launch (the default).  N >= 2: the session keeps a record
     * buffer for up to N frames (32 B x spp x pixels each, counted against memory_budget_bytes; N is lowered to what
     * fits) and f3d_session_enqueue_frames traces N frames in ONE launch -- what a frame traces does not depend on
     * the frames before it -- and then runs the cheap ordered half (reservoir chain, accumulation) per frame.  No
     * per-frame tail: the way thin multi-GPU strips stay throughput-bound.  Results do not depend on N.
     * F3D_FRAMES_IN_FLIGHT_AUTO: 16 for images too small to fill the chip with one frame, else 0 (what
     * f3d_terrain_ref_render uses).  f3d_session_frames_in_flight reports the effective value. */
    uint32_t frames_in_flight;
} f3d_session_opts;

int f3d_session_create(const f3d_terrain_ref_desc *desc, const f3d_session_opts *opts,
                       f3d_session **session, char *err, size_t errlen);
void f3d_session_destroy(f3d_session *session);

/* Enqueue accumulation frames [first_frame, first_frame + count) on the session
 * stream; asynchronous.  Multi-strip callers enqueue one frame at a time and
 * exchange the 4-row halos of reservoir buffer (frame & 1) between frames.  When
 * collect_stats_on_last != 0 the last frame of the batch also publishes the
 * convergence statistic read by f3d_session_window_stats. */
int f3d_session_enqueue_frames(f3d_session *session, uint32_t first_frame, uint32_t count,
                               int32_t collect_stats_on_last, char *err, size_t errlen);
/* Sessions with frames in flight, driven frame by frame (multi-GPU strips exchange halos between merges): trace
 * frames [first_frame, first_frame + count), count <= frames in flight, then merge each of them in order. */
int f3d_session_enqueue_trace(f3d_session *session, uint32_t first_frame, uint32_t count, char *err, size_t errlen);
int f3d_session_enqueue_merge(f3d_session *session, uint32_t frame, int32_t collect_stats, char *err, size_t errlen);
uint32_t f3d_session_frames_in_flight(f3d_session *session);
/* The batch size f3d_session_enqueue_frames would trace next at `frame` with `remaining` frames to go (short batches
 * for the first frames, then frames_in_flight): what a frame-by-frame driver passes to f3d_session_enqueue_trace. */
uint32_t f3d_session_trace_batch(f3d_session *session, uint32_t frame, uint32_t remaining);
/* Diagnostics (synchronises): pixel-frames traced a second time because the sun direction their frame head read was
 * mispredicted (see frames_in_flight); a handful per frame after the first two. */
int f3d_session_retraced_pixels(f3d_session *session, uint64_t *total);
/* One frame in two steps, for strips of a multi-GPU job: part 1 = the strip's EDGE bands (they contain the
 * first and last 4 pixel rows, the halo a neighbouring strip needs; on return the session stream is ordered
 * after them), part 2 = the interior bands.  The caller starts the halo exchange between the two, so that it
 * overlaps the interior.  Same result as f3d_session_enqueue_frames(frame, 1). */
int f3d_session_enqueue_frame_part(f3d_session *session, uint32_t frame, uint32_t part, int32_t collect_stats,
                                   char *err, size_t errlen);
/* Variance gate input for the window that ends after `frames` frames
 * (render_terrain.rs:1206-1231): synchronises the stream, returns max m2 over the
 * owned pixels (NOT yet divided by n-1) and whether a non-finite m2 was seen. */
int f3d_session_window_stats(f3d_session *session, float *max_m2, int32_t *nonfinite, char *err,
                             size_t errlen);
/* Device pointer + byte size of the halo rows of reservoir buffer `which` (0/1):
 * side 0 = the f3d_halo_rows() owned rows at the top (to send up), 1 = the owned rows at the
 * bottom (to send down), 2 = halo above the strip (to receive), 3 = halo below. */
int f3d_session_halo(f3d_session *session, int32_t which, int32_t side, void **ptr, uint64_t *bytes);
/* ---- peer halos: the strips of one node without the host or a collective in the frame chain --------------------
 * The classic strip loop (forge3d_amd/distributed.py) posts an RCCL send / recv pair from Python after every frame.
 * Here a strip PULLS its neighbours' edge rows itself: every session keeps a counter "frames merged" in device memory;
 * after a frame the library launches k_halo_pull, which raises that counter (it runs behind the frame's kernels) and
 * waits -- on the device -- until the neighbour's counter says its frame is done, then copies the neighbour's 4 edge
 * rows (peer memory mapped with hipIpcOpenMemHandle, read over xGMI with cache-bypassing loads) into the strip's own
 * halo rows.  Only READS cross the link and every strip writes nothing but its own memory, so no cache of another
 * device can hold a stale line.  f3d_session_enqueue_batch_strip enqueues a whole window of frames that way in one
 * call; RCCL is left with the per-window all-reduce and the final gather.
 *   f3d_session_halo_export   what a neighbour needs to map this strip (the session must own its reservoirs: no
 *                             ext_reservoirs); plain bytes, moved between the ranks by any means (all_gather)
 *   f3d_session_halo_connect  side 0 = the strip above (smaller rows), 1 = the strip below; peer = its export
 *   f3d_session_halo_probe    link check before the first frame: mode 0 stores `nonce` into this strip's counter block the
 *                             way the frame counter is stored, mode 1 reads the neighbours' words back (seen[0] = above,
 *                             seen[1] = below; 0 where there is no neighbour) with the loads the pull uses.  With a barrier
 *                             of the caller's between the two, a neighbour whose word does not read back (no peer access
 *                             between the devices, a stale mapping) is found before a frame depends on it
 *                             Modes 2 / 3 do the same with a REAL block: mode 2 fills this strip's two edge blocks of
 *                             reservoir buffer 0 with a pattern of `nonce` (a many-workgroup kernel, as the frame kernels
 *                             leave their rows) and publishes the nonce behind it; mode 3 (seen[0] / seen[1] = the nonces the
 *                             strips above / below published) waits for them on the device, pulls both blocks with the frame
 *                             loop's own kernel and compares their sums with the pattern's: seen[side] = 1 / 0, status 4 if a
 *                             block is not what its owner wrote.  Mode 4 clears reservoir buffer 0 again -- only after EVERY
 *                             strip has finished mode 3 (a barrier of the caller's: a neighbour may still be pulling this
 *                             strip's rows), and before any strip renders (another barrier)
 *   f3d_session_halo_status   device-side wait time-outs of the LAST f3d_session_enqueue_batch_strip (a dead neighbour must
 *                             not hang the GPU: a wait gives up after F3D_HALO_TIMEOUT_MS, default 20 000, and counts here;
 *                             a strip that has timed out stops pulling for the rest of that call -- its halo rows are stale --
 *                             and the caller turns a non-zero count into an error on every rank)
 *   f3d_session_halo_stats    how long this strip's pulls stood waiting for its neighbours since the last reset
 * Frame numbers of a connected session only rise: f3d_session_enqueue_batch_strip refuses a frame it has enqueued before
 * (the neighbours' counters are never cleared). */
typedef struct f3d_halo_stats {
    uint32_t reset;            /* in: 1 = clear the wait times and the pull count after reading them */
    uint32_t frames_published; /* this strip's frame counter */
    uint32_t timeouts, pulls;  /* waits that gave up (last call); neighbour blocks pulled */
    double wait_ms[2];         /* total device time the pulls waited for the strip above / below */
    double longest_wait_ms;    /* longest single wait */
    double timeout_ms;         /* the limit in force */
} f3d_halo_stats;
int f3d_session_halo_stats(f3d_session *session, f3d_halo_stats *out, char *err, size_t errlen);
typedef struct f3d_halo_export {
    uint8_t handle[3][64]; /* hipIpcMemHandle_t of reservoir buffer 0, 1 and of the counter block */
    uint64_t offset[3];    /* byte offset of the object inside the exported allocation */
    uint64_t address[3];   /* the objects' addresses in the exporting process (what a neighbour of the SAME process uses) */
    uint32_t rows, width;  /* owned rows of the strip, image width */
    int32_t device;        /* HIP device ordinal of the exporting process */
    uint32_t pid;          /* exporting process: a process cannot open its own handles */
} f3d_halo_export;
int f3d_session_halo_export(f3d_session *session, f3d_halo_export *out, char *err, size_t errlen);
int f3d_session_halo_connect(f3d_session *session, int32_t side, const f3d_halo_export *peer, char *err, size_t errlen);
int f3d_session_halo_probe(f3d_session *session, int32_t mode, uint32_t nonce, uint32_t *seen, char *err, size_t errlen);
int f3d_session_halo_status(f3d_session *session, uint32_t *timeouts, char *err, size_t errlen);
/* Frames [first, first + count) of a connected strip: per frame the frame's kernels (fused, or trace batch + merge with
 * frames in flight), the counter, the pull of both neighbours' rows.  One call, no host synchronisation. */
int f3d_session_enqueue_batch_strip(f3d_session *session, uint32_t first_frame, uint32_t count, int32_t collect_stats_on_last,
                                    char *err, size_t errlen);
/* Final resolve (last spatial reuse pass, reservoir validity, Reinhard + f16 + u8
 * quantisation, AOV conversion) and copy of the owned rows into host buffers that
 * cover ONLY the strip ((rows, width, C) each).  frames = accumulated frame count. */
int f3d_session_resolve(f3d_session *session, uint32_t frames, uint8_t *rgba, float *albedo,
                        float *normal, float *depth, int32_t *any_valid_reservoir, char *err,
                        size_t errlen);
/* Composition hook: replace the session's accumulated radiance sums by caller-supplied ones ((rows, width, 4) f32 host
 * memory: rgb = sums over `frames` frames, a ignored); the next f3d_session_resolve(frames) then tone-maps -- and, with
 * desc.atmosphere, sends through the AETHER post -- THAT radiance over the session's own depth / G-buffer.  How
 * BASELINE.json configs[2] combines the PBR tracer's multi-bounce radiance over the DEM (f3d_wavefront.h terrain
 * primitive) with the atmosphere post: forge3d_amd.offline.render_terrain_gi. */
int f3d_session_set_accumulation(f3d_session *session, const float *sums_rgba, char *err, size_t errlen);
/* Same, but leaves the results in caller-owned DEVICE buffers (for an RCCL gather). */
int f3d_session_resolve_device(f3d_session *session, uint32_t frames, void *d_rgba, void *d_albedo,
                               void *d_normal, void *d_depth, char *err, size_t errlen);
/* Host wall time of f3d_session_create by phase, ms: out[0] total, [1] validation + uniforms, [2] hashing the DEM (scene
 * cache key), [3] DEM upload, [4] acceleration-table build (synchronised), [5] mesh / environment / atmosphere uploads and
 * the mesh BVH, [6] allocation + clears of the per-pixel state, [7] enqueueing the G-buffer and certificate passes (their
 * device time is NOT in it: the call returns with them in flight).  [3] and [4] are 0 for a DEM the scene cache holds. */
#define F3D_SETUP_PHASES 8
int f3d_session_setup_ms(f3d_session *session, double *out, uint32_t count);
/* Memory / layout diagnostics of a session. */
int f3d_session_info(f3d_session *session, uint64_t *gpu_resource_bytes, uint64_t *minmax_pyramid_bytes,
                     uint64_t *peak_host_visible_bytes, uint32_t *rows, uint32_t *width);
/* Average device time in ms of the `count` most recent frame-kernel launches,
 * measured with hipEvents recorded on the session stream around every launch when
 * timing is enabled (bench.py's roofline leg).  enable: 1 start, 0 stop. */
int f3d_session_kernel_timing(f3d_session *session, int32_t enable, double *avg_ms, uint32_t *launches);
/* Sample lanes per pixel the frame kernel of this session runs with (1, 2, 4 or 8); 0 on a NULL session. */
uint32_t f3d_session_sample_lanes(f3d_session *session);
/* (ABI 5) Cost of the session's last fused frame by image row (out[rows], rows = the session's; 100 MHz ticks of wave time,
 * a tile's duration spread over its rows): what the strip driver balances row strips on.  Synchronises the stream. */
int f3d_session_row_costs(f3d_session *session, float *out, uint32_t rows, char *err, size_t errlen);
/* (ABI 5) The session's primary-ray certificates (f3d_cone.h): a DEVICE pointer to rows x width records {f32 bits of t_clear,
 * u32 level}, written by the G-buffer pass the session's creation enqueued on its stream; NULL when the camera's pixels are too
 * wide for certificates.  Valid while the session lives; the call waits for the session's stream, so the records are written
 * when it returns.  The PBR path tracer takes it as f3d_wf_scene.primary_start (same camera, image size and heightfield). */
const void *f3d_session_primary_start(f3d_session *session);
/* Device memory the library has freed is kept for its next allocation of the same size (F3D_DEVICE_POOL_MB, default
 * 1024, 0 = off): a camera path or a smoke sequence allocates the same buffers frame after frame.  This hands everything
 * that is waiting back to the driver -- call it before another allocator (torch, RCCL) sizes large buffers on the device:
 * what the pool holds is invisible to them. */
void f3d_device_pool_trim(void);
/* Rows of neighbour state a strip keeps above and below its own (4: the spatial pass reaches -3 .. +4 rows); the halo
 * blocks of f3d_session_halo and the extra rows of ext_reservoirs are this many rows. */
uint32_t f3d_halo_rows(void);

/* Diagnostics: out[0..15] = hashes of everything a frame launch of this session reads.  [0] camera, [1] light,
 * [2] terrain scalars, [3] mesh scalars, [4] other scalars, [5] leaf table, [6] band tables, [7] mesh vertices,
 * [8] mesh indices, [9] BVH nodes, [10] BVH triangles, [11] environment, [12] G-buffer, [13] reservoirs,
 * [14] accumulation + Welford, [15] frame-head records.  Synchronises the session.  count >= 16. */
int f3d_session_fingerprint(f3d_session *session, uint64_t *out, uint32_t count);

/* Diagnostics (only in builds with -DF3D_WAVE_TIMES; F3D_STATUS_VALUE otherwise): every frame-kernel workgroup
 * writes its {start, end} wall clock (100 MHz ticks) to device_buffer[2 * workgroup]; NULL switches it off. */
int f3d_session_debug_wave_times(f3d_session *session, void *device_buffer);

/* ---- post filter ---------------------------------------------- */
/* Edge-aware a-trous denoiser: replaces forge3d.denoise.atrous_denoise
 * (reference python/forge3d/denoise.py:18-127).  color / albedo / normal: H x W x 3 f32,
 * depth: H x W f32; albedo, normal, depth may be NULL.  `out` (H x W x 3 f32) is caller-owned.
 * iterations < 1 runs one pass, like the reference. */
int f3d_atrous_denoise(const float *color, const float *albedo, const float *normal, const float *depth,
                       uint32_t width, uint32_t height, int32_t iterations, float sigma_color, float sigma_albedo,
                       float sigma_normal, float sigma_depth, float *out, char *err, size_t errlen);

/* ---- AETHER atmosphere LUT baker (the offline half) ------------------------------------
 * Replaces bake_atmosphere_luts (reference src/core/atmosphere/bake.rs:1481-1666; cargo feature `atmosphere-bake`,
 * single-thread host code there): the tables of an f3d_aether_luts for any AtmosphereConfig, not only the five shipped
 * turbidity anchors.  Outputs are RGBA16F bit patterns in the reference's layouts (x fastest): transmittance
 * [height][mu]; single and accumulated scattering [height][nu][mu_sun][mu_view]; aerial [height][mu_view][distance];
 * order_deltas[scattering_orders] = mean |field| of every order (AtmosphereLuts::order_deltas). */
typedef struct f3d_aether_bake_config { /* AtmosphereConfig + LutDimensions, bake.rs:32-42,132-144 */
    float turbidity, ozone_du, mie_g, bottom_radius_m, top_radius_m, rayleigh_scale_height_m, mie_scale_height_m,
        max_aerial_distance_m, ground_albedo;
    uint32_t scattering_orders; /* 2..8 */
    uint32_t transmittance_mu, transmittance_height, scattering_mu_view, scattering_mu_sun, scattering_height, scattering_nu,
        aerial_distance, aerial_mu_view, aerial_height; /* 2..256 each */
} f3d_aether_bake_config;
int f3d_aether_bake(const f3d_aether_bake_config *config, uint16_t *transmittance, uint16_t *single_scattering,
                    uint16_t *accumulated_scattering, uint16_t *aerial, float *order_deltas, double *seconds, char *err,
                    size_t errlen);

/* ---- smoke volume ray-marcher (BASELINE.json configs[4]) --------------------------
 * Replaces SmokeVolume::raymarch_rgba / raymarch_projection_rgba (reference src/smoke/render.rs:7-178, bound to
 * Python as SmokeDomain.render_rgba / render_projection_rgba, src/smoke/py.rs:531-625).  Fields are the reference's
 * (x fastest, then y, then z: index = (z * ny + y) * nx + x, types.rs:359-361). */
typedef struct f3d_smoke_volume {
    const float *density, *temperature, *soot, *humidity, *emission, *age; /* emission = emission_rate, age = particle_age */
    uint32_t dims[3];    /* nx, ny, nz (each >= 2; nx*ny*nz <= 256^3 like the reference) */
    float voxel_size[3];
    float origin[3];
    uint32_t frame_index; /* jitter seed term (SmokeVolume::frame_index as u32) */
} f3d_smoke_volume;

typedef struct f3d_smoke_view {
    uint32_t width, height;
    int32_t projection;       /* 0: perspective camera (raymarch_rgba); 1: map-aligned parallel projection */
    float camera_pos[3], target[3], up[3], fovy_deg; /* perspective */
    float view_direction[3];  /* projection */
    float sun_direction[3];
} f3d_smoke_view;

typedef struct f3d_smoke_settings { /* SmokeRenderSettings, reference src/smoke/types.rs:227-269 */
    float density_scale, extinction, scattering, absorption, phase_g, step_size;
    uint32_t max_steps;
    int32_t self_shadow;
    uint32_t shadow_steps;
    float shadow_step_size, jitter_strength, exposure;
    float thin_color[3], dense_color[3];
    float soot_absorption, fire_glow;
} f3d_smoke_settings;

/* rgba: caller-owned height x width x 4 bytes (straight colour, alpha = 1 - transmittance).  kernel_seconds
 * (optional) receives the ray-march kernel's device time.  Errors carry the reference's message texts.
 * Each of the six fields, and rgba, may be a DEVICE pointer: such a field is read where it is and a device image is
 * left on the device (a resident smoke sequence: f3d_smoke_step on device fields -> f3d_smoke_render -> f3d_smoke_composite).
 * The three handle-less smoke entry points launch on the NULL stream, return when their kernels have finished (as in ABI 4)
 * and keep their scratch between calls (freed by f3d_device_pool_trim).  A resident sequence uses the handle below
 * (f3d_smoke_seq_*: own streams, asynchronous calls, scratch freed with the handle).  ABI-5 shim, for one revision: a
 * thread that has called f3d_smoke_set_stream gets round 5's behaviour for its handle-less calls -- they launch on the
 * stream it named and, when their results stay on the device and no time is asked for, return with their launches enqueued. */
int f3d_smoke_render(const f3d_smoke_volume *volume, const f3d_smoke_view *view, const f3d_smoke_settings *settings,
                     uint8_t *rgba, double *kernel_seconds, char *err, size_t errlen);
/* ABI-5 SHIM (superseded by f3d_smoke_seq_*; kept for one revision).
 * The stream (a hipStream_t; NULL = the null stream) on which the calling thread's next f3d_smoke_step / f3d_smoke_render /
 * f3d_smoke_composite calls enqueue.  With the solver on one stream and the marcher on another, step f + 1 runs beside
 * the march of frame f: the marcher reads the volume's fields only in its first kernels (it re-packs them), and
 * f3d_smoke_wait_fields_read makes `stream` wait for exactly that point of the calling thread's LAST f3d_smoke_render
 * (a no-op before the first).  Scratch is per entry point, not per stream: keep each entry point on one stream.
 * (No counterpart in the reference, whose solver and marcher are host loops.) */
void f3d_smoke_set_stream(void *stream);
int f3d_smoke_wait_fields_read(void *stream);

/* ---- smoke transport solver (the 120-frame sequence of BASELINE.json configs[4] needs its fields from somewhere) -----------
 * Replaces SmokeVolume::step / add_emitter (reference src/smoke/sim.rs:7-139, bound to Python as SmokeDomain.step,
 * src/smoke/py.rs:459-480; single-threaded host code there): `steps` steps of the solver on the device -- emitters,
 * forces (wind, buoyancy, procedural turbulence), semi-Lagrangian / MacCormack advection, diffusion, vorticity
 * confinement, Jacobi pressure projection, boundary damping, sub-grid density eddies, decay and ageing.  All fields are
 * caller-owned arrays, read and written in place ((nz, ny, nx) f32, velocity (nz, ny, nx, 3)): nine HOST arrays (uploaded and
 * read back by the call) or nine DEVICE arrays (the solver works on them where they are: nothing crosses the bus). */
typedef struct f3d_smoke_state {
    float *density, *temperature, *fuel, *soot, *humidity, *emission_rate, *particle_age, *velocity, *pressure;
    uint32_t dims[3];
    float voxel_size[3], origin[3];
    float sparse_threshold; /* SmokeDomainConfig::sparse_threshold (default 1e-5) */
    float time_seconds;     /* in/out */
    uint32_t frame_index;   /* in/out */
} f3d_smoke_state;
typedef struct f3d_smoke_step_settings { /* SmokeStepSettings, reference src/smoke/types.rs:142-180 */
    float dt, density_decay, temperature_decay, velocity_damping, diffusion, buoyancy, vorticity;
    uint32_t pressure_iterations;
    float turbulence_strength;
    uint32_t turbulence_seed;
    int32_t mac_cormack, mass_conservation, terrain_collision;
    float boundary_damping;
    float wind[3];
} f3d_smoke_step_settings;
typedef struct f3d_smoke_emitter { /* SmokeEmitter, reference src/smoke/types.rs:69-99 */
    float center[3], radius, density_rate, temperature_rate, fuel_rate, soot_rate, humidity_rate, emission_rate, velocity[3], start_time,
        end_time;
} f3d_smoke_emitter;
int f3d_smoke_step(f3d_smoke_state *state, const f3d_smoke_step_settings *settings, const f3d_smoke_emitter *emitters,
                   uint32_t emitter_count, uint32_t steps, double *device_seconds, char *err, size_t errlen);

/* ---- smoke over terrain: the per-pixel composites of BASELINE.json configs[4] ----
 * The reference builds each frame of its smoke sequence on the host with numpy and Pillow
 * (examples/california_cigar_smoke_demo.py, tested by tests/test_california_cigar_smoke_hybrid.py);
 * these are the same three operations as one device pass over RGBA8 images:
 *   F3D_COMPOSITE_ATMOSPHERIC  composite_atmospheric_smoke(base, layer)   :8527-8544 -- a ray-marched smoke layer
 *                              (f3d_smoke_render) over a path-traced terrain frame; the output alpha is 255
 *   F3D_COMPOSITE_SMOKE_MAPS   composite_main_smoke_maps(base = atmospheric, layer = physical or NULL,
 *                              base_alpha, layer_alpha) :3367-3380, alpha capped at max_alpha
 *   F3D_COMPOSITE_OVER         PIL.Image.alpha_composite(base, layer placed at offset) (:8721-8725, :3335-3349);
 *                              the layer may have its own size and is clipped to the base
 * base, layer and out may be host or device pointers (width * height * 4 bytes, rows tightly packed); out may
 * alias base.  kernel_seconds (may be NULL) receives the device time of the pass. */
#define F3D_COMPOSITE_ATMOSPHERIC 0u
#define F3D_COMPOSITE_SMOKE_MAPS 1u
#define F3D_COMPOSITE_OVER 2u
typedef struct f3d_composite_desc {
    uint32_t struct_size; /* sizeof(f3d_composite_desc) of the caller's header */
    uint32_t mode;
    uint32_t width, height;             /* base and output */
    uint32_t layer_width, layer_height; /* must equal width, height except in F3D_COMPOSITE_OVER */
    int32_t offset_x, offset_y;         /* F3D_COMPOSITE_OVER only, else 0 */
    const uint8_t *base;
    const uint8_t *layer; /* NULL only in F3D_COMPOSITE_SMOKE_MAPS (`physical_rgba is None`) */
    float base_alpha, layer_alpha; /* F3D_COMPOSITE_SMOKE_MAPS: atmospheric_alpha (0.42), physical_alpha (0.92) */
    uint32_t max_alpha;            /* F3D_COMPOSITE_SMOKE_MAPS: HYBRID_SMOKE_MAX_ALPHA (168) */
} f3d_composite_desc;
int f3d_smoke_composite(const f3d_composite_desc *desc, uint8_t *out_rgba, double *kernel_seconds, char *err, size_t errlen);

/* ---- a resident smoke sequence behind a handle (ABI 6) ----
 * BASELINE.json configs[4] is a 120-frame sequence: per frame a solver step, a march, a composite, all on state that stays
 * on the device.  The handle holds what such a sequence needs between calls -- which stream the solver runs on and which the
 * marcher (the caller's hipStream_t values, which must outlive the handle; NULL = the null stream; the same stream twice =
 * no overlap), the order between them, and the scratch of the three entry points:
 *   f3d_smoke_seq_step       runs on the solver's stream, behind the last render's READS of the fields (the marcher re-packs
 *                            the fields in its first kernels; the rest of the march runs beside the next step);
 *   f3d_smoke_seq_render     runs on the marcher's stream, behind the last step;
 *   f3d_smoke_seq_composite  runs on the marcher's stream (behind the render whose layer it reads).
 * Arguments are those of f3d_smoke_step / f3d_smoke_render / f3d_smoke_composite.  A call whose results stay on the device
 * and whose time pointer is NULL returns with its launches enqueued: order other streams behind the handle's streams
 * (f3d_smoke_seq_streams + an event) before reading its outputs; kernel errors then surface in a later call.
 * f3d_smoke_seq_destroy waits for both streams and gives the sequence's scratch back.  Two handles share nothing; one
 * handle is for one thread at a time.  The reference has no counterpart (its solver and marcher are single-threaded host
 * loops, src/smoke/sim.rs:270-317, src/smoke/render.rs:7-96). */
typedef struct f3d_smoke_seq f3d_smoke_seq;
typedef struct f3d_smoke_seq_stats_t {
    uint64_t scratch_bytes;               /* device memory the sequence holds between calls */
    uint32_t shadow_list_chunks;          /* capacity of the marcher's deferred self-shadow list in the last render, in chunks */
    uint32_t shadow_list_chunks_used;     /* chunks that render asked for (above the capacity: the excess was walked in the one-kernel form) */
    uint32_t shadow_list_slots_per_chunk; /* 52 bytes a slot */
    uint32_t reserved;
} f3d_smoke_seq_stats_t;
int f3d_smoke_seq_create(void *stream_solver, void *stream_march, f3d_smoke_seq **out, char *err, size_t errlen);
void f3d_smoke_seq_destroy(f3d_smoke_seq *seq);
int f3d_smoke_seq_streams(f3d_smoke_seq *seq, void **stream_solver, void **stream_march);
int f3d_smoke_seq_step(f3d_smoke_seq *seq, f3d_smoke_state *state, const f3d_smoke_step_settings *settings, const f3d_smoke_emitter *emitters,
                       uint32_t emitter_count, uint32_t steps, double *device_seconds, char *err, size_t errlen);
int f3d_smoke_seq_render(f3d_smoke_seq *seq, const f3d_smoke_volume *volume, const f3d_smoke_view *view, const f3d_smoke_settings *settings,
                         uint8_t *rgba, double *kernel_seconds, char *err, size_t errlen);
int f3d_smoke_seq_composite(f3d_smoke_seq *seq, const f3d_composite_desc *desc, uint8_t *out_rgba, double *kernel_seconds, char *err, size_t errlen);
int f3d_smoke_seq_stats(f3d_smoke_seq *seq, f3d_smoke_seq_stats_t *out, char *err, size_t errlen);

/* ---- AETHER acceptance reference: stochastic spectral transport (no LUT, black environment) ----
 * Replaces _forge3d.hybrid_render_aether_spectral_reference (reference src/py_functions/path_tracing/
 * aether_reference.rs:14-146 -> HybridPathTracer::render_aether_spectral_reference, hybrid_compute/
 * aether_reference.rs:203-560 -> WGSL main_aether_spectral_reference, shaders/atmosphere/
 * prometheus_spectral_reference.wgsl:423-480): delta-tracking paths at 11 wavelengths over the terrain tracer's own
 * camera rays, heightfield hits and sun visibility; the acceptance check of the LUT post pass (f3d_terrain_ref_desc.
 * atmosphere).  Same validation and messages as the reference (status F3D_STATUS_RENDER), plus: the DEM must lie
 * inside the top of the atmosphere (a terrain hit then always precedes the top-of-atmosphere exit, which the
 * reference's shadow test assumes).  At most 8 000 000 wavelength paths (width * height * spp * 11). */
typedef struct f3d_aether_ref_desc { /* AetherSpectralReferenceDesc, aether_reference.rs:14-43 */
    uint32_t struct_size;            /* sizeof(f3d_aether_ref_desc) of the caller's header */
    uint32_t dem_width, dem_height;
    const float *heights;            /* row-major (dem_height, dem_width) */
    float spacing_x, spacing_z, exaggeration;
    float cam_origin[3], cam_look_at[3], cam_up[3], fov_y_deg;
    float sun_azimuth_deg, sun_elevation_deg, sun_intensity;
    float turbidity, ozone_du, mie_g, ground_albedo;
    uint32_t width, height, seed, spp;
    int32_t enabled;                 /* 0: explicit black (all outputs zero, converged) */
    float variance_threshold;
} f3d_aether_ref_desc;
typedef struct f3d_aether_ref_out { /* AetherSpectralReferenceOutput, aether_reference.rs:46-60 */
    float *mean_xyz;   /* width * height * 3: unclipped per-pixel mean CIE XYZ */
    float *linear_rgb; /* width * height * 3: max(signed linear RGB of the mean, 0) */
    float variance;    /* max over pixels of the estimated variance of the sample-mean CIE Y */
    int32_t converged;
    uint64_t terrain_primary_hits; /* camera samples whose primary ray hit the terrain */
    uint64_t gpu_resource_bytes;
    double kernel_seconds;         /* device time of the three launches */
} f3d_aether_ref_out;
int f3d_aether_reference_render(const f3d_aether_ref_desc *desc, f3d_aether_ref_out *out, char *err, size_t errlen);

/* ---- test hooks (KATs restated from the reference's Rust unit tests) ----------- */
/* build_minmax_mips on the GPU (reference terrain_heightfield.rs:132-202); output in
 * the reference's layout: levels back to back, finest first, each (ph, pw, 2) f32;
 * dims_out (pw, ph) pairs.  levels_out may be NULL to query level count / size. */
int f3d_build_minmax_mips(const float *heights, uint32_t width, uint32_t height, float *levels_out,
                          uint32_t *dims_out, uint32_t max_levels, uint64_t *total_floats, char *err,
                          size_t errlen);
/* terrain_trace over a ray batch (reference test seam
 * main_helios_production_terrain_trace_proof, terrain_heightfield.rs:1646-1671).
 * rays: n x 8 f32 (origin xyz, tmin, direction xyz, tmax). */
int f3d_terrain_trace_batch(const float *heights, uint32_t width, uint32_t height, float origin_x,
                            float origin_z, float spacing_x, float spacing_z, float exaggeration,
                            float inv_two_r_prime, uint32_t curvature_enabled, const float *rays,
                            uint32_t n, int32_t any_hit, int32_t apply_curvature, uint32_t *out_hit,
                            float *out_t, float *out_normal, char *err, size_t errlen);
/* geo::refraction::effective_radius_m (reference src/geo/refraction.rs:137-148). */
int f3d_effective_radius_m(int32_t earth_model, double latitude_deg, double sphere_radius_m,
                           int32_t refraction_model, double pressure_mbar, double temperature_c,
                           double refraction_k, double azimuth_deg, double *radius_out, char *err,
                           size_t errlen);

/* Scene cache: the acceleration tables of the most recently rendered DEMs (default 2 per process) stay on the
 * device and are shared by every session / one-shot call that renders the same heights, dims and exaggeration --
 * a camera path does not rebuild them per frame.  0 entries switches the cache off and frees it. */
void f3d_scene_cache_limit(uint32_t entries);
uint32_t f3d_scene_cache_entries(void);

int f3d_device_count(void);
const char *f3d_device_name(int32_t device); /* gcnArchName, "" when unavailable */
const char *f3d_version(void);
/* F3D_ABI_VERSION the library was built with: a binding checks it once after loading. */
uint32_t f3d_abi_version(void);
/* First 16 hex digits of the SHA-256 of the sources + compiler flags the library was built from ("unknown" for a
 * build that did not go through __graft_entry__.build_hip). */
const char *f3d_source_digest(void);
/* Diagnostics (f3d_devmem.h): pattern 0..255 = every device buffer the library allocates from now on lies between two
 * 256 KiB guard regions and buffer and guards are filled with that byte; negative = off.  No result may depend on the
 * pattern: a dependence means a kernel reads memory nobody wrote (uninitialised, or beyond a buffer).  The environment
 * variable F3D_POISON=<0..255> starts a process in this mode. */
void f3d_debug_poison(int32_t pattern);

#ifdef __cplusplus
}
#endif
#endif /* F3D_TERRAIN_PT_H */
