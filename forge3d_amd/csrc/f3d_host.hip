// forge3d_amd/csrc/f3d_host.hip -- host side of libf3dhip.so (C ABI in include/f3d_terrain_pt.h).
//
// Mirrors, for the one hot path, what the reference does in Rust:
//   validate_desc                       src/path_tracing/hybrid_compute/render_terrain.rs:474-557
//   TerrainPtScene::new / pyramid       .../terrain_heightfield.rs:132-202, :390-494
//   EarthCurvatureUniforms::new         .../terrain_heightfield.rs:52-84 + src/geo/refraction.rs
//   HybridPathTracer::render_terrain_reference   .../render_terrain.rs:563-1434
// with a different runtime: no per-frame host synchronisation (frames are enqueued
// back to back on one HIP stream; the host reads ONE 16-byte record per 32-frame window
// instead of 8 B/pixel), packed 16-byte reservoirs, and acceleration tables built on the
// GPU.  There is no CPU fallback: every entry point that computes needs a HIP device.
#include <hip/hip_runtime.h>

#include <atomic>
#include <deque>
#include <exception>
#include <memory>
#include <map>
#include <mutex>
#include <new>
#include <unordered_map>

#include <unistd.h>

#include "f3d_devmem.h"
#include "f3d_launch.h"
#include "f3d_lbvh.h"
#include "f3d_meshgrid.h"
#include "f3d_setup.h"
#include "f3d_tables.h"

using namespace f3d;

namespace {

void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) fail(F3D_STATUS_DEVICE, "HIP failure in %s: %s", what, hipGetErrorString(e));
}

// Binds a device for the duration of a C-ABI call and gives the caller's current device back afterwards:
// entry points may be called from any thread and with any device current.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (device < 0 || hipGetDevice(&prev) != hipSuccess) return;
        if (prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// Runs `body` behind the C ABI: no C++ exception crosses the extern "C" boundary.
template <class Body>
int c_abi(char *err, size_t errlen, Body &&body) {
    if (err && errlen) err[0] = 0;
    try {
        body();
    } catch (const Failure &f) {
        return report(f, err, errlen);
    } catch (const std::exception &e) {  // std::bad_alloc from the host-side builders, ...
        if (err && errlen) snprintf(err, errlen, "host failure: %s", e.what());
        return F3D_STATUS_DEVICE;
    } catch (...) {
        if (err && errlen) snprintf(err, errlen, "unknown host failure");
        return F3D_STATUS_DEVICE;
    }
    return F3D_STATUS_OK;
}
f3d_session &checked(f3d_session *s) {
    if (!s) fail(F3D_STATUS_VALUE, "null session handle");
    return *s;
}

}  // namespace

#include "f3d_host_mem.h"  // poison mode, device memory pool, ledger, acceleration tables, scene cache

// ---------------------------------------------------------------------------------------
// session
// ---------------------------------------------------------------------------------------
struct f3d_session {
    double setup_ms[kSetupPhases] = {};
    Ledger mem;
    hipStream_t stream = nullptr;
    int device = 0;
    FrameParams params{};
    std::shared_ptr<CachedTables> scene;  // shared, immutable acceleration tables (scene cache)
    std::shared_ptr<CachedMesh> mesh;     // shared, immutable device copy of the mesh and its BVH (mesh cache)
    TerrainTables tables;
    uint32_t width = 0, height = 0, row_begin = 0, row_end = 0, rows = 0;
    PackedReservoir *res[2] = {nullptr, nullptr};
    float4 *gbuffer_n = nullptr;
    float *depth = nullptr;
    uint32_t *stats = nullptr;
    uint32_t *host_stats = nullptr;  // pinned
    // peer halos (include/f3d_terrain_pt.h), a 256-byte counter block the neighbours map: [0] frames merged (they poll it),
    // [1] wait time-outs of this strip, [2] link-probe nonce, [4..5] link-probe words read, [6..7] checksums of the probe blocks
    // pulled, [8..11] clock ticks spent waiting for the strip above / below (64 bit each), [12] pulls, [13] longest wait (ticks)
    uint32_t *halo_flags = nullptr;
    uint32_t halo_frames_published = 0;   // frames this session has enqueued through enqueue_batch_strip: counters only rise
    unsigned long long halo_timeout_ticks = 0;  // of the wall clock; F3D_HALO_TIMEOUT_MS, default 20 s
    double wall_clock_khz = 100000.0;
    struct PeerLink {
        void *opened[3] = {nullptr, nullptr, nullptr};  // hipIpcOpenMemHandle results (closed with the session)
        const PackedReservoir *res[2] = {nullptr, nullptr};
        const uint32_t *flags = nullptr;
        uint32_t rows = 0;
        bool connected = false;
    } peer[2];  // 0 = the strip above, 1 = the strip below
    bool owns_reservoirs = false;
    uint8_t *d_rgba = nullptr;
    float *d_albedo = nullptr, *d_normal = nullptr;
    int variant = 0;
    AetherDev aether{};  // enabled = 0 without desc.atmosphere
    bool require_valid_reservoirs = false;
    uint64_t budget = 0;
    // band pipelining (f3d_session_opts.bands): horizontal bands of the strip, their streams and events
    struct Band {
        uint32_t begin = 0, end = 0;   // image rows
        hipStream_t stream = nullptr;  // == the session stream when the strip is one band
        bool edge = true;              // contains rows a neighbouring strip needs as halo (frame part 1)
        hipEvent_t done[2] = {nullptr, nullptr};  // frame f of this band has finished: done[f & 1]
        int64_t last = -1;             // last frame enqueued
        bool unjoined = false;         // the session stream has not been ordered after `last` yet
    };
    std::vector<Band> bands;
    std::vector<hipStream_t> band_streams;
    hipEvent_t fork = nullptr;  // position of the session stream when a batch of band launches began
    // longest-first dispatch (f3d_kernels.hip k_tile_order): one-band sessions with the default tile map
    uint32_t *tile_cost = nullptr, *tile_order = nullptr;
    int64_t cost_frame = -1, order_frame = -1;  // newest frame whose wave durations are in tile_cost / went into tile_order
    // frames in flight (f3d_kernels.hip k_trace / k_merge): record buffer for `fd_frames` frames; 0 = the fused path
    uint32_t fd_frames = 0;
    // wavefront form of the trace batches (f3d_kernels.hip k_wf_primary / k_wf_occl): occlusion rays through queues in HBM
    bool wavefront = false;
    uint32_t wf_quorum = 0;
    int64_t trace_first = -1;
    uint32_t trace_count = 0;
    // per-launch timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    ~f3d_session() {
        for (auto &e : events) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        for (auto &b : bands)
            for (hipEvent_t e : b.done)
                if (e) (void)hipEventDestroy(e);
        if (fork) (void)hipEventDestroy(fork);
        for (hipStream_t st : band_streams) (void)hipStreamDestroy(st);
        if (host_stats) (void)hipHostFree(host_stats);
        for (auto &link : peer)
            for (void *base : link.opened)
                if (base) (void)hipIpcCloseMemHandle(base);
        if (halo_flags) (void)hipFree(halo_flags);
        mem.release();
    }
};

namespace {

void plan_bands(f3d_session &s, uint32_t want, uint32_t want_streams);

// AETHER LUT payload -> device tables (reference AetherPostPass::new, aether_post.rs:42-100: config.validate,
// validate_luts, three RGBA16F texture uploads).  The tables are decoded to float4 once here.
void upload_aether(f3d_session &s, const f3d_aether_luts &L, const f3d_terrain_ref_desc &d) {
    auto bad = [](const char *why) { fail(F3D_STATUS_RENDER, "invalid AETHER PT settings: %s", why); };
    const float scalars[9] = {L.turbidity, L.ozone_du, L.mie_g, L.bottom_radius_m, L.top_radius_m, L.rayleigh_scale_height_m,
                              L.mie_scale_height_m, L.max_aerial_distance_m, L.ground_albedo};
    for (float v : scalars)
        if (!std::isfinite(v)) bad("all scalar parameters must be finite");  // AtmosphereConfig::validate, bake.rs:164-229
    if (!(L.turbidity >= 1.0f && L.turbidity <= 10.0f)) bad("turbidity must be in [1, 10]");
    if (!(L.ozone_du >= 0.0f && L.ozone_du <= 600.0f)) bad("ozone must be in [0, 600] DU");
    if (!(L.mie_g >= 0.0f && L.mie_g <= 0.99f)) bad("mie_g must be in [0, 0.99]");
    if (L.bottom_radius_m <= 0.0f || L.top_radius_m <= L.bottom_radius_m) bad("top radius must exceed a positive bottom radius");
    if (L.rayleigh_scale_height_m <= 0.0f || L.mie_scale_height_m <= 0.0f || L.max_aerial_distance_m <= 0.0f)
        bad("scale heights and aerial distance must be positive");
    if (!(L.ground_albedo >= 0.0f && L.ground_albedo <= 1.0f)) bad("ground albedo must be in [0, 1]");
    if (L.scattering_orders < 2u || L.scattering_orders > 8u) bad("scattering_orders must be in [2, 8]");
    const uint32_t dims[9] = {L.transmittance_mu, L.transmittance_height, L.scattering_mu_view, L.scattering_mu_sun,
                              L.scattering_height, L.scattering_nu, L.aerial_distance, L.aerial_mu_view, L.aerial_height};
    for (uint32_t v : dims)
        if (v < 2u || v > 4096u) fail(F3D_STATUS_RENDER, "PROMETHEUS AETHER LUT dimensions do not match metadata");
    if (!L.transmittance || !L.accumulated_scattering || !L.aerial)
        fail(F3D_STATUS_RENDER, "PROMETHEUS AETHER requires rgba16float accumulated-scattering LUTs");
    struct Table {
        const uint16_t *src;
        size_t texels;
        float max_value;
        bool alpha_only;
        const char *label;
    };
    const Table tables[3] = {
        {L.transmittance, (size_t)L.transmittance_mu * L.transmittance_height, 1.0f, false, "transmittance"},
        {L.accumulated_scattering, (size_t)L.scattering_mu_view * L.scattering_mu_sun * L.scattering_height * L.scattering_nu,
         65504.0f, false, "accumulated-scattering"},
        {L.aerial, (size_t)L.aerial_distance * L.aerial_mu_view * L.aerial_height, 1.0f, true, "aerial-perspective"}};
    const float4 *dev[3];
    for (int t = 0; t < 3; t++) {
        std::vector<float> f(tables[t].texels * 4);
        for (size_t i = 0; i < f.size(); i++) {
            const float v = half_value(tables[t].src[i]);
            // validate_lut_payload / the aerial semantics check, runtime.rs:124-150, :229-240
            if (!std::isfinite(v) || v < 0.0f || v > tables[t].max_value)
                fail(F3D_STATUS_RENDER, "runtime %s payload component %zu must be finite and in [0, %g], got %g", tables[t].label, i,
                     (double)tables[t].max_value, (double)v);
            if (tables[t].alpha_only && (i & 3u) != 3u && v != 0.0f)
                fail(F3D_STATUS_RENDER,
                     "runtime aerial-perspective payload must store zero RGB and unit-bounded transmittance alpha");
            f[i] = v;
        }
        float4 *p = (float4 *)s.mem.alloc(f.size() * sizeof(float), "AETHER LUT");
        hip_check(hipMemcpy(p, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice), "AETHER LUT upload");
        dev[t] = p;
    }
    AetherDev &A = s.aether;
    A.transmittance = dev[0];
    A.scattering = dev[1];
    A.aerial = dev[2];
    A.t_mu = L.transmittance_mu;
    A.t_h = L.transmittance_height;
    A.s_view = L.scattering_mu_view;
    A.s_sun = L.scattering_mu_sun;
    A.s_h = L.scattering_height;
    A.s_nu = L.scattering_nu;
    A.a_dist = L.aerial_distance;
    A.a_mu = L.aerial_mu_view;
    A.a_h = L.aerial_height;
    A.bottom_radius = L.bottom_radius_m;
    A.top_radius = L.top_radius_m;
    A.max_aerial_distance = L.max_aerial_distance_m;
    A.ozone_du = L.ozone_du;
    A.turbidity = L.turbidity;
    A.sun_intensity = aether_clamp_scale(f_clamp(d.sun_intensity, 0.0f, 65504.0f));
    A.exposure = aether_clamp_scale(f_clamp(d.exposure, 0.0f, 65504.0f));
    A.enabled = 1u;
}

// A caller compiled against another revision of the header passes structs of another size: refuse them before a
// single member is read (include/f3d_terrain_pt.h F3D_ABI_VERSION).
void check_abi(const f3d_terrain_ref_desc *d, const f3d_session_opts *opts) {
    char msg[256];
    if (d && d->struct_size != sizeof(f3d_terrain_ref_desc)) {
        snprintf(msg, sizeof msg, "f3d_terrain_ref_desc.struct_size is %u, this library (ABI version %u) expects %zu: the caller "
                 "was built against another revision of f3d_terrain_pt.h", d->struct_size, F3D_ABI_VERSION, sizeof(f3d_terrain_ref_desc));
        fail(F3D_STATUS_VALUE, msg);
    }
    if (opts && opts->struct_size != sizeof(f3d_session_opts)) {
        snprintf(msg, sizeof msg, "f3d_session_opts.struct_size is %u, this library (ABI version %u) expects %zu: the caller "
                 "was built against another revision of f3d_terrain_pt.h", opts->struct_size, F3D_ABI_VERSION, sizeof(f3d_session_opts));
        fail(F3D_STATUS_VALUE, msg);
    }
}

void session_init(f3d_session &s, const f3d_terrain_ref_desc &d, const f3d_session_opts *opts) {
    SetupClock clock(s.setup_ms);
    check_abi(&d, opts);
    validate_desc(d);
    // every DEM sample is looked at ONCE: finiteness (validate_scene) and the scene cache's key come out of one pass
    DemFingerprint fp;
    const bool fp_known = d.heights && d.dem_width >= 2 && d.dem_height >= 2 && d.dem_width <= 8193 && d.dem_height <= 8193;
    if (fp_known) fp = dem_fingerprint(d.heights, (size_t)d.dem_width * d.dem_height);
    validate_scene(d, fp_known ? (fp.finite ? 1 : 0) : -1);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        fail(F3D_STATUS_DEVICE, "no HIP device available: libf3dhip has no CPU fallback");
    s.device = (opts && opts->device >= 0) ? opts->device : -1;
    if (s.device >= 0) hip_check(hipSetDevice(s.device), "hipSetDevice");
    else hip_check(hipGetDevice(&s.device), "hipGetDevice");
    s.stream = opts ? (hipStream_t)opts->stream : nullptr;
    s.variant = opts ? opts->kernel_variant : 0;
    s.budget = (opts && opts->memory_budget_bytes) ? opts->memory_budget_bytes : (512ull << 20);
    s.width = d.width;
    s.height = d.height;
    s.row_begin = opts ? opts->row_begin : 0u;
    s.row_end = (opts && opts->row_end) ? opts->row_end : d.height;
    if (s.row_begin >= s.row_end || s.row_end > d.height)
        fail(F3D_STATUS_VALUE, "invalid row strip [%u, %u) for image height %u", s.row_begin, s.row_end, d.height);
    s.rows = s.row_end - s.row_begin;

    FrameParams &P = s.params;
    s.require_valid_reservoirs = fill_uniforms(d, P);  // curvature, camera, lighting
    clock.lap(kSetupValidate);

    // DEM upload + GPU table build (reference: CPU build + per-level write_texture), or the cached tables of this DEM
    bool was_cached = false;
    g_setup_ms = s.setup_ms;
    try {
        s.scene = acquire_tables(s.device, d.heights, d.dem_width, d.dem_height, d.exaggeration, s.stream, &was_cached, &fp);
    } catch (...) {
        g_setup_ms = nullptr;
        throw;
    }
    g_setup_ms = nullptr;
    clock.last = std::chrono::steady_clock::now();  // (acquire_tables lapped its own phases)
    s.tables = s.scene->tables;
    s.mem.device_bytes += s.scene->mem.device_bytes;  // shared, but part of this render's working set
    apply_layout(s.tables.layout, P.terrain);
    P.terrain.leaves = s.tables.leaves;
    P.terrain.nodes = s.tables.nodes;
    P.terrain.bands = s.tables.bands;
    P.terrain.mesh_bands = s.tables.bands;  // (no mesh grid: the fused march tests the terrain's band twice, f3d_scene.h)
#if !defined(F3D_NO_IBL_STOP)  // A/B builds (tools/build_variant.sh)
    // Opt-in (F3D_IBL_HORIZON=1): measured on MI355X, the table costs 12.9 ms to build for the 2048^2 headline DEM and
    // saves 0.03 ms per 8-spp 1080p frame (+1.2 %) -- its blocks' lowest points see higher horizons than the hit points
    // themselves; the per-pixel form of round 2's branch saved 0.13 ms per frame for 26 ms per render.  Neither pays
    // for itself below several hundred frames, so sessions do not build it unless asked (profiles/README.md).
    if (const char *hz_env = getenv("F3D_IBL_HORIZON"); hz_env && hz_env[0] == '1') {
        // far-horizon table of this DEM at this spacing: from the scene cache, or built now (one quadtree walk per block)
        std::lock_guard<std::mutex> lock(g_scene_mutex);
        const CachedTables::Horizon *have = nullptr;
        for (const auto &hz : s.scene->horizons)
            if (hz.spacing_x == P.terrain.spacing_x && hz.spacing_z == P.terrain.spacing_z) have = &hz;
        if (!have) {
            CachedTables::Horizon hz{};
            hz.spacing_x = P.terrain.spacing_x;
            hz.spacing_z = P.terrain.spacing_z;
            horizon_table_dims(P.terrain.cell_w, P.terrain.cell_h, &hz.level, &hz.bx, &hz.bz);
            const size_t bytes = (size_t)hz.bx * hz.bz * kIblSectors * sizeof(float);
            hz.table = (float *)s.scene->mem.alloc(bytes, "far-horizon table");
            hip_check(launch_horizon_build(P.terrain, hz.table, s.stream), "far-horizon table build");
            hip_check(hipStreamSynchronize(s.stream), "far-horizon table build");  // other sessions may use it from their streams
            s.mem.device_bytes += bytes;
            s.scene->horizons.push_back(hz);
            have = &s.scene->horizons.back();
        }
        P.terrain.horizon = have->table;
        P.terrain.horizon_level = have->level;
        P.terrain.horizon_bx = have->bx;
    }
#endif

    // environment map as rgb+pad texels (the reference uploads RGBA32F, terrain_heightfield.rs:443-482)
    if (d.env_map) {
        const size_t n = (size_t)d.env_width * d.env_height;
        const std::vector<float> rgba = pad_rgb_to_rgba(d.env_map, n, 1.0f);
        float4 *tex = (float4 *)s.mem.alloc(n * sizeof(float4), "env map");
        hip_check(hipMemcpy(tex, rgba.data(), n * sizeof(float4), hipMemcpyHostToDevice), "env upload");
        P.env.texels = tex;
        P.env.width = d.env_width;
        P.env.height = d.env_height;
    }
    // mesh (HybridScene::mesh_only + upload, src/sdf/hybrid.rs:285-366: vec4-padded vertices) with its acceleration structure
    // (reference: accel::build_bvh on the CPU, render_terrain.rs:597-627 -- which its kernel then never reads; here the rays
    // actually walk it, f3d_bvh.h).  Immutable once built, so -- like the DEM tables -- shared through a small per-process
    // cache (acquire_mesh, f3d_host_mem.h): the second session of a mesh (a strip job's probe and then its strip, a camera
    // path) pays neither the 80 ms host build of 600 000 triangles nor the upload.
    if (d.mesh_vertices) {
        // builder: 0 / 1 the binned-SAH host build (better trees, the default); 2 the GPU linear BVH (f3d_lbvh.hip:
        // ~1 ms for 600 000 triangles instead of 80 ms -- meshes that change every frame)
        uint32_t builder = opts ? opts->mesh_builder : 0u;
        if (builder == 0u) {
            const char *env = getenv("F3D_MESH_BVH");
            builder = (env && strcmp(env, "lbvh") == 0) ? 2u : ((env && strcmp(env, "binary") == 0) ? 3u : 1u);
        }
        if (builder < 1u || builder > 3u)
            fail(F3D_STATUS_VALUE, "mesh_builder must be 0 (automatic), 1 (host SAH, walked 4 wide), 2 (GPU LBVH) or 3 (host SAH, binary walk), got %u", builder);
        s.mesh = acquire_mesh(s.device, d.mesh_vertices, d.mesh_vertex_count, d.mesh_indices, d.mesh_index_count, builder, s.stream);
        s.mem.device_bytes += s.mesh->mem.device_bytes;  // shared, but part of this render's working set
        P.mesh = s.mesh->dev;
#if defined(F3D_MESH_FUSED)  // A/B build (f3d_shade.h occluded: measured slower, not in the shipped library)
        // ... and as a second band of the terrain's pyramid for the occlusion rays' march (f3d_meshgrid.h; F3D_MESH_GRID=0: the tree
        // walk for every ray).  Depends on the mesh AND on where the DEM's cells lie: kept with the cached mesh per grid geometry.
        if (const char *env = getenv("F3D_MESH_GRID"); !(env && env[0] == '0')) {
            const CachedMesh::Grid *grid = acquire_mesh_grid(*s.mesh, s.tables.layout, P.terrain, d.mesh_vertices, d.mesh_vertex_count, d.mesh_indices,
                                                             d.mesh_index_count);
            if (grid && grid->bands) {
                P.terrain.mesh_bands = grid->bands;
                P.terrain.mesh_cell_start = grid->cell_start;
                P.terrain.mesh_cell_tris = grid->tris;
                P.terrain.mesh_top = grid->top;
                s.mem.device_bytes += grid->bytes;
            }
        }
#endif
    }
    if (d.atmosphere) upload_aether(s, *d.atmosphere, d);
    clock.lap(kSetupScene);
    P.row_begin = s.row_begin;
    P.row_end = s.row_end;
    P.band_begin = s.row_begin;
    P.band_end = s.row_end;
    // variant = kernel variant + 1000 * tile map; map 0 = default (tile rows dealt round-robin
    // to the XCDs: 1.77x faster than contiguous bands on the headline scene, whose sky bands
    // left whole XCDs idle -- profiles/README.md)
    P.tile_map = (s.variant / 1000) % 10 ? (uint32_t)((s.variant / 1000) % 10) : 2u;
    // + 10000 * leaf quorum (f3d_trace.h "leaf gating"); 0 = default
    P.terrain.leaf_quorum = (s.variant / 10000) % 100 ? (uint32_t)((s.variant / 10000) % 100) : kDefaultLeafQuorum;
    // + 10000000 * ray-sharing threshold (f3d_march.h: deal the IBL rays when at most this many lanes of a wave
    // still march; 0 = default 16, 64 = share from the first step -- test coverage of the dealing code)
    P.terrain.share_below = (uint32_t)((s.variant / 10000000) % 100);
    // + 1000000 * sample lanes per pixel (f3d_kernels.hip frame_lanes): 1, 2, 4, 8; 0 = automatic.
    // A wave of the 1-lane kernel lasts spp x 3 traversals whatever the image size, so small images
    // and thin multi-GPU strips are latency-bound and even a 1080p frame ends in a ~1 ms tail of
    // half-empty SIMDs; sample lanes trade pixels per wave for shorter waves.
    {
        uint32_t lanes = (uint32_t)((s.variant / 1000000) % 10);
        if (lanes == 0u && s.variant % 1000 != 0) lanes = 1u;  // register-budget A/B kernels exist for 1 lane only
        if (lanes == 0u) {
            // measured (Msamples/s for 1 / 2 / 4 / 8 lanes): 1080p 3881 / 4860 / 5255 / 5261; 3840x2160
            // 5612 / 5738 / 5820 / 5458; 4096^2 with a 600k-triangle mesh 2271 / - / 2474 / 2440; an eighth of
            // a 1080p frame only scales with 8.  So: 4 lanes when spp allows, 8 for small strips.
            constexpr uint64_t kSmallStripWaves = 98304;  // 16 x (256 CUs x 4 SIMDs x 6 waves)
            lanes = P.spp >= 4u ? 4u : (P.spp >= 2u ? 2u : 1u);
            if (lanes == 4u && P.spp >= 8u && ((uint64_t)s.rows * s.width * 4u + 63u) / 64u < kSmallStripWaves) lanes = 8u;
        }
        if (lanes != 1u && lanes != 2u && lanes != 4u && lanes != 8u)
            fail(F3D_STATUS_VALUE, "kernel_variant: sample lanes must be 1, 2, 4 or 8 (got %u)", lanes);
        P.sample_lanes = lanes;
    }

    // per-pixel state
    const size_t px = (size_t)s.rows * s.width;
    const size_t res_n = (size_t)(s.rows + 2 * kHaloRows) * s.width;
    {
        // memory-budget gate, render_terrain.rs:875-888 -- evaluated on the planned working
        // set BEFORE the large allocations are made (the reference allocates, then checks).
        const uint64_t planned = s.mem.device_bytes + 2 * (uint64_t)res_n * sizeof(PackedReservoir) +
                                 (uint64_t)px * (sizeof(float4) + sizeof(float) + sizeof(float4) + sizeof(float) + 4 +
                                                 3 * sizeof(float) + 3 * sizeof(float) +
                                                 (P.sample_lanes > 1u ? sizeof(uint2) : 0) +
                                                 sizeof(uint2) + sizeof(float2) /* ray certificates (f3d_cone.h) */ + 1 /* tile costs and order */) + 16;
        if (planned > s.budget)
            fail(F3D_STATUS_RENDER,
                 "terrain PT exceeds the memory budget before rendering: tracked total %llu (host-visible %llu) > "
                 "limit %llu",
                 (unsigned long long)planned, (unsigned long long)s.mem.host_visible_peak,
                 (unsigned long long)s.budget);
        // frames in flight: as many as were asked for and fit the budget; needs one band.  same_sun: the two sun
        // directions the frame head chooses between (wi, normalize(wi)) are the same bits -- nothing to predict
        uint32_t want = opts ? opts->frames_in_flight : 0u;
        if (want == F3D_FRAMES_IN_FLIGHT_AUTO) {
            // images whose frame kernel cannot fill the chip twice are bound by the latency of a wave's ray chain
            // however many lanes share a pixel: 256 x 256 at 8 spp 1 174 -> 2 177 Msamples/s with 16 frames in flight,
            // 512 x 512 at 16 spp 3 393 -> 3 596, 1024 x 768 at 8 spp 4 627 -> 4 512 (tools/fd_probe3.py)
            const uint64_t waves = ((uint64_t)px * P.sample_lanes + 63u) / 64u;
            want = waves < 2u * 6144u ? 16u : 0u;
        }
        const bool same_sun = f_bits(P.light.wi.x) == f_bits(P.light.wi_reuse.x) && f_bits(P.light.wi.y) == f_bits(P.light.wi_reuse.y) &&
                              f_bits(P.light.wi.z) == f_bits(P.light.wi_reuse.z);
        P.same_sun = same_sun ? 1u : 0u;
        // Wavefront trace (terrain-only scenes: the mesh walk stays in the fused kernels): F3D_WAVEFRONT=1 asks for it,
        // =0 forbids it; it needs frames in flight (F3D_WF_FRAMES, default 2) and 84 instead of 32 bytes per sample in flight.
        const char *wf_env = getenv("F3D_WAVEFRONT");
        const bool wf_want = wf_env && wf_env[0] == '1' && P.mesh.traversal_mode != 0u;
        if (wf_want && want < 2u) {
            const char *n = getenv("F3D_WF_FRAMES");
            want = n ? (uint32_t)std::max(2, atoi(n)) : 2u;
        }
        if (want >= 2u && (!opts || opts->bands <= 1u)) {
            // what frames in flight allocate besides the records: re-trace list and counters, head records, tile costs
            const uint64_t fd_fixed = (uint64_t)px * sizeof(uint32_t) + (P.sample_lanes > 1u ? 0u : (uint64_t)px * sizeof(uint2)) + (1u << 20);
            const uint64_t room = s.budget > planned + fd_fixed ? s.budget - planned - fd_fixed : 0u;
            const uint64_t rec_bytes = 2u * sizeof(float4), ray_bytes = 2u * sizeof(float4) + sizeof(float4) + sizeof(float);
            uint64_t per_frame = (uint64_t)px * P.spp * (rec_bytes + (wf_want ? ray_bytes + ray_bytes / 4u : 0u));  // (+ a quarter: regions of partly filled edge tiles and rounds)
            uint64_t fit = per_frame ? room / per_frame : 0u;
            s.wavefront = wf_want && fit >= 2u;
            if (wf_want && !s.wavefront) {  // the queues do not fit the budget: plain frames in flight, if those do
                per_frame = (uint64_t)px * P.spp * rec_bytes;
                fit = per_frame ? room / per_frame : 0u;
                if (opts && opts->frames_in_flight < 2u) fit = 0u;  // (they were only asked for as part of the wavefront form)
            }
            s.fd_frames = (uint32_t)std::min<uint64_t>(want, fit);
            if (s.fd_frames < 2u) {
                s.fd_frames = 0u;
                s.wavefront = false;
            }
        }
    }
    if (s.fd_frames) {
        P.trace = (float4 *)s.mem.alloc((size_t)s.fd_frames * P.spp * px * 2u * sizeof(float4), "frames-in-flight records");
        P.trace_first = 0u;
        // touch the buffer now: the first write to fresh device memory pays for its page mappings (measured: the first
        // batch of a strip 0.42 ms per frame against 0.33 ms once the pages exist)
        hip_check(hipMemsetAsync(P.trace, 0, (size_t)s.fd_frames * P.spp * px * 2u * sizeof(float4), s.stream), "record buffer touch");
        P.fix_list = (uint32_t *)s.mem.alloc(px * sizeof(uint32_t), "retrace list");
        P.fix_count = (uint32_t *)s.mem.alloc(4 * sizeof(uint32_t), "retrace counters");
        hip_check(hipMemsetAsync(P.fix_count, 0, 4 * sizeof(uint32_t), s.stream), "retrace counters clear");
        if (s.wavefront) {
            // one region of 64 slots per (frame in flight, tile, round of the tile's wave): f3d_scene.h WfQueues
            P.band_begin = s.row_begin;
            P.band_end = s.row_end;
            const uint32_t lanes = P.sample_lanes ? P.sample_lanes : 1u;
            P.wf.regions_per_frame = frame_tile_count(P, nullptr) * ((P.spp + lanes - 1u) / lanes);
            const size_t regions = (size_t)s.fd_frames * P.wf.regions_per_frame, cap = regions * kWfRegion;
            P.wf.sun_o = (float4 *)s.mem.alloc(cap * sizeof(float4), "wavefront sun-ray queue");
            P.wf.sun_stop = (float *)s.mem.alloc(cap * sizeof(float), "wavefront sun-ray queue");
            P.wf.ibl_o = (float4 *)s.mem.alloc(cap * sizeof(float4), "wavefront IBL-ray queue");
            P.wf.ibl_d = (float4 *)s.mem.alloc(cap * sizeof(float4), "wavefront IBL-ray queue");
            P.wf.counts = (uint32_t *)s.mem.alloc(regions * sizeof(uint32_t), "wavefront region counts");
            P.wf.cursors = (uint32_t *)s.mem.alloc(4 * sizeof(uint32_t), "wavefront chunk cursors");
            hip_check(hipMemsetAsync(P.wf.sun_o, 0, cap * sizeof(float4), s.stream), "queue touch");  // page mappings, as above
            hip_check(hipMemsetAsync(P.wf.sun_stop, 0, cap * sizeof(float), s.stream), "queue touch");
            hip_check(hipMemsetAsync(P.wf.ibl_o, 0, cap * sizeof(float4), s.stream), "queue touch");
            hip_check(hipMemsetAsync(P.wf.ibl_d, 0, cap * sizeof(float4), s.stream), "queue touch");
            hip_check(hipMemsetAsync(P.wf.counts, 0, regions * sizeof(uint32_t), s.stream), "queue touch");
            const char *q = getenv("F3D_WF_QUORUM");
            s.wf_quorum = q ? (uint32_t)std::max(1, std::min(64, atoi(q))) : 0u;
        }
    } else {
        P.trace = nullptr;
        P.trace_first = 0u;
        P.fix_list = P.fix_count = nullptr;
    }
    for (int i = 0; i < 2; i++) {
        if (opts && opts->ext_reservoirs[i]) {
            s.res[i] = (PackedReservoir *)opts->ext_reservoirs[i];
            s.mem.device_bytes += res_n * sizeof(PackedReservoir);  // caller-owned, still part of the working set
        } else {
            s.res[i] = (PackedReservoir *)s.mem.alloc(res_n * sizeof(PackedReservoir), "reservoirs");
            s.owns_reservoirs = true;
        }
        hip_check(hipMemsetAsync(s.res[i], 0, res_n * sizeof(PackedReservoir), s.stream), "reservoir clear");
    }
    if (P.sample_lanes > 1u || s.fd_frames) P.head = (uint2 *)s.mem.alloc(px * sizeof(uint2), "frame head records");
    P.accum_mean = (float4 *)s.mem.alloc(px * sizeof(float4), "accumulation");
    P.welford_m2 = (float *)s.mem.alloc(px * sizeof(float), "welford");
    s.gbuffer_n = (float4 *)s.mem.alloc(px * sizeof(float4), "g-buffer");
    // ray certificates (f3d_cone.h), computed by the G-buffer pass; the A/B switches leave the pointer null = off
#if !defined(F3D_NO_PRIMARY_START)
    P.primary_start = (uint2 *)s.mem.alloc(px * sizeof(uint2), "primary-ray certificates");
#endif
#if !defined(F3D_NO_SUN_CLEAR)
    P.sun_clear = (float2 *)s.mem.alloc(px * sizeof(float2), "sun-ray certificates");
#endif
    s.depth = (float *)s.mem.alloc(px * sizeof(float), "depth AOV");
    s.d_rgba = (uint8_t *)s.mem.alloc(px * 4, "rgba8 output");
    s.d_albedo = (float *)s.mem.alloc(px * 3 * sizeof(float), "albedo AOV");
    s.d_normal = (float *)s.mem.alloc(px * 3 * sizeof(float), "normal AOV");
    if (opts && opts->ext_stats) s.stats = (uint32_t *)opts->ext_stats;
    else s.stats = (uint32_t *)s.mem.alloc(4 * sizeof(uint32_t), "stats");
    hip_check(hipMemsetAsync(P.accum_mean, 0, px * sizeof(float4), s.stream), "accum clear");
    hip_check(hipMemsetAsync(P.welford_m2, 0, px * sizeof(float), s.stream), "welford clear");
    hip_check(hipMemsetAsync(s.stats, 0, 4 * sizeof(uint32_t), s.stream), "stats clear");
    hip_check(hipHostMalloc((void **)&s.host_stats, 4 * sizeof(uint32_t), hipHostMallocDefault), "pinned stats");
    s.mem.note_host_visible(4 * sizeof(uint32_t));
    P.gbuffer_n = s.gbuffer_n;
    P.stats = s.stats;
    clock.lap(kSetupAlloc);

    // memory-budget gate, render_terrain.rs:875-888
    if (s.mem.device_bytes > s.budget)
        fail(F3D_STATUS_RENDER,
             "terrain PT exceeds the memory budget before rendering: tracked total %llu (host-visible %llu) > limit %llu",
             (unsigned long long)s.mem.device_bytes, (unsigned long long)s.mem.host_visible_peak,
             (unsigned long long)s.budget);

    plan_bands(s, opts ? opts->bands : 0u, opts ? opts->band_streams : 0u);
    // tile-map digit 4 = the default map WITHOUT longest-first dispatch (A/B)
    const bool lpt = s.bands.size() == 1 && ((s.variant / 1000) % 10 == 0);
    if ((s.variant / 1000) % 10 == 4) P.tile_map = 2u;
    if (lpt) {
        uint32_t grid = 0;
        const uint32_t tiles = frame_tile_count(P, &grid);
        s.tile_cost = (uint32_t *)s.mem.alloc((size_t)tiles * sizeof(uint32_t), "tile costs");
        s.tile_order = (uint32_t *)s.mem.alloc((size_t)(grid ? grid : 1u) * sizeof(uint32_t), "tile order");
        hip_check(hipMemsetAsync(s.tile_cost, 0, (size_t)tiles * sizeof(uint32_t), s.stream), "tile cost clear");
    }

    // one-shot G-buffer + AOV pass, render_terrain.rs:1091-1121
    P.frame_index = 0;
    P.res_in = s.res[1];
    P.res_out = s.res[0];
    P.collect_stats = 0;
    hip_check(launch_gbuffer(P, s.gbuffer_n, s.depth, s.stream), "g-buffer pass");
    if (s.fd_frames) {
        P.band_begin = s.row_begin;
        P.band_end = s.row_end;
        hip_check(launch_trace_init(P, s.stream), "trace prediction init");
    }
    clock.lap(kSetupPasses);
}

// One band of one frame: frame head (sample-lane form) + frame kernel over the band's rows, on the band's
// stream, after the frame before of this band and of its two neighbours (the head's spatial reuse reads the
// previous frame's reservoirs of +-3 rows; everything else a band touches is its own).
void enqueue_band(f3d_session &s, f3d_session::Band &b, size_t index, uint32_t frame, bool collect) {
    FrameParams &P = s.params;
    P.frame_index = frame;
    P.res_out = s.res[frame & 1u];
    P.res_in = s.res[(frame & 1u) ^ 1u];
    P.collect_stats = collect ? 1u : 0u;
    P.band_begin = b.begin;
    P.band_end = b.end;
    const bool piped = b.stream != s.stream;
    if (b.last != (int64_t)frame - 1 && !(b.last < 0 && frame == 0u) && piped)
        fail(F3D_STATUS_VALUE, "frames must be enqueued in order (band %zu is at frame %lld, frame %u wanted)", index,
             (long long)b.last, frame);
    if (piped) {
        hip_check(hipStreamWaitEvent(b.stream, s.fork, 0), "band fork");
        for (int d = -1; d <= 1; d += 2) {
            const size_t n = index + (size_t)d;  // index - 1 wraps for band 0
            if (n >= s.bands.size() || frame == 0u) continue;
            f3d_session::Band &nb = s.bands[n];
            // the neighbour has frame - 1 enqueued, or (enqueued before this band) frame itself already: its
            // done[(frame - 1) & 1] still holds frame - 1 -- the events alternate and nobody is two frames ahead
            if (nb.last != (int64_t)frame - 1 && nb.last != (int64_t)frame)
                fail(F3D_STATUS_VALUE, "frames must be enqueued in order (band %zu is at frame %lld, frame %u wanted)", n,
                     (long long)nb.last, frame);
            if (nb.stream != b.stream) hip_check(hipStreamWaitEvent(b.stream, nb.done[(frame - 1u) & 1u], 0), "band wait");
        }
    }
    if (P.sample_lanes > 1u) hip_check(launch_head(P, b.stream), "frame head kernel");
    // longest-first dispatch: the order is rebuilt from the newest wave durations every kOrderEvery frames
    // (a tile costs about the same from frame to frame); the first frame of a session runs in image order
    constexpr int64_t kOrderEvery = 4;
    P.tile_cost = s.tile_cost;
    P.tile_order = nullptr;
    if (s.tile_cost && s.cost_frame >= 0) {
        if (s.order_frame < 0 || s.cost_frame - s.order_frame >= kOrderEvery) {
            hip_check(launch_tile_order(P, s.tile_cost, s.tile_order, b.stream), "tile order kernel");
            s.order_frame = s.cost_frame;
        }
        P.tile_order = s.tile_order;
    }
    // the timed bracket is the frame kernel alone (bench.py prices it with ITS bytes: the head record it reads,
    // not the head kernel's own traffic), so that it can be compared with rocprofv3's per-kernel average
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s.timing) {
        hip_check(hipEventCreate(&e0), "event");
        hip_check(hipEventCreate(&e1), "event");
        hip_check(hipEventRecord(e0, b.stream), "event record");
    }
    hip_check(launch_frame(P, s.variant, b.stream), "frame kernel");
    if (s.tile_cost) s.cost_frame = (int64_t)frame;
    P.tile_order = nullptr;
    P.tile_cost = nullptr;
    if (s.timing) {
        hip_check(hipEventRecord(e1, b.stream), "event record");
        s.events.emplace_back(e0, e1);
    }
    if (piped) {
        hip_check(hipEventRecord(b.done[frame & 1u], b.stream), "band done");
        b.unjoined = true;
    }
    b.last = frame;
    P.band_begin = s.row_begin;
    P.band_end = s.row_end;
}

// Start of a batch of band launches: they are ordered after what the session stream holds now.
void fork_bands(f3d_session &s, bool clear_stats) {
    if (clear_stats) hip_check(hipMemsetAsync(s.stats, 0, 2 * sizeof(uint32_t), s.stream), "stats clear");
    if (s.fork) hip_check(hipEventRecord(s.fork, s.stream), "band fork");
}

// Order the session stream after every band launch made so far (before anything reads the results).
void join_bands(f3d_session &s, bool edges_only = false) {
    for (auto &b : s.bands)
        if (b.unjoined && (b.edge || !edges_only)) {
            hip_check(hipStreamWaitEvent(s.stream, b.done[b.last & 1], 0), "band join");
            b.unjoined = false;
        }
}

// How many frames to trace at once from `frame` on: 2, 2, 4, 8, ... up to the session's frames in flight.  A pixel whose
// sun-direction prediction fails is traced again by k_fix inside the ordered chain of merges (~0.17 ms for one pixel of
// the headline scene) in every remaining frame of its batch, because the prediction only learns from merges; which
// direction a pixel's head reads settles within the first few frames (reservoirs spread 3 pixels a frame), so the early
// batches are short and the long ones start from settled predictions (17 re-traced pixel-frames in 130 frames at 1080p).
// Frames of the next trace batch.  The first two batches are two frames each: the sun-direction choice k_trace predicts
// (wi or normalize(wi), from the flags the merges leave behind) settles within the first frames, and a mispredicted
// pixel-frame is traced again one lane at a time.  From frame 4 on the batches are as large as the session holds (round 2
// ramped 4, 8, 16 with the frame number: measured 3 % slower over frames 4..35 of a thin strip, with no fewer re-traces --
// tools/experiments/strip_ramp.py; F3D_FD_FULL_FROM=<frame> moves the switch for experiments).
uint32_t trace_batch(const f3d_session &s, uint32_t frame, uint32_t remaining) {
    static const uint32_t full_from = getenv("F3D_FD_FULL_FROM") ? (uint32_t)std::max(2, atoi(getenv("F3D_FD_FULL_FROM"))) : 4u;
    if (frame >= full_from) return std::max(1u, std::min(s.fd_frames, remaining));
    return std::max(1u, std::min(std::min(s.fd_frames, remaining), frame < 2u ? 2u - frame : 2u));
}

// ---- frames in flight: trace a batch of frames in one launch, then merge them in order ------------------------
void enqueue_trace(f3d_session &s, uint32_t first, uint32_t count) {
    if (!s.fd_frames) fail(F3D_STATUS_VALUE, "this session has no frames in flight (f3d_session_opts.frames_in_flight)");
    if (count == 0u || count > s.fd_frames) fail(F3D_STATUS_VALUE, "a trace batch holds 1..%u frames (got %u)", s.fd_frames, count);
    FrameParams &P = s.params;
    P.frame_index = first;
    P.trace_first = first;
    P.band_begin = s.row_begin;
    P.band_end = s.row_end;
    constexpr int64_t kOrderEvery = 4;
    P.tile_cost = s.tile_cost;
    P.tile_order = nullptr;
    if (s.tile_cost && s.cost_frame >= 0) {
        if (s.order_frame < 0 || s.cost_frame - s.order_frame >= kOrderEvery) {
            hip_check(launch_tile_order(P, s.tile_cost, s.tile_order, s.stream), "tile order kernel");
            s.order_frame = s.cost_frame;
        }
        P.tile_order = s.tile_order;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s.timing) {
        hip_check(hipEventCreate(&e0), "event");
        hip_check(hipEventCreate(&e1), "event");
        hip_check(hipEventRecord(e0, s.stream), "event record");
    }
    if (s.wavefront) hip_check(launch_trace_wavefront(P, count, s.wf_quorum, s.stream), "wavefront trace kernels");
    else hip_check(launch_trace(P, count, s.stream), "trace kernel");
    if (s.tile_cost) s.cost_frame = (int64_t)first;
    P.tile_order = nullptr;
    P.tile_cost = nullptr;
    if (s.timing) {
        hip_check(hipEventRecord(e1, s.stream), "event record");
        s.events.emplace_back(e0, e1);
    }
    s.trace_first = first;
    s.trace_count = count;
}

void enqueue_merge(f3d_session &s, uint32_t frame, bool collect) {
    if (!s.fd_frames || s.trace_first < 0 || frame < (uint32_t)s.trace_first || frame >= (uint32_t)s.trace_first + s.trace_count)
        fail(F3D_STATUS_VALUE, "frame %u is not in the traced batch [%lld, %lld)", frame, (long long)s.trace_first,
             (long long)s.trace_first + s.trace_count);
    FrameParams &P = s.params;
    if (collect) hip_check(hipMemsetAsync(s.stats, 0, 2 * sizeof(uint32_t), s.stream), "stats clear");
    P.frame_index = frame;
    P.trace_first = (uint32_t)s.trace_first;
    P.res_out = s.res[frame & 1u];
    P.res_in = s.res[(frame & 1u) ^ 1u];
    P.collect_stats = collect ? 1u : 0u;
    P.band_begin = s.row_begin;
    P.band_end = s.row_end;
    hip_check(launch_merge(P, s.stream), "merge kernel");
    for (auto &b : s.bands) b.last = frame;
}

// Frames [first, first + count), the last one closing a convergence window if `collect_last`.
void enqueue_frame(f3d_session &s, uint32_t frame, bool collect, uint32_t part, bool fork);
void fork_bands(f3d_session &s, bool clear_stats);
void enqueue_range(f3d_session &s, uint32_t first, uint32_t count, bool collect_last) {
    if (s.fd_frames) {  // frames in flight: batches traced in one launch, merged in order
        for (uint32_t done = 0; done < count;) {
            const uint32_t n = trace_batch(s, first + done, count - done);
            enqueue_trace(s, first + done, n);
            for (uint32_t i = 0; i < n; i++) enqueue_merge(s, first + done + i, collect_last && done + i + 1 == count);
            done += n;
        }
        return;
    }
    if (count) fork_bands(s, collect_last);
    for (uint32_t i = 0; i < count; i++) enqueue_frame(s, first + i, collect_last && i + 1 == count, 0u, false);
}

// part: 0 every band; 1 the edge bands (halo donors of a multi-GPU strip), then the session stream is
// ordered after them; 2 the interior bands
void enqueue_frame(f3d_session &s, uint32_t frame, bool collect, uint32_t part, bool fork) {
    if (fork && part != 2u) fork_bands(s, collect);
    for (size_t i = 0; i < s.bands.size(); i++) {
        f3d_session::Band &b = s.bands[i];
        if ((part == 1u && !b.edge) || (part == 2u && b.edge)) continue;
        enqueue_band(s, b, i, frame, collect);
    }
    if (part == 1u) join_bands(s, true);
}

// Cut the strip into bands (multiples of 8 rows from its first row: every tile shape of the frame kernels and
// the 8x8 tiles of the head kernel start on such a row).  With three or more bands the first and the last are
// EDGE bands -- 8 rows at the top, the last 3..10 rows -- so that a multi-GPU strip can ship its halo rows
// while the interior bands still render.
void plan_bands(f3d_session &s, uint32_t want, uint32_t want_streams) {
    const uint32_t rows = s.rows;
    // automatic = one band: cutting the strip into bands on several streams reproduces the image bit for bit
    // (tests) but measured SLOWER on MI355X (profiles/README.md "band pipelining": kernels of different
    // streams did not overlap), whereas longest-first dispatch of one launch removes most of the tail
    if (want == 0u) want = 1u;
    if (want > 64u) want = 64u;
    std::vector<std::pair<uint32_t, uint32_t>> cuts;  // (first row, end row) relative to the strip
    const uint32_t bottom = rows > kHaloRows ? ((rows - kHaloRows) / 8u) * 8u : 0u;
    if (want >= 3u && bottom >= 16u) {
        const uint32_t units = (bottom - 8u) / 8u, inner = std::min(want - 2u, units);
        cuts.emplace_back(0u, 8u);
        for (uint32_t k = 0; k < inner; k++)
            cuts.emplace_back(8u + (units * k / inner) * 8u, 8u + (units * (k + 1u) / inner) * 8u);
        cuts.emplace_back(bottom, rows);
    } else if (want == 2u && rows >= 16u) {
        const uint32_t mid = (rows / 16u) * 8u;
        cuts.emplace_back(0u, mid);
        cuts.emplace_back(mid, rows);
    } else {
        cuts.emplace_back(0u, rows);
    }
    const size_t n = cuts.size();
    uint32_t n_streams = n == 1 ? 0u : (want_streams ? want_streams : 4u);
    if (n_streams > n) n_streams = (uint32_t)n;
    for (uint32_t i = 0; i < n_streams; i++) {
        hipStream_t st = nullptr;
        hip_check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "band stream");
        s.band_streams.push_back(st);
    }
    if (n > 1) hip_check(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming), "band event");
    s.bands.resize(n);
    for (size_t i = 0; i < n; i++) {
        f3d_session::Band &b = s.bands[i];
        b.begin = s.row_begin + cuts[i].first;
        b.end = s.row_begin + cuts[i].second;
        b.edge = n < 3 || i == 0 || i + 1 == n;
        b.stream = n == 1 ? s.stream : s.band_streams[i % n_streams];
        if (n > 1)
            for (auto &e : b.done) hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "band event");
    }
}

bool closes_window(uint32_t frame, uint32_t max_frames) {
    const uint32_t frames = frame + 1u;
    return frames % kWelfordWindow == 0u || frames == max_frames;
}

void resolve(f3d_session &s, uint32_t frames, uint8_t *d_rgba, float *d_albedo, float *d_normal) {
    join_bands(s);
    ResolveParams R{};
    R.frame = s.params;
    R.frame.res_in = s.res[(frames - 1u) & 1u];
    R.frames = frames;
    R.rgba = d_rgba;
    R.albedo = d_albedo;
    R.normal = d_normal;
    R.aether = s.aether;
    R.depth = s.depth;
    hip_check(hipMemsetAsync(s.stats + 2, 0, 2 * sizeof(uint32_t), s.stream), "stats clear");
    hip_check(launch_resolve(R, s.stream), "resolve kernel");
}

}  // namespace

#include "f3d_host_halo.h"  // peer halos: the pull kernel, the batch enqueue and their C ABI

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int f3d_session_create(const f3d_terrain_ref_desc *desc, const f3d_session_opts *opts, f3d_session **session,
                       char *err, size_t errlen) {
    if (err && errlen) err[0] = 0;
    if (!desc || !session) return F3D_STATUS_VALUE;
    f3d_session *s = new (std::nothrow) f3d_session();
    if (!s) return F3D_STATUS_DEVICE;
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    const int rc = c_abi(err, errlen, [&] { session_init(*s, *desc, opts); });
    if (caller_device >= 0 && caller_device != s->device) (void)hipSetDevice(caller_device);  // session_init bound its own
    if (rc != F3D_STATUS_OK) {
        DeviceGuard g(s->device);
        delete s;
        return rc;
    }
    *session = s;
    return F3D_STATUS_OK;
}

void f3d_session_destroy(f3d_session *session) {
    if (!session) return;
    DeviceGuard g(session->device);
    for (hipStream_t st : session->band_streams) (void)hipStreamSynchronize(st);
    (void)hipStreamSynchronize(session->stream);
    delete session;
}

int f3d_session_enqueue_frames(f3d_session *s, uint32_t first_frame, uint32_t count, int32_t collect_stats_on_last,
                               char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        enqueue_range(*s, first_frame, count, collect_stats_on_last != 0);
    });
}

int f3d_session_enqueue_trace(f3d_session *s, uint32_t first_frame, uint32_t count, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        enqueue_trace(*s, first_frame, count);
    });
}

int f3d_session_enqueue_merge(f3d_session *s, uint32_t frame, int32_t collect_stats, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        enqueue_merge(*s, frame, collect_stats != 0);
    });
}

uint32_t f3d_session_frames_in_flight(f3d_session *s) { return s ? s->fd_frames : 0u; }
uint32_t f3d_session_trace_batch(f3d_session *s, uint32_t frame, uint32_t remaining) {
    return (s && s->fd_frames && remaining) ? trace_batch(*s, frame, remaining) : 0u;
}

int f3d_session_retraced_pixels(f3d_session *s, uint64_t *total) {
    return c_abi(nullptr, 0, [&] {
        DeviceGuard g(checked(s).device);
        uint32_t host[4] = {0, 0, 0, 0};
        if (s->fd_frames) {
            hip_check(hipStreamSynchronize(s->stream), "stream sync");
            hip_check(hipMemcpy(host, s->params.fix_count, sizeof(host), hipMemcpyDeviceToHost), "retrace counters");
        }
        if (total) *total = host[2];
    });
}

int f3d_session_enqueue_frame_part(f3d_session *s, uint32_t frame, uint32_t part, int32_t collect_stats, char *err,
                                   size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        if (part != 1u && part != 2u) fail(F3D_STATUS_VALUE, "frame part must be 1 (edge rows) or 2 (interior)");
        if (s->fd_frames) fail(F3D_STATUS_VALUE, "sessions with frames in flight are driven by enqueue_trace / enqueue_merge");
        enqueue_frame(*s, frame, collect_stats != 0, part, true);
    });
}

int f3d_session_window_stats(f3d_session *s, float *max_m2, int32_t *nonfinite, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        join_bands(*s);
        hip_check(hipMemcpyAsync(s->host_stats, s->stats, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream),
                  "stats readback");
        hip_check(hipStreamSynchronize(s->stream), "stream sync");
        if (max_m2) *max_m2 = f_from_bits(s->host_stats[0]);
        if (nonfinite) *nonfinite = s->host_stats[1] != 0u;
    });
}

int f3d_session_set_accumulation(f3d_session *s, const float *sums_rgba, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        if (!sums_rgba) fail(F3D_STATUS_VALUE, "null accumulation");
        const size_t px = (size_t)s->rows * s->width;
        // accum_mean holds (sum r, sum g, sum b, Welford mean): the radiance sums are replaced, the statistic is kept
        std::vector<float4> host(px);
        hip_check(hipStreamSynchronize(s->stream), "accumulation sync");
        hip_check(hipMemcpy(host.data(), s->params.accum_mean, px * sizeof(float4), hipMemcpyDeviceToHost), "accumulation read-back");
        for (size_t i = 0; i < px; i++) {
            host[i].x = sums_rgba[4 * i];
            host[i].y = sums_rgba[4 * i + 1];
            host[i].z = sums_rgba[4 * i + 2];
        }
        hip_check(hipMemcpy(s->params.accum_mean, host.data(), px * sizeof(float4), hipMemcpyHostToDevice), "accumulation upload");
    });
}

int f3d_session_resolve_device(f3d_session *s, uint32_t frames, void *d_rgba, void *d_albedo, void *d_normal,
                               void *d_depth, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        if (frames == 0) fail(F3D_STATUS_VALUE, "resolve needs at least one accumulated frame");
        resolve(*s, frames, d_rgba ? (uint8_t *)d_rgba : s->d_rgba, d_albedo ? (float *)d_albedo : s->d_albedo,
                d_normal ? (float *)d_normal : s->d_normal);
        if (d_depth)
            hip_check(hipMemcpyAsync(d_depth, s->depth, (size_t)s->rows * s->width * sizeof(float),
                                     hipMemcpyDeviceToDevice, s->stream), "depth copy");
    });
}

int f3d_session_resolve(f3d_session *s, uint32_t frames, uint8_t *rgba, float *albedo, float *normal, float *depth,
                        int32_t *any_valid_reservoir, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        if (frames == 0) fail(F3D_STATUS_VALUE, "resolve needs at least one accumulated frame");
        resolve(*s, frames, s->d_rgba, s->d_albedo, s->d_normal);
        const size_t px = (size_t)s->rows * s->width;
        hip_check(hipMemcpyAsync(s->host_stats, s->stats, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream),
                  "stats readback");
        if (rgba) hip_check(hipMemcpyAsync(rgba, s->d_rgba, px * 4, hipMemcpyDeviceToHost, s->stream), "rgba readback");
        if (albedo)
            hip_check(hipMemcpyAsync(albedo, s->d_albedo, px * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream),
                      "albedo readback");
        if (normal)
            hip_check(hipMemcpyAsync(normal, s->d_normal, px * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream),
                      "normal readback");
        if (depth)
            hip_check(hipMemcpyAsync(depth, s->depth, px * sizeof(float), hipMemcpyDeviceToHost, s->stream),
                      "depth readback");
        hip_check(hipStreamSynchronize(s->stream), "stream sync");
        s->mem.note_host_visible(px * 3 * sizeof(float));
        if (s->host_stats[3] != 0u)
            fail(F3D_STATUS_RENDER, "terrain PT reservoir bookkeeping produced non-finite values");
        if (any_valid_reservoir) *any_valid_reservoir = s->host_stats[2] != 0u;
    });
}

int f3d_session_setup_ms(f3d_session *s, double *out, uint32_t count) {
    if (!s || !out) return F3D_STATUS_VALUE;
    for (uint32_t i = 0; i < count; i++) out[i] = i < (uint32_t)kSetupPhases ? s->setup_ms[i] : 0.0;
    return F3D_STATUS_OK;
}

int f3d_session_info(f3d_session *s, uint64_t *gpu_resource_bytes, uint64_t *minmax_pyramid_bytes,
                     uint64_t *peak_host_visible_bytes, uint32_t *rows, uint32_t *width) {
    if (!s) return F3D_STATUS_VALUE;
    if (gpu_resource_bytes) *gpu_resource_bytes = s->mem.device_bytes;
    if (minmax_pyramid_bytes) *minmax_pyramid_bytes = s->tables.bytes;
    if (peak_host_visible_bytes) *peak_host_visible_bytes = s->mem.host_visible_peak;
    if (rows) *rows = s->rows;
    if (width) *width = s->width;
    return F3D_STATUS_OK;
}

int f3d_session_kernel_timing(f3d_session *s, int32_t enable, double *avg_ms, uint32_t *launches) {
    if (!s) return F3D_STATUS_VALUE;
    DeviceGuard g(s->device);
    if (enable) {
        for (auto &e : s->events) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        s->events.clear();
        s->timing = true;
        return F3D_STATUS_OK;
    }
    s->timing = false;
    for (hipStream_t st : s->band_streams)
        if (hipStreamSynchronize(st) != hipSuccess) return F3D_STATUS_DEVICE;
    if (hipStreamSynchronize(s->stream) != hipSuccess) return F3D_STATUS_DEVICE;
    double total = 0.0;
    for (auto &e : s->events) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.first, e.second) != hipSuccess) return F3D_STATUS_DEVICE;
        total += ms;
    }
    if (launches) *launches = (uint32_t)s->events.size();
    if (avg_ms) *avg_ms = s->events.empty() ? 0.0 : total / (double)s->events.size();
    return F3D_STATUS_OK;
}

uint32_t f3d_session_sample_lanes(f3d_session *s) { return s ? s->params.sample_lanes : 0u; }
// (waits for the session's stream: the certificate pass wrote the buffer there, and whoever takes the pointer -- the PBR tracer
// launches on the null stream -- is not ordered behind a non-blocking stream; a start read too early would skip terrain)
const void *f3d_session_primary_start(f3d_session *s) {
    if (!s || !(s->params.cam.cone_delta >= 0.0f) || !s->params.primary_start) return nullptr;
    DeviceGuard guard(s->device);
    if (hipStreamSynchronize(s->stream) != hipSuccess) return nullptr;
    return s->params.primary_start;
}

// What the last fused frame cost, row by row: every wave of the frame kernel leaves its duration in tile_cost (the input of
// the longest-first dispatch); a tile's time is spread over its rows.  The strip driver cuts the image where these sums are
// equal (forge3d_amd/distributed.py) -- one probe frame instead of rounds of whole-loop probe renders.
int f3d_session_row_costs(f3d_session *s, float *out, uint32_t rows, char *err, size_t errlen) {
    try {
        if (!s || !out) fail(F3D_STATUS_VALUE, "null argument");
        if (rows != s->rows) fail(F3D_STATUS_VALUE, "the session owns %u rows, %u asked for", s->rows, rows);
        if (!s->tile_cost || s->cost_frame < 0) fail(F3D_STATUS_VALUE, "no frame has left its tile costs yet (a one-band session with the default tile map does, from its first fused frame on)");
        DeviceGuard guard(s->device);
        hip_check(hipStreamSynchronize(s->stream), "row costs");
        FrameParams P = s->params;
        P.band_begin = s->row_begin;
        P.band_end = s->row_end;
        const uint32_t tiles = frame_tile_count(P, nullptr);
        std::vector<uint32_t> cost(tiles);
        hip_check(hipMemcpy(cost.data(), s->tile_cost, (size_t)tiles * sizeof(uint32_t), hipMemcpyDeviceToHost), "tile costs");
        const uint32_t lanes = P.sample_lanes ? P.sample_lanes : 1u;
        const uint32_t log_s = lanes == 1u ? 0u : (lanes == 2u ? 1u : (lanes == 4u ? 2u : 3u)), log_w = lanes <= 2u ? 3u : 2u, log_h = 6u - log_s - log_w;  // TileShape<S>
        const uint32_t tiles_x = (s->width + (1u << log_w) - 1u) >> log_w, th = 1u << log_h;
        std::vector<double> sum(rows, 0.0);
        for (uint32_t t = 0; t < tiles; t++) {
            const uint32_t r0 = (t / tiles_x) * th, r1 = std::min(rows, r0 + th);
            for (uint32_t r = r0; r < r1; r++) sum[r] += (double)cost[t] / (double)(r1 - r0);
        }
        for (uint32_t r = 0; r < rows; r++) out[r] = (float)sum[r];
        return F3D_STATUS_OK;
    } catch (const Failure &f) {
        return report(f, err, errlen);
    }
}
uint32_t f3d_halo_rows(void) { return kHaloRows; }

// Diagnostics: FNV-style hashes of everything a frame launch reads -- the by-value uniforms (camera, light, terrain and
// mesh scalars) and the device buffers behind them, plus the per-pixel state.  Equal fingerprints => equal frames.
int f3d_session_fingerprint(f3d_session *s, uint64_t *out, uint32_t count) {
    if (!s || !out || count < 16u) return F3D_STATUS_VALUE;
    try {
        DeviceGuard guard(s->device);
        hip_check(hipStreamSynchronize(s->stream), "fingerprint");
        for (auto &b : s->bands)
            if (b.stream) hip_check(hipStreamSynchronize(b.stream), "fingerprint");
        const FrameParams &P = s->params;
        auto dev = [&](const void *p, size_t bytes) -> uint64_t {
            if (!p || !bytes) return 0ull;
            std::vector<uint8_t> host(bytes);
            hip_check(hipMemcpy(host.data(), p, bytes, hipMemcpyDeviceToHost), "fingerprint download");
            return hash_bytes(host.data(), bytes, 0x243F6A8885A308D3ull);
        };
        const TableLayout &L = s->scene ? s->scene->tables.layout : s->tables.layout;
        const size_t px = (size_t)s->rows * s->width, res_n = (size_t)(s->rows + 2u * kHaloRows) * s->width;
        TerrainDev t = P.terrain;
        t.leaves = nullptr;
        t.nodes = nullptr;
        t.bands = nullptr;
        t.horizon = nullptr;
        t.mesh_bands = nullptr;  // (device addresses are not part of what a launch computes with: the tables they name are hashed by content)
        t.mesh_cell_start = nullptr;
        t.mesh_cell_tris = nullptr;
        MeshDev m = P.mesh;
        m.vertices = nullptr;
        m.indices = nullptr;
        m.bvh_nodes = nullptr;
        m.bvh_tris = nullptr;
        m.bvh4_nodes = nullptr;
        const uint32_t scalars[8] = {P.spp, P.row_begin, P.row_end, P.tile_map, P.sample_lanes, P.same_sun, P.env.width, P.env.height};
        out[0] = hash_bytes(&P.cam, sizeof(P.cam), 1);
        out[1] = hash_bytes(&P.light, sizeof(P.light), 2);
        out[2] = hash_bytes(&t, sizeof(t), 3);
        out[3] = hash_bytes(&m, sizeof(m), 4);
        out[4] = hash_bytes(scalars, sizeof(scalars), 5) ^ hash_bytes(&P.env.intensity, sizeof(float), 6);
        out[5] = dev(P.terrain.leaves, L.leaf_count * sizeof(LeafRec));
        out[6] = dev(P.terrain.bands, L.band_count * sizeof(NodeRec));
        out[7] = dev(P.mesh.vertices, (size_t)P.mesh.vertex_count * sizeof(float4));
        out[8] = dev(P.mesh.indices, (size_t)P.mesh.index_count * sizeof(uint32_t));
        out[9] = P.mesh.bvh4_nodes ? dev(P.mesh.bvh4_nodes, (size_t)P.mesh.bvh4_node_count * sizeof(Bvh4Node))
                                   : dev(P.mesh.bvh_nodes, (size_t)P.mesh.bvh_node_count * sizeof(BvhNode));
        out[10] = dev(P.mesh.bvh_tris, (size_t)P.mesh.index_count * sizeof(float4));
        out[11] = dev(P.env.texels, (size_t)P.env.width * P.env.height * sizeof(float4));
        out[12] = dev(s->gbuffer_n, px * sizeof(float4));
        out[13] = dev(s->res[0], res_n * sizeof(PackedReservoir)) ^ (dev(s->res[1], res_n * sizeof(PackedReservoir)) * 3ull);
        out[14] = dev(P.accum_mean, px * sizeof(float4)) ^ (dev(P.welford_m2, px * sizeof(float)) * 3ull);
        out[15] = dev(P.head, P.head ? px * sizeof(uint2) : 0);
        return F3D_STATUS_OK;
    } catch (...) {
        return F3D_STATUS_DEVICE;
    }
}

int f3d_session_debug_wave_times(f3d_session *s, void *device_buffer) {
#if defined(F3D_WAVE_TIMES)
    if (!s) return F3D_STATUS_VALUE;
    s->params.wave_times = (unsigned long long *)device_buffer;
    return F3D_STATUS_OK;
#else
    (void)s;
    (void)device_buffer;
    return F3D_STATUS_VALUE;  // diagnostics are compiled out of the shipped library
#endif
}

int f3d_terrain_ref_render(const f3d_terrain_ref_desc *desc, f3d_terrain_ref_out *out, char *err, size_t errlen) {
    if (err && errlen) err[0] = 0;
    if (!desc || !out) return F3D_STATUS_VALUE;
    f3d_session *s = new (std::nothrow) f3d_session();
    if (!s) return F3D_STATUS_DEVICE;
    int rc = F3D_STATUS_OK;
    try {
        const double t_setup = now_s();
        f3d_session_opts one_shot{};
        one_shot.struct_size = (uint32_t)sizeof(one_shot);
        one_shot.device = -1;
        one_shot.frames_in_flight = F3D_FRAMES_IN_FLIGHT_AUTO;  // small images: batches of frames per launch
        session_init(*s, *desc, &one_shot);
        hip_check(hipStreamSynchronize(s->stream), "setup sync");
        out->setup_seconds = now_s() - t_setup;

        // accumulate until converged or capped, render_terrain.rs:1123-1244.  The reference
        // checks after every frame whether a 32-frame window just closed; here whole windows
        // are enqueued without touching the host and only the closing frame reports.
        uint32_t frames = 0;
        float variance = INFINITY;
        bool converged = false;
        const double t_loop = now_s();
        while (frames < desc->max_frames) {
            uint32_t stop = (frames / kWelfordWindow + 1u) * kWelfordWindow;
            if (stop > desc->max_frames) stop = desc->max_frames;
            enqueue_range(*s, frames, stop - frames, true);
            frames = stop;
            const uint32_t n_window = ((frames - 1u) % kWelfordWindow) + 1u;
            if (n_window >= 2u) {
                float m2 = 0.0f;
                int32_t nonfinite = 0;
                char e2[256];
                if (f3d_session_window_stats(s, &m2, &nonfinite, e2, sizeof(e2)) != 0) fail(F3D_STATUS_DEVICE, "%s", e2);
                if (nonfinite) fail(F3D_STATUS_RENDER, "terrain PT produced non-finite variance (NaN in accumulation)");
                variance = f_max(0.0f, m2 / ((float)n_window - 1.0f));
                if (frames >= desc->min_frames && variance < desc->variance_threshold) {
                    converged = true;
                    break;
                }
            }
        }
        hip_check(hipStreamSynchronize(s->stream), "loop sync");
        out->loop_seconds = now_s() - t_loop;
        out->frames = frames;
        out->variance = variance;
        out->converged = converged ? 1 : 0;
        if (!converged)
            fail(F3D_STATUS_RENDER,
                 "terrain PT did not converge: per-pixel luminance variance %s over the last %u-frame window after "
                 "%u frames (threshold %s); raise max_frames or simplify the scene \xe2\x80\x94 refusing to return a "
                 "fake reference",
                 rust_exp(variance, 3).c_str(), kWelfordWindow, frames, rust_exp(desc->variance_threshold, 1).c_str());

        const double t_read = now_s();
        int32_t any_valid = 0;
        char e2[256];
        rc = f3d_session_resolve(s, frames, out->rgba, out->albedo, out->normal, out->depth, &any_valid, e2, sizeof(e2));
        if (rc != 0) fail(rc, "%s", e2);
        if (s->require_valid_reservoirs && !any_valid)
            fail(F3D_STATUS_RENDER,
                 "terrain PT ReSTIR reuse chain produced no valid reservoirs for a sun-lit scene \xe2\x80\x94 "
                 "temporal/spatial reuse is broken");
        out->readback_seconds = now_s() - t_read;
        out->gpu_resource_bytes = s->mem.device_bytes;
        out->minmax_pyramid_bytes = s->tables.bytes;
        out->peak_host_visible_bytes = s->mem.host_visible_peak;
        // budget guardrail on the host-visible peak, render_terrain.rs:1397-1404
        if (out->peak_host_visible_bytes > s->budget)
            fail(F3D_STATUS_RENDER, "terrain PT exceeded the host-visible budget: peak %llu > limit %llu",
                 (unsigned long long)out->peak_host_visible_bytes, (unsigned long long)s->budget);
    } catch (const Failure &f) {
        rc = report(f, err, errlen);
    } catch (const std::exception &e) {
        if (err && errlen) snprintf(err, errlen, "host failure: %s", e.what());
        rc = F3D_STATUS_DEVICE;
    } catch (...) {
        rc = F3D_STATUS_DEVICE;
    }
    f3d_session_destroy(s);
    return rc;
}

int f3d_build_minmax_mips(const float *heights, uint32_t width, uint32_t height, float *levels_out, uint32_t *dims_out,
                          uint32_t max_levels, uint64_t *total_floats, char *err, size_t errlen) {
    if (err && errlen) err[0] = 0;
    Ledger mem;
    int rc = F3D_STATUS_OK;
    try {
        if (width < 2 || height < 2)
            fail(F3D_STATUS_UPLOAD, "terrain heightfield must be at least 2x2 texels, got %ux%u", width, height);
        const size_t n = (size_t)width * height;
        for (size_t i = 0; i < n; i++)
            if (!std::isfinite(heights[i])) fail(F3D_STATUS_UPLOAD, "terrain heightfield contains non-finite samples");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            fail(F3D_STATUS_DEVICE, "no HIP device available: libf3dhip has no CPU fallback");
        float *d_h = (float *)mem.alloc(n * sizeof(float), "DEM upload");
        hip_check(hipMemcpy(d_h, heights, n * sizeof(float), hipMemcpyHostToDevice), "DEM upload");
        TerrainTables t = build_tables(mem, d_h, width, height, 1.0f, nullptr, true);
        hip_check(hipDeviceSynchronize(), "table build");
        const TableLayout &L = t.layout;
        const uint32_t levels = L.levels;
        uint64_t tot = 0;
        for (uint32_t l = 0; l < levels; l++) tot += (uint64_t)L.level_w[l] * L.level_h[l] * 2;
        if (total_floats) *total_floats = tot;
        if (dims_out)
            for (uint32_t l = 0; l < levels && l < max_levels; l++) {
                dims_out[2 * l] = L.level_w[l];
                dims_out[2 * l + 1] = L.level_h[l];
            }
        if (levels_out) {
            std::vector<LeafRec> leaves(L.leaf_count);
            std::vector<NodeRec> nodes(L.node_count ? L.node_count : 1);
            hip_check(hipMemcpy(leaves.data(), t.leaves, L.leaf_count * sizeof(LeafRec), hipMemcpyDeviceToHost), "readback");
            if (L.node_count)
                hip_check(hipMemcpy(nodes.data(), t.nodes, L.node_count * sizeof(NodeRec), hipMemcpyDeviceToHost), "readback");
            uint64_t off = 0;
            for (uint32_t l = 0; l < levels; l++) {
                for (uint32_t y = 0; y < L.level_h[l]; y++)
                    for (uint32_t x = 0; x < L.level_w[l]; x++) {
                        float mn, mx;
                        if (l == 0) {
                            if (x < t.dev.cell_w && y < t.dev.cell_h) {
                                const LeafRec &h = leaves[tiled_index(x, y, t.dev.tiles_x[0])];
                                mn = min4(h);
                                mx = max4(h);
                            } else {
                                mn = INFINITY;
                                mx = -INFINITY;
                            }
                        } else {
                            const NodeRec &r = nodes[t.dev.node_offset[l] + tiled_index(x, y, t.dev.tiles_x[l])];
                            mn = r.mn;
                            mx = r.mx;
                        }
                        levels_out[off++] = mn;
                        levels_out[off++] = mx;
                    }
            }
        }
        rc = (int)levels;
    } catch (const Failure &f) {
        rc = -report(f, err, errlen);
    }
    mem.release();
    return rc;
}

int f3d_terrain_trace_batch(const float *heights, uint32_t width, uint32_t height, float origin_x, float origin_z,
                            float spacing_x, float spacing_z, float exaggeration, float inv_two_r_prime,
                            uint32_t curvature_enabled, const float *rays, uint32_t n, int32_t any_hit,
                            int32_t apply_curvature, uint32_t *out_hit, float *out_t, float *out_normal, char *err,
                            size_t errlen) {
    if (err && errlen) err[0] = 0;
    Ledger mem;
    int rc = F3D_STATUS_OK;
    try {
        if (width < 2 || height < 2)
            fail(F3D_STATUS_UPLOAD, "terrain heightfield must be at least 2x2 texels, got %ux%u", width, height);
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            fail(F3D_STATUS_DEVICE, "no HIP device available: libf3dhip has no CPU fallback");
        const size_t dem_n = (size_t)width * height;
        float *d_h = (float *)mem.alloc(dem_n * sizeof(float), "DEM upload");
        hip_check(hipMemcpy(d_h, heights, dem_n * sizeof(float), hipMemcpyHostToDevice), "DEM upload");
        TerrainTables t = build_tables(mem, d_h, width, height, exaggeration, nullptr, true);
        RayBatchParams B{};
        B.terrain = t.dev;
        B.terrain.origin_x = origin_x;
        B.terrain.origin_z = origin_z;
        B.terrain.spacing_x = spacing_x;
        B.terrain.spacing_z = spacing_z;
        B.terrain.inv_spacing_x = 1.0f / spacing_x;
        B.terrain.inv_spacing_z = 1.0f / spacing_z;
        B.terrain.inv_two_r_prime = inv_two_r_prime;
        B.terrain.curvature_enabled = curvature_enabled;
        float4 *d_rays = (float4 *)mem.alloc((size_t)n * 32, "rays");
        hip_check(hipMemcpy(d_rays, rays, (size_t)n * 32, hipMemcpyHostToDevice), "ray upload");
        B.rays = d_rays;
        B.n = n;
        // 0 closest / 1 any-hit through the sorted descent; 2 any / 3 closest through the march,
        // +4: the march starts in the origin cell (secondary rays) instead of at the root
        // bits 8..15: ray-sharing threshold override (0 = default)
        B.any_hit = (uint32_t)any_hit & 3u;
        B.start_in_cell = ((uint32_t)any_hit >> 2) & 1u;
        B.terrain.share_below = ((uint32_t)any_hit >> 8) & 255u;
        B.apply_curvature = apply_curvature != 0;
        B.out_hit = (uint32_t *)mem.alloc((size_t)n * 4, "hits");
        B.out_t = (float *)mem.alloc((size_t)n * 4, "t");
        B.out_normal = (float *)mem.alloc((size_t)n * 12, "normals");
        hip_check(launch_ray_batch(B, nullptr), "ray batch kernel");
        hip_check(hipDeviceSynchronize(), "ray batch");
        hip_check(hipMemcpy(out_hit, B.out_hit, (size_t)n * 4, hipMemcpyDeviceToHost), "readback");
        if (out_t) hip_check(hipMemcpy(out_t, B.out_t, (size_t)n * 4, hipMemcpyDeviceToHost), "readback");
        if (out_normal) hip_check(hipMemcpy(out_normal, B.out_normal, (size_t)n * 12, hipMemcpyDeviceToHost), "readback");
    } catch (const Failure &f) {
        rc = report(f, err, errlen);
    }
    mem.release();
    return rc;
}

int f3d_effective_radius_m(int32_t earth_model, double latitude_deg, double sphere_radius_m, int32_t refraction_model,
                           double pressure_mbar, double temperature_c, double refraction_k, double azimuth_deg,
                           double *radius_out, char *err, size_t errlen) {
    if (err && errlen) err[0] = 0;
    try {
        *radius_out = effective_radius(earth_model, latitude_deg, sphere_radius_m, refraction_model, pressure_mbar,
                                       temperature_c, refraction_k, azimuth_deg);
    } catch (const Failure &f) {
        return report(f, err, errlen);
    }
    return F3D_STATUS_OK;
}

void f3d_scene_cache_limit(uint32_t entries) {
    std::lock_guard<std::mutex> lock(g_scene_mutex);
    g_scene_limit = entries;
    while (g_scene_cache.size() > g_scene_limit) {
        size_t oldest = 0;
        for (size_t i = 1; i < g_scene_cache.size(); i++)
            if (g_scene_cache[i]->stamp < g_scene_cache[oldest]->stamp) oldest = i;
        g_scene_cache.erase(g_scene_cache.begin() + (long)oldest);
    }
    while (g_mesh_cache.size() > g_scene_limit) {  // (the mesh cache follows the same limit)
        size_t oldest = 0;
        for (size_t i = 1; i < g_mesh_cache.size(); i++)
            if (g_mesh_cache[i]->stamp < g_mesh_cache[oldest]->stamp) oldest = i;
        g_mesh_cache.erase(g_mesh_cache.begin() + (long)oldest);
    }
}

uint32_t f3d_scene_cache_entries(void) {
    std::lock_guard<std::mutex> lock(g_scene_mutex);
    return (uint32_t)g_scene_cache.size();
}

int f3d_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *f3d_device_name(int32_t device) {
    static thread_local char name[256];
    name[0] = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) snprintf(name, sizeof(name), "%s", prop.gcnArchName);
    return name;
}

const char *f3d_version(void) { return "forge3d_amd 0.3.0 (gfx950 terrain path tracer)"; }
uint32_t f3d_abi_version(void) { return F3D_ABI_VERSION; }
void f3d_device_pool_trim(void) {
    f3d::workspace_trim();  // (first: its buffers go to the pool, which is emptied next)
    f3d::pool_trim();
}

#ifndef F3D_SOURCE_DIGEST
#define F3D_SOURCE_DIGEST "unknown"
#endif
// SHA-256 (first 16 hex digits) of the sources and compiler flags this library was built from: __graft_entry__.build_hip
// passes it in, forge3d_amd/_native.py compares it with the sources next to the library, so "what ran" is "what is in the tree".
const char *f3d_source_digest(void) { return F3D_SOURCE_DIGEST; }

// Diagnostics: pattern 0..255 = every device buffer allocated from now on lies between two guard regions and all of it
// is filled with that byte (f3d_devmem.h); negative = off.  Results must not depend on it.
void f3d_debug_poison(int32_t pattern) { f3d::g_poison_pattern.store(pattern < 0 ? -1 : (pattern & 0xFF)); }

}  // extern "C"
