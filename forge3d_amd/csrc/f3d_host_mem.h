// forge3d_amd/csrc/f3d_host_mem.h -- a FRAGMENT of f3d_host.hip (included there, once, after its helpers): the device
// allocator's poison mode and pool (f3d_devmem.h), the memory ledger of a session, the acceleration tables and their GPU
// build, the staged upload, the set-up clock and the scene cache that keeps the tables of recent DEMs on the device.
// Split out of f3d_host.hip in round 4 (that file had grown to 2 000 lines); one translation unit as before.
#pragma once

// ---- poison mode of the device allocator (f3d_devmem.h) ----
namespace f3d {
namespace {
int poison_from_env() {  // F3D_POISON=<0..255>: a whole process (e.g. the GPU test suite) in poison mode
    const char *v = getenv("F3D_POISON");
    return (v && *v) ? (atoi(v) & 0xFF) : -1;
}
std::atomic<int> g_poison_pattern{poison_from_env()};
std::mutex g_poison_mutex;
std::unordered_map<void *, void *> g_poison_bases;
}  // namespace
int poison_pattern() { return g_poison_pattern.load(); }

// ---- device memory pool (f3d_devmem.h) ----
namespace {
struct PoolBlock {
    void *p;
    size_t bytes;
    int device;
};
std::mutex g_pool_mutex;
std::vector<PoolBlock> &g_pool_free = *new std::vector<PoolBlock>();          // waiting to be handed out again
std::unordered_map<void *, PoolBlock> &g_pool_live = *new std::unordered_map<void *, PoolBlock>();  // handed out: size and device by address
size_t g_pool_bytes = 0;
size_t pool_limit() {
    static const size_t limit = [] {
        const char *e = getenv("F3D_DEVICE_POOL_MB");
        return (size_t)(e ? std::max(0.0, atof(e)) : 1024.0) << 20;
    }();
    return limit;
}
}  // namespace
hipError_t pool_take(void **out, size_t bytes) {
    if (pool_limit() == 0) return hipErrorOutOfMemory;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (size_t i = g_pool_free.size(); i-- > 0;)
        if (g_pool_free[i].bytes == bytes && g_pool_free[i].device == device) {
            *out = g_pool_free[i].p;
            g_pool_live[*out] = g_pool_free[i];
            g_pool_bytes -= bytes;
            g_pool_free.erase(g_pool_free.begin() + (long)i);
            return hipSuccess;
        }
    return hipErrorOutOfMemory;
}
void pool_note(void *p, size_t bytes) {
    if (pool_limit() == 0) return;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pool_live[p] = PoolBlock{p, bytes, device};
}
bool pool_give(void *p) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    auto it = g_pool_live.find(p);
    if (it == g_pool_live.end()) return false;
    const PoolBlock b = it->second;
    g_pool_live.erase(it);
    if (b.bytes > pool_limit() || g_pool_free.size() >= 256u) return false;
    {
        // a block that goes back to the pool must not be handed out while a kernel of an abandoned call (an error path)
        // still writes to it: wait for the BLOCK's device (hipFree, the other way out, waits by itself)
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (prev != b.device) (void)hipSetDevice(b.device);
        (void)hipDeviceSynchronize();
        if (prev >= 0 && prev != b.device) (void)hipSetDevice(prev);
    }
    while (g_pool_bytes + b.bytes > pool_limit() && !g_pool_free.empty()) {  // make room: the oldest go back to the driver
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(g_pool_free.front().device);
        (void)hipFree(g_pool_free.front().p);
        if (prev >= 0) (void)hipSetDevice(prev);
        g_pool_bytes -= g_pool_free.front().bytes;
        g_pool_free.erase(g_pool_free.begin());
    }
    g_pool_free.push_back(b);
    g_pool_bytes += b.bytes;
    return true;
}
void pool_trim() {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (const PoolBlock &b : g_pool_free) {
        (void)hipSetDevice(b.device);
        (void)hipFree(b.p);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    g_pool_free.clear();
    g_pool_bytes = 0;
}
void poison_register(void *user, void *base) {
    std::lock_guard<std::mutex> lock(g_poison_mutex);
    g_poison_bases[user] = base;
}
void *poison_take(void *user) {
    std::lock_guard<std::mutex> lock(g_poison_mutex);
    auto it = g_poison_bases.find(user);
    if (it == g_poison_bases.end()) return nullptr;
    void *base = it->second;
    g_poison_bases.erase(it);
    return base;
}
}  // namespace f3d

namespace {

// ---- device memory ledger (the reference's TrackedGpu / global memory tracker) ----
struct Ledger {
    std::vector<void *> owned;
    uint64_t device_bytes = 0;
    uint64_t host_visible_peak = 0;
    void *alloc(size_t bytes, const char *what) {
        void *p = nullptr;
        hip_check(device_alloc(&p, bytes), what);
        owned.push_back(p);
        device_bytes += bytes;
        return p;
    }
    void free(void *p, size_t bytes) {  // give a buffer back before the session ends (build-time scratch)
        for (auto it = owned.begin(); it != owned.end(); ++it)
            if (*it == p) {
                owned.erase(it);
                (void)device_free(p);
                device_bytes -= bytes;
                return;
            }
    }
    void adopt(void *p, size_t bytes) {  // take ownership of a device buffer somebody else allocated
        owned.push_back(p);
        device_bytes += bytes;
    }
    void note_host_visible(uint64_t bytes) {
        if (bytes > host_visible_peak) host_visible_peak = bytes;
    }
    void release() {
        for (void *p : owned) (void)device_free(p);
        owned.clear();
    }
};

// ---- acceleration tables, built on the GPU ----
struct TerrainTables {
    TerrainDev dev{};
    TableLayout layout;
    uint64_t bytes = 0;  // leaf + node tables
    LeafRec *leaves = nullptr;
    NodeRec *nodes = nullptr;
    NodeRec *bands = nullptr;  // row-major (min,max) of every level (the march's table)
};

// keep_nodes: the tiled node table (levels >= 1) is the input of the band tables and of the sorted descent
// kept for the test hook (f3d_terrain_trace_batch modes 0 / 1, f3d_build_minmax_mips); the frame kernel's
// march reads the band tables only, so sessions give the node table back once the bands are built.
TerrainTables build_tables(Ledger &mem, const float *d_heights, uint32_t w, uint32_t h, float exaggeration,
                           hipStream_t stream, bool keep_nodes) {
    TerrainTables t;
    t.layout = table_layout(w, h);
    const TableLayout &L = t.layout;
    t.leaves = (LeafRec *)mem.alloc(L.leaf_count * sizeof(LeafRec), "leaf table");
    t.nodes = (NodeRec *)mem.alloc((L.node_count ? L.node_count : 1) * sizeof(NodeRec), "node table");
    t.bands = (NodeRec *)mem.alloc(L.band_count * sizeof(NodeRec), "band tables");
    t.bytes = L.leaf_count * sizeof(LeafRec) + (L.node_count + L.band_count) * sizeof(NodeRec);
    hip_check(launch_leaf_build(leaf_build_params(L, d_heights, w, h, exaggeration, t.leaves), stream),
              "leaf table build");
    for (uint32_t l = 1; l < L.levels; l++)
        hip_check(launch_level_build(level_build_params(L, l, t.leaves, t.nodes), stream), "node table build");
    for (uint32_t l = 0; l < L.levels; l++)
        hip_check(launch_band_build(band_build_params(L, l, t.leaves, t.nodes, t.bands), stream), "band table build");
#if !defined(F3D_TRAVERSAL_DESCENT)  // (A/B builds of the frame kernel on the sorted descent need the node table)
    if (!keep_nodes) {
        hip_check(hipStreamSynchronize(stream), "table build");
        mem.free(t.nodes, (L.node_count ? L.node_count : 1) * sizeof(NodeRec));
        t.nodes = nullptr;
        t.bytes = L.leaf_count * sizeof(LeafRec) + L.band_count * sizeof(NodeRec);
    }
#endif
    apply_layout(L, t.dev);
    t.dev.leaves = t.leaves;
    t.dev.nodes = t.nodes;
    t.dev.bands = t.bands;
    return t;
}

}  // namespace

// Where session set-up spends its time (f3d_session_setup_ms; bench.py reports it beside the loop it prepares).
enum SetupPhase { kSetupTotal = 0, kSetupValidate, kSetupHash, kSetupUpload, kSetupTables, kSetupScene, kSetupAlloc, kSetupPasses, kSetupPhases };
struct SetupClock {
    double *ms;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    explicit SetupClock(double *out) : ms(out) {}
    void lap(SetupPhase phase) {
        const auto now = std::chrono::steady_clock::now();
        ms[phase] += std::chrono::duration<double, std::milli>(now - last).count();
        ms[kSetupTotal] = std::chrono::duration<double, std::milli>(now - t0).count();
        last = now;
    }
};
static thread_local double *g_setup_ms = nullptr;  // the session being created on this thread (acquire_tables laps into it)

// ---------------------------------------------------------------------------------------
// scene cache: acceleration tables of recently rendered DEMs stay on the device
// ---------------------------------------------------------------------------------------
// A caller rendering a camera path calls the one-shot entry point once per frame with the same DEM; rebuilding
// 123 MB of tables (and uploading 17 MB) every time is wasted work.  Tables are immutable once built, so sessions
// SHARE them: the cache maps (device, DEM bytes, dims, exaggeration) to a reference-counted table set and keeps
// the most recent kSceneCacheEntries of them after their last session has gone.  A per-process cache behind a
// mutex -- the only global state of the library besides the HIP context.
namespace {

struct CachedTables {
    int device = 0;
    uint64_t key = 0, key2 = 0, dem_bytes = 0;  // two independent 64-bit hashes of the DEM bytes
    uint32_t w = 0, h = 0;
    float exaggeration = 0.0f;
    Ledger mem;  // owns leaf + band tables (+ the far-horizon tables)
    TerrainTables tables;
    uint64_t stamp = 0;
    // far-horizon tables of the IBL rays (f3d_cone.h): they depend on the heights AND on the cell spacing, which is not
    // part of the cache key (the band tables do not care), so one table per spacing this DEM has been rendered with
    struct Horizon {
        float spacing_x, spacing_z;
        float *table;
        uint32_t level, bx, bz;
    };
    std::vector<Horizon> horizons;
    ~CachedTables() {
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(device);
        mem.release();
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

std::mutex g_scene_mutex;
// deliberately never destroyed: a static destructor would call hipFree after the HIP runtime has been torn down at
// interpreter exit (the process's memory goes back to the driver anyway)
std::vector<std::shared_ptr<CachedTables>> &g_scene_cache = *new std::vector<std::shared_ptr<CachedTables>>();
uint64_t g_scene_stamp = 0;
size_t g_scene_limit = 2;  // f3d_scene_cache_limit

uint64_t hash_bytes(const void *data, size_t n, uint64_t seed) {  // 8 bytes at a time, multiply-xorshift
    const uint8_t *p = (const uint8_t *)data;
    uint64_t h = seed ^ (n * 0x9E3779B97F4A7C15ull);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t v;
        memcpy(&v, p + i, 8);
        h = (h ^ v) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 32;
    }
    for (; i < n; i++) h = (h ^ p[i]) * 0x100000001B3ull;
    return h ^ (h >> 29);
}

// Tables for this DEM on this device: from the cache, or built now (and cached when the limit allows).
// Host -> device through the library's own pinned staging pair (two 4 MiB buffers a device, allocated on first use):
// a pageable hipMemcpy of the 16.8 MB headline DEM took 7.3 ms the first time a process made one (the runtime sets up its
// staging then) and the copy into pinned memory overlaps the DMA of the chunk before.
// One pair PER DEVICE (round-4 advice: a process that drives several GPUs -- f3d_session_halo_connect supports it -- recorded
// device A's events on device B's stream, which hipEventRecord refuses): events belong to the device that was current when
// they were created, the buffers are pinned portably, and uploads to different devices do not wait for each other.
struct StagingPair {
    std::mutex mutex;
    void *buffer[2] = {nullptr, nullptr};
    hipEvent_t drained[2] = {nullptr, nullptr};
};
StagingPair &staging_for_current_device() {
    static std::mutex map_mutex;
    static std::map<int, std::unique_ptr<StagingPair>> &pairs = *new std::map<int, std::unique_ptr<StagingPair>>();  // (never destroyed: see g_scene_cache)
    int device = 0;
    hip_check(hipGetDevice(&device), "current device");
    std::lock_guard<std::mutex> lock(map_mutex);
    std::unique_ptr<StagingPair> &slot = pairs[device];
    if (!slot) slot.reset(new StagingPair());
    return *slot;
}
void upload_staged(void *dst, const void *src, size_t bytes, hipStream_t stream) {
    constexpr size_t kChunk = 4u << 20;
    StagingPair &pair = staging_for_current_device();
    std::lock_guard<std::mutex> lock(pair.mutex);
    for (int i = 0; i < 2; i++)  // (also for a small first upload: the pair is part of a device's start-up, not of a later render)
        if (!pair.buffer[i]) {
            hip_check(hipHostMalloc(&pair.buffer[i], kChunk, hipHostMallocPortable), "pinned staging buffer");
            hip_check(hipEventCreateWithFlags(&pair.drained[i], hipEventDisableTiming), "staging event");
        }
    if (bytes < (256u << 10)) {  // small: one plain copy
        hip_check(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), "upload");
        return;
    }
    size_t done = 0;
    for (int turn = 0; done < bytes; turn ^= 1) {
        const size_t n = std::min(kChunk, bytes - done);
        hip_check(hipEventSynchronize(pair.drained[turn]), "staging buffer");  // (never recorded: returns at once)
        memcpy(pair.buffer[turn], (const char *)src + done, n);
        hip_check(hipMemcpyAsync((char *)dst + done, pair.buffer[turn], n, hipMemcpyHostToDevice, stream), "upload");
        hip_check(hipEventRecord(pair.drained[turn], stream), "staging event");
        done += n;
    }
    hip_check(hipEventSynchronize(pair.drained[0]), "upload");  // the staging pair is free again, the data is on its way in order
    hip_check(hipEventSynchronize(pair.drained[1]), "upload");
}

std::shared_ptr<CachedTables> acquire_tables(int device, const float *heights, uint32_t w, uint32_t h, float exaggeration,
                                             hipStream_t stream, bool *was_cached, const DemFingerprint *known = nullptr) {
    const uint64_t bytes = (uint64_t)w * h * sizeof(float);
    double none[kSetupPhases] = {};
    SetupClock clock(g_setup_ms ? g_setup_ms : none);
    const DemFingerprint fp = known ? *known : dem_fingerprint(heights, (size_t)w * h);
    const uint64_t key = hash_bytes(&exaggeration, sizeof(float), fp.key ^ ((uint64_t)w << 32 | h));
    const uint64_t key2 = fp.key2 + 0x3c6ef372fe94f82bull * (uint64_t)w;
    clock.lap(kSetupHash);
    {
        std::lock_guard<std::mutex> lock(g_scene_mutex);
        for (auto &e : g_scene_cache)
            if (e->device == device && e->key == key && e->key2 == key2 && e->w == w && e->h == h && e->exaggeration == exaggeration &&
                e->dem_bytes == bytes) {
                e->stamp = ++g_scene_stamp;
                *was_cached = true;
                return e;
            }
    }
    *was_cached = false;
    auto e = std::make_shared<CachedTables>();
    e->device = device;
    e->key = key;
    e->key2 = key2;
    e->dem_bytes = bytes;
    e->w = w;
    e->h = h;
    e->exaggeration = exaggeration;
    float *d_heights = (float *)e->mem.alloc(bytes, "DEM upload");
    upload_staged(d_heights, heights, bytes, stream);  // (in stream order: the table build follows on the same stream)
    clock.lap(kSetupUpload);
    e->tables = build_tables(e->mem, d_heights, w, h, exaggeration, stream, false);
    // the corner records hold every height (x exaggeration): the raw upload is build-time scratch
    hip_check(hipStreamSynchronize(stream), "table build");
    e->mem.free(d_heights, bytes);
    clock.lap(kSetupTables);
    std::lock_guard<std::mutex> lock(g_scene_mutex);
    e->stamp = ++g_scene_stamp;
    if (g_scene_limit > 0) {
        g_scene_cache.push_back(e);
        while (g_scene_cache.size() > g_scene_limit) {  // drop the least recently used (sessions holding it keep it alive)
            size_t oldest = 0;
            for (size_t i = 1; i < g_scene_cache.size(); i++)
                if (g_scene_cache[i]->stamp < g_scene_cache[oldest]->stamp) oldest = i;
            g_scene_cache.erase(g_scene_cache.begin() + (long)oldest);
        }
    }
    return e;
}

// ---- mesh cache: a mesh's device copy and its BVH, shared like the DEM tables (round 6) -------------------------------------
struct CachedMesh {
    int device = 0;
    uint64_t key_v = 0, key_i = 0;
    uint32_t vertex_count = 0, index_count = 0, builder = 0;
    Ledger mem;
    MeshDev dev{};
    uint64_t stamp = 0;
    // the mesh as a second band of a DEM's pyramid (f3d_meshgrid.h), per grid geometry; bands == nullptr: this mesh has none there
    struct Grid {
        uint32_t cell_w = 0, cell_h = 0;
        float origin_x = 0, origin_z = 0, spacing_x = 0, spacing_z = 0;
        const NodeRec *bands = nullptr;
        const uint32_t *cell_start = nullptr;
        const float4 *tris = nullptr;
        float top = 0;
        size_t bytes = 0;
    };
    std::deque<Grid> grids;  // (a deque: sessions keep pointers to its elements)
    ~CachedMesh() {
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(device);
        mem.release();
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
std::vector<std::shared_ptr<CachedMesh>> &g_mesh_cache = *new std::vector<std::shared_ptr<CachedMesh>>();  // (never destroyed: see g_scene_cache)

std::shared_ptr<CachedMesh> acquire_mesh(int device, const float *vertices, uint32_t vertex_count, const uint32_t *indices, uint32_t index_count,
                                         uint32_t builder, hipStream_t stream) {
    const uint64_t key_v = hash_bytes(vertices, (size_t)vertex_count * 3u * sizeof(float), 0x6D657368ull);
    const uint64_t key_i = hash_bytes(indices, (size_t)index_count * sizeof(uint32_t), 0x696E6478ull);
    {
        std::lock_guard<std::mutex> lock(g_scene_mutex);
        for (auto &e : g_mesh_cache)
            if (e->device == device && e->key_v == key_v && e->key_i == key_i && e->vertex_count == vertex_count && e->index_count == index_count &&
                e->builder == builder) {
                e->stamp = ++g_scene_stamp;
                return e;
            }
    }
    auto e = std::make_shared<CachedMesh>();
    e->device = device;
    e->key_v = key_v;
    e->key_i = key_i;
    e->vertex_count = vertex_count;
    e->index_count = index_count;
    e->builder = builder;
    const std::vector<float> v4 = pad_rgb_to_rgba(vertices, vertex_count, 0.0f);
    float4 *dv = (float4 *)e->mem.alloc(v4.size() * sizeof(float), "mesh vertices");
    uint32_t *di = (uint32_t *)e->mem.alloc((size_t)index_count * sizeof(uint32_t), "mesh indices");
    hip_check(hipMemcpy(dv, v4.data(), v4.size() * sizeof(float), hipMemcpyHostToDevice), "mesh upload");
    hip_check(hipMemcpy(di, indices, (size_t)index_count * sizeof(uint32_t), hipMemcpyHostToDevice), "mesh upload");
    MeshDev &M = e->dev;
    M.vertices = dv;
    M.indices = di;
    M.vertex_count = vertex_count;
    M.index_count = index_count;
    M.traversal_mode = 0u;
    if (builder == 2u) {
        LbvhResult lb;
        hip_check(build_mesh_lbvh(dv, vertex_count, di, index_count, stream, &lb), "GPU LBVH build");
        hip_check(hipStreamSynchronize(stream), "GPU LBVH build");  // other sessions may walk it from their streams
        if (lb.nodes) {
            e->mem.adopt(lb.nodes, lb.node_bytes);
            e->mem.adopt(lb.tris, lb.tri_bytes);
            M.bvh_nodes = lb.nodes;
            M.bvh_tris = lb.tris;
            M.bvh_node_count = lb.node_count;
        }
    } else {
        const MeshBvh bvh = build_mesh_bvh(vertices, vertex_count, indices, index_count);
        if (!bvh.nodes.empty()) {
            // the walk's form: four children wide (one 128-byte record per ENTERED node, f3d_shade.h mesh_bvh4) unless the
            // tree is too deep for the walk's per-level words or the binary form is asked for (3: A/B, the round-3 walk)
            std::vector<Bvh4Node> wide;
            if (builder == 1u) wide = collapse_bvh4(bvh);
            float4 *dt = (float4 *)e->mem.alloc(bvh.tris.size() * sizeof(float), "mesh BVH triangles");
            hip_check(hipMemcpy(dt, bvh.tris.data(), bvh.tris.size() * sizeof(float), hipMemcpyHostToDevice), "BVH upload");
            M.bvh_tris = dt;
            if (!wide.empty()) {
                Bvh4Node *dw = (Bvh4Node *)e->mem.alloc(wide.size() * sizeof(Bvh4Node), "mesh BVH nodes (4-wide)");
                hip_check(hipMemcpy(dw, wide.data(), wide.size() * sizeof(Bvh4Node), hipMemcpyHostToDevice), "BVH upload");
                M.bvh4_nodes = dw;
                M.bvh4_node_count = (uint32_t)wide.size();
            } else {
                BvhNode *dn = (BvhNode *)e->mem.alloc(bvh.nodes.size() * sizeof(BvhNode), "mesh BVH nodes");
                hip_check(hipMemcpy(dn, bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice), "BVH upload");
                M.bvh_nodes = dn;
                M.bvh_node_count = (uint32_t)bvh.nodes.size();
            }
        }
    }
    std::lock_guard<std::mutex> lock(g_scene_mutex);
    e->stamp = ++g_scene_stamp;
    if (g_scene_limit > 0) {
        g_mesh_cache.push_back(e);
        while (g_mesh_cache.size() > g_scene_limit) {  // least recently used out (sessions holding it keep it alive)
            size_t oldest = 0;
            for (size_t i = 1; i < g_mesh_cache.size(); i++)
                if (g_mesh_cache[i]->stamp < g_mesh_cache[oldest]->stamp) oldest = i;
            g_mesh_cache.erase(g_mesh_cache.begin() + (long)oldest);
        }
    }
    return e;
}

// The grid of a cached mesh on this DEM's cells: from the cache entry, or built now (host) and uploaded into its memory.
const CachedMesh::Grid *acquire_mesh_grid(CachedMesh &mesh, const TableLayout &L, const TerrainDev &T, const float *vertices, uint32_t vertex_count,
                                          const uint32_t *indices, uint32_t index_count) {
    std::lock_guard<std::mutex> lock(g_scene_mutex);
    for (const auto &g : mesh.grids)
        if (g.cell_w == T.cell_w && g.cell_h == T.cell_h && g.origin_x == T.origin_x && g.origin_z == T.origin_z && g.spacing_x == T.spacing_x &&
            g.spacing_z == T.spacing_z)
            return &g;
    CachedMesh::Grid out;
    out.cell_w = T.cell_w, out.cell_h = T.cell_h;
    out.origin_x = T.origin_x, out.origin_z = T.origin_z, out.spacing_x = T.spacing_x, out.spacing_z = T.spacing_z;
    const MeshGrid g = build_mesh_grid(L, T.origin_x, T.origin_z, T.spacing_x, T.spacing_z, vertices, vertex_count, indices, index_count);
    if (g.ok) {
        const size_t before = mesh.mem.device_bytes;
        NodeRec *db = (NodeRec *)mesh.mem.alloc(g.bands.size() * sizeof(NodeRec), "mesh band tables");
        uint32_t *dc = (uint32_t *)mesh.mem.alloc(g.cell_start.size() * sizeof(uint32_t), "mesh cell lists");
        float4 *dt = (float4 *)mesh.mem.alloc(std::max<size_t>(g.tris.size(), 4u) * sizeof(float), "mesh cell triangles");
        hip_check(hipMemcpy(db, g.bands.data(), g.bands.size() * sizeof(NodeRec), hipMemcpyHostToDevice), "mesh grid upload");
        hip_check(hipMemcpy(dc, g.cell_start.data(), g.cell_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "mesh grid upload");
        if (!g.tris.empty()) hip_check(hipMemcpy(dt, g.tris.data(), g.tris.size() * sizeof(float), hipMemcpyHostToDevice), "mesh grid upload");
        out.bands = db;
        out.cell_start = dc;
        out.tris = dt;
        out.top = g.top;
        out.bytes = mesh.mem.device_bytes - before;
    }
    mesh.grids.push_back(out);
    return &mesh.grids.back();
}

}  // namespace

namespace f3d {
SharedTerrain acquire_shared_terrain(const float *heights, uint32_t w, uint32_t h, float exaggeration, hipStream_t stream) {
    int device = 0;
    hip_check(hipGetDevice(&device), "hipGetDevice");
    bool was_cached = false;
    std::shared_ptr<CachedTables> e = acquire_tables(device, heights, w, h, exaggeration, stream, &was_cached);
    SharedTerrain out;
    out.dev = TerrainDev{};
    apply_layout(e->tables.layout, out.dev);
    out.dev.leaves = e->tables.leaves;
    out.dev.nodes = e->tables.nodes;
    out.dev.bands = e->tables.bands;
    out.bytes = e->mem.device_bytes;
    out.keep = e;
    return out;
}
}  // namespace f3d

