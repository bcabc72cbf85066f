// tests/emul/f3d_emul.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the product's kernel bodies (forge3d_amd/csrc/f3d_{trace,shade,build}.h, the same
// source the gfx950 kernels are compiled from) on the host, one "lane" at a time, so that
// the kernel logic can be compared bit-for-bit with the CPU oracle in the GPU-less
// container.  It is never shipped, never imported by forge3d_amd, and proves nothing about
// the GPU build by itself -- the `-m gpu` tests repeat the comparison through libf3dhip.so.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <ucontext.h>
#include <vector>

#include "../../forge3d_amd/csrc/f3d_setup.h"
#include "../../forge3d_amd/csrc/f3d_meshgrid.h"
#include "../../forge3d_amd/csrc/f3d_composite.h"
#include "../../forge3d_amd/csrc/f3d_smoke_sim.h"
#include "../../forge3d_amd/csrc/f3d_shade.h"
#include "../../forge3d_amd/csrc/f3d_wf_host.h"
#include "../../forge3d_amd/csrc/f3d_aether_ref_host.h"
#include "../../forge3d_amd/csrc/f3d_div_known.h"

using namespace f3d;

namespace {
// Per-ray step log for the SIMT-utilisation model (tools/utilisation_model.py): kind
// 2 = primary, 3 = IBL occlusion, 7 = sun shadow; bit k of leaf_mask = step k was a fat leaf.
struct RayLog {
    uint32_t kind, steps;
    uint64_t leaf_mask;
};
struct ArrayPending {
    static constexpr bool kMesh = true;  // (f3d_lds.h: only the device compiles terrain-only kernels)
    uint32_t w[kMaxLevels];
    std::vector<RayLog> *log = nullptr;
    void note(int kind) {
        if (!log) return;
        if (kind >= 2) {
            log->push_back(RayLog{(uint32_t)kind, 0u, 0ull});
        } else if (kind < 0) {  // band test passed at this step: remember it in bits 8.. of `kind`
            if (!log->empty()) log->back().kind = (log->back().kind & 0xFFu) | (log->back().steps << 8);
        } else if (!log->empty()) {
            RayLog &r = log->back();
            if (kind == 1 && r.steps < 32u) r.leaf_mask |= 1ull << r.steps;
            r.steps++;
        }
    }
    // statistics: a per-ray feature (e.g. the direction's elevation) for scheduling models, kept in leaf_mask's
    // upper half (tools/march_model.py)
    float hinted = NAN;  // a feature supplied by the shading code for the NEXT ray (overrides the march's own)
    void hint(float f) { hinted = f; }
    void feature(float f) {
        if (!log || log->empty()) return;
        if (hinted == hinted) f = hinted;
        hinted = NAN;
        uint32_t bits;
        memcpy(&bits, &f, 4);
        log->back().leaf_mask = (log->back().leaf_mask & 0xFFFFFFFFull) | ((uint64_t)bits << 32);
    }
    bool leaf_gate(bool) const { return true; }  // one lane at a time: the gate is always open
    // ray sharing needs other lanes: never offered here
    uint32_t bvh4_stack[kBvh4MaxLevels];  // per-level words of the 4-wide mesh walk (f3d_shade.h mesh_bvh4)
    void stack_put(uint32_t level, uint32_t word) { bvh4_stack[level] = word; }
    uint32_t stack_get(uint32_t level) const { return bvh4_stack[level]; }
    uint32_t lane() const { return 0u; }
    bool share_now(bool, uint32_t = 0u) const { return false; }
    template <bool CURVED>
    void deal(const TerrainDev &, MarchSlice &, MarchState &) const {}
    void verdict_post(bool) const {}
    void verdict_set(uint32_t) const {}
    bool verdict_get(uint32_t) const { return false; }
    static constexpr bool kShareClosest = false;
    void nearest_post() const {}
    void nearest_set(uint32_t, uint32_t) const {}
    uint32_t nearest_get(uint32_t) const { return kNoNearest; }
    // deferred leaf FIFO of the march: a single lane drains only when its FIFO is full or its march
    // has ended, i.e. as LATE as possible -- the opposite extreme of the device's wave vote, so the
    // order-independence the deferral relies on is exercised by every CPU parity test
    uint32_t q_cell[kLeafFifoRows];
    float q_lo[kLeafFifoRows], q_hi[kLeafFifoRows];
    void fifo_retag(uint32_t k, uint32_t cell) { q_cell[k] = cell; }
    void fifo_put(uint32_t k, uint32_t cell, float lo, float hi) {
        q_cell[k] = cell;
        q_lo[k] = lo;
        q_hi[k] = hi;
    }
    void fifo_get(uint32_t k, uint32_t &cell, float &lo, float &hi) const {
        cell = q_cell[k];
        // one-word entries (f3d_march.h kFifoWords): the drain must form the interval itself -- poison what it is handed
        lo = kFifoWords == 3u ? q_lo[k] : f_from_bits(0x7fc00000u);
        hi = kFifoWords == 3u ? q_hi[k] : f_from_bits(0x7fc00000u);
    }
    bool flush_now(uint32_t queued, bool marching) const { return queued >= kLeafFifo || (!marching && queued != 0u); }
    bool any(bool pred) const { return pred; }
    unsigned long long ballot(bool pred) const { return pred ? 1ull : 0ull; }  // (march_stream: a wave of one lane)
    void put(uint32_t l, uint32_t v) { w[l] = v; }
    uint32_t get(uint32_t l) const { return w[l]; }
    void band_entry(const TerrainDev &T, uint32_t l, uint32_t &offset, uint32_t &shift) const {
        offset = T.band_offset[l];
        shift = T.band_shift[l];
    }
    void level_entry(const TerrainDev &T, uint32_t l, uint32_t &offset, uint32_t &tiles_x) const {
        offset = T.node_offset[l];
        tiles_x = T.tiles_x[l];
    }
};

// ---- a 64-lane wave on the host -----------------------------------------------------------------------
// The march's ray sharing (f3d_march.h march_shared / march_deal) is wave-cooperative code: ballots, cross-lane
// shuffles, a verdict board shared by the lanes.  To run it in the GPU-less container every lane of a wave is a
// FIBER (ucontext): a lane runs until it reaches a wave primitive, parks there, and when every lane that is still
// alive has parked at the same kind of primitive the exchange happens and they all continue -- the lockstep the
// device gives for free.  Lanes that return leave the wave, like lanes whose EXEC bit has gone.  Divergent regions
// that contain primitives must be entered through Wave::run with the participating lanes (the kernels' own
// structure: everybody calls march_ray together).
struct Wave {
    static constexpr int kLanes = 64;
    static constexpr size_t kStack = 256 * 1024;
    enum Op { kNone, kBallot, kShfl };
    ucontext_t sched{}, ctx[kLanes]{};
    std::vector<char> stacks;
    bool alive[kLanes]{}, parked[kLanes]{};
    Op op[kLanes]{};
    uint32_t val[kLanes]{};
    int src[kLanes]{};
    uint64_t ballot_result = 0;
    uint32_t shfl_result[kLanes]{};
    uint32_t board[kLanes]{};  // verdict board of the ray sharing
    int current = -1;
    std::function<void(int)> body;
    uint64_t exchanges = 0, deals = 0;

    Wave() : stacks(kLanes * kStack) {}
    static void trampoline(unsigned lo, unsigned hi) {
        Wave *w = (Wave *)(((uintptr_t)hi << 32) | (uintptr_t)lo);
        const int lane = w->current;
        w->body(lane);
        w->alive[lane] = false;
        swapcontext(&w->ctx[lane], &w->sched);
    }
    void park(int lane) {
        parked[lane] = true;
        swapcontext(&ctx[lane], &sched);
    }
    uint64_t ballot(int lane, bool pred) {
        op[lane] = kBallot;
        val[lane] = pred ? 1u : 0u;
        park(lane);
        return ballot_result;
    }
    uint32_t shfl(int lane, uint32_t v, int from) {
        op[lane] = kShfl;
        val[lane] = v;
        src[lane] = from;
        park(lane);
        return shfl_result[lane];
    }
    // run body(lane) for the lanes of `mask` as one wave
    void run(uint64_t mask, std::function<void(int)> fn) {
        body = std::move(fn);
        for (int l = 0; l < kLanes; l++) {
            alive[l] = (mask >> l) & 1ull;
            parked[l] = false;
            if (!alive[l]) continue;
            getcontext(&ctx[l]);
            ctx[l].uc_stack.ss_sp = stacks.data() + (size_t)l * kStack;
            ctx[l].uc_stack.ss_size = kStack;
            ctx[l].uc_link = &sched;
            const uintptr_t self = (uintptr_t)this;
            makecontext(&ctx[l], (void (*)())trampoline, 2, (unsigned)(self & 0xFFFFFFFFu), (unsigned)(self >> 32));
        }
        for (;;) {
            bool any = false;
            for (int l = 0; l < kLanes; l++) {
                if (!alive[l] || parked[l]) continue;
                current = l;
                swapcontext(&sched, &ctx[l]);  // until it parks at a primitive or returns
            }
            Op kind = kNone;
            for (int l = 0; l < kLanes; l++) {
                if (!alive[l]) continue;
                any = true;
                if (kind == kNone) kind = op[l];
                if (op[l] != kind) {
                    fprintf(stderr, "wave emulator: lanes parked at different primitives\n");
                    abort();
                }
            }
            if (!any) return;
            exchanges++;
            if (kind == kBallot) {
                ballot_result = 0;
                for (int l = 0; l < kLanes; l++)
                    if (alive[l] && val[l]) ballot_result |= 1ull << l;
            } else {
                for (int l = 0; l < kLanes; l++) {
                    if (!alive[l]) continue;
                    const int from = src[l] & (kLanes - 1);
                    if (!alive[from]) {  // the device would read garbage: the product code must never do this
                        fprintf(stderr, "wave emulator: lane %d shuffles from dead lane %d\n", l, from);
                        abort();
                    }
                    shfl_result[l] = val[from];
                }
            }
            for (int l = 0; l < kLanes; l++) parked[l] = false;
        }
    }
};

// Per-lane traversal context of a wave lane: FIFO in lane-private storage, votes through the wave.
struct WavePending : ArrayPending {
    Wave *wave = nullptr;
    int me = 0;
    uint32_t share_below = kShareBelow;
    uint32_t lane() const { return (uint32_t)me; }
    unsigned long long ballot(bool pred) const { return wave->ballot(me, pred); }
    float shfl(float v, int from) const { return f_from_bits(wave->shfl(me, f_bits(v), from)); }
    uint32_t shfl(uint32_t v, int from) const { return wave->shfl(me, v, from); }
    float fast_log2(float x) const { return log2f(x); }
    float fast_exp2(float x) const { return exp2f(x); }
    bool any(bool pred) const { return ballot(pred) != 0ull; }
    bool flush_now(uint32_t queued, bool marching) const {  // LdsPending::flush_now with the default quorum (64)
        const unsigned long long have = ballot(queued != 0u);
        if (have == 0ull) return false;
        return bits_set(have) >= 64u || ballot(queued >= kLeafFifo) != 0ull || ballot(marching) == 0ull;
    }
    bool share_now(bool marching) const { return share_now(marching, share_below); }
    bool share_now(bool marching, uint32_t below) const {
        const uint32_t n = bits_set(ballot(marching));
        return n != 0u && n <= below && bits_set(ballot(true)) >= kShareAvail * n;
    }
    template <bool CURVED>
    void deal(const TerrainDev &T, MarchSlice &s, MarchState &m) const {
        wave->deals++;
        march_deal<CURVED>(T, s, m, *this);
    }
    void verdict_post(bool hit) const { wave->board[me] = hit ? 1u : 0u; }
    void verdict_set(uint32_t owner) const { wave->board[owner & 63u] = 1u; }
    bool verdict_get(uint32_t owner) const { return wave->board[owner & 63u] != 0u; }
    // closest-hit rays shared as well (f3d_march.h march_shared_closest): the board holds the smallest key posted (0xFFFFFFFF: none)
    static constexpr bool kShareClosest = true;
    void nearest_post() const { wave->board[me] = kNoNearest; }
    void nearest_set(uint32_t owner, uint32_t key) const {
        if (key < wave->board[owner & 63u]) wave->board[owner & 63u] = key;
    }
    uint32_t nearest_get(uint32_t owner) const { return wave->board[owner & 63u]; }
};

struct HostTables {
    TableLayout L;
    std::vector<LeafRec> leaves;
    std::vector<NodeRec> nodes, bands;
    std::vector<float> horizon;  // far-horizon table of the IBL rays (f3d_cone.h), records built on first use (F3D_HORIZON_LAZY)
    void attach(TerrainDev &T) const {
        apply_layout(L, T);
        T.leaves = leaves.data();
        T.nodes = nodes.data();
        T.bands = bands.data();
        T.mesh_bands = bands.data();  // (no mesh grid: the fused march tests the terrain's band twice)
        T.mesh_cell_start = nullptr;
        T.mesh_cell_tris = nullptr;
        T.mesh_top = 0.0f;
    }
    // after attach() and fill_uniforms(): the scene's mesh as a second band of the pyramid (f3d_meshgrid.h), as the product's sessions build it
    MeshGrid grid;
    void attach_mesh_grid(TerrainDev &T, const float *vertices, uint32_t vertex_count, const uint32_t *indices, uint32_t index_count) {
        if (getenv("F3D_EMUL_NO_MESH_GRID")) return;
        grid = build_mesh_grid(L, T.origin_x, T.origin_z, T.spacing_x, T.spacing_z, vertices, vertex_count, indices, index_count);
        if (getenv("F3D_EMUL_MESH_GRID_VERBOSE"))
            fprintf(stderr, "mesh grid: %s, %u triangles, %zu listed in %u x %u cells\n", grid.ok ? "built" : "none", index_count / 3u, grid.tris.size() / 12u, L.cell_w, L.cell_h);
        if (!grid.ok) return;
        T.mesh_bands = grid.bands.data();
        T.mesh_cell_start = grid.cell_start.data();
        T.mesh_cell_tris = (const float4 *)grid.tris.data();
        T.mesh_top = grid.top;
    }
    // after attach() and fill_uniforms(): the table depends on the spacing too
    void attach_horizon(TerrainDev &T) {
        if (getenv("F3D_EMUL_NO_IBL_STOP")) return;
        T.horizon_level = horizon_block_level(T.cell_w, T.cell_h);
        T.horizon_bx = (T.cell_w + (1u << T.horizon_level) - 1u) >> T.horizon_level;
        const uint32_t bz = (T.cell_h + (1u << T.horizon_level) - 1u) >> T.horizon_level;
        horizon.assign((size_t)T.horizon_bx * bz * kIblSectors, std::nanf(""));
        T.horizon = horizon.data();
    }
};

HostTables build_tables_host(const float *heights, uint32_t w, uint32_t h, float exaggeration) {
    HostTables t;
    t.L = table_layout(w, h);
    t.leaves.resize(t.L.leaf_count);
    t.nodes.resize(t.L.node_count ? t.L.node_count : 1);
    PyramidBuildParams lb = leaf_build_params(t.L, heights, w, h, exaggeration, t.leaves.data());
    for (uint32_t y = 0; y < lb.leaf_dim_y; y++)
        for (uint32_t x = 0; x < lb.leaf_dim_x; x++) leaf_build_at(lb, x, y);
    for (uint32_t l = 1; l < t.L.levels; l++) {
        LevelBuildParams b = level_build_params(t.L, l, t.leaves.data(), t.nodes.data());
        for (uint32_t y = 0; y < b.dst_dim_y; y++)
            for (uint32_t x = 0; x < b.dst_dim_x; x++) level_build_at(b, x, y);
    }
    t.bands.resize(t.L.band_count);
    for (uint32_t l = 0; l < t.L.levels; l++) {
        BandBuildParams b = band_build_params(t.L, l, t.leaves.data(), t.nodes.data(), t.bands.data());
        for (uint32_t z = 0; z < b.height; z++)
            for (uint32_t x = 0; x < b.width; x++) band_build_at(b, x, z);
    }
    return t;
}
}  // namespace

// CPU mirror of the frame kernel's sample-lane form (f3d_kernels.hip frame_lanes<S>): the S
// "lanes" of a pixel are array slots stepped in lockstep -- hit flags predicted from the G-buffer,
// primaries re-traced until every sample started from the right stream state, contributions
// replayed in sample order.  Same helpers (sample_primary / sample_shade / accumulate_sample) as
// the device code, so the CPU parity tests pin the speculation scheme against the oracle.
#if defined(F3D_MESH_STATS_HOST)  // statistics build (F3D_EMUL_CXXFLAGS=-DF3D_MESH_STATS_HOST): tools/experiments/bvh4_order.py
namespace f3d { unsigned long long g_host_mesh_stats[8]; }
extern "C" void emul_mesh_stats(unsigned long long *out, int32_t reset) {
    for (int i = 0; i < 8; i++) {
        out[i] = g_host_mesh_stats[i];
        if (reset) g_host_mesh_stats[i] = 0ull;
    }
}
#endif
static int g_use_bvh = 2;  // 0: the reference's sweep over all triangles (A/B of the BVH itself); 1: the threaded binary walk; 2: four children wide (the product's default)
static uint32_t g_sample_lanes = 1u;
static uint64_t g_retraces = 0;  // primaries traced a second time (statistics for the tests)

template <class Pending>
static float frame_pixel_lanes(const FrameParams &P, uint32_t gx, uint32_t gy, uint32_t S, Pending &pend) {
    const FrameHead h = frame_head(P, gx, gy);
    uint32_t stream = h.rng;
    V3 radiance = V3{0.0f, 0.0f, 0.0f};
    Reservoir cand = empty_reservoir();
    const uint32_t group = (1u << S) - 1u;
    uint64_t retraces = 0;
    for (uint32_t s0 = 0u; s0 < P.spp; s0 += S) {
        const uint32_t n_act = P.spp - s0 < S ? P.spp - s0 : S;
        uint32_t pred = h.centre_hit ? group : 0u;
        uint32_t traced[8];
        PrimaryHit ph[8];
        for (uint32_t j = 0; j < 8u; j++) {
            traced[j] = 0xFFFFFFFFu;
            ph[j].hit.kind = 0u;
            ph[j].rng = 0u;
        }
        for (;;) {
            uint32_t draws[8];
            bool need[8], any = false;
            for (uint32_t j = 0; j < S; j++) {
                draws[j] = 2u * j + 2u * (uint32_t)__builtin_popcount(pred & ((1u << j) - 1u));
                need[j] = j < n_act && draws[j] != traced[j];
                any = any || need[j];
            }
            if (!any) break;
            for (uint32_t j = 0; j < S; j++) {
                if (!need[j]) continue;
                if (traced[j] != 0xFFFFFFFFu) retraces++;
                uint32_t st = stream;
                rng_skip(st, draws[j]);
                ph[j] = sample_primary(P, gx, gy, st, pend);
                traced[j] = draws[j];
            }
            pred = 0u;
            for (uint32_t j = 0; j < n_act; j++)
                if (ph[j].hit.kind != 0u) pred |= 1u << j;
        }
        SampleOut o[8];
        for (uint32_t j = 0; j < n_act; j++) {
            uint32_t rng = ph[j].rng;
            o[j] = sample_shade(P, h, ph[j], rng, pend);
        }
        for (uint32_t k = 0; k < n_act; k++) accumulate_sample(cand, radiance, o[k].a, o[k].b, o[k].target_pdf);
        rng_skip(stream, 2u * n_act + 2u * (uint32_t)__builtin_popcount(pred));
    }
    if (retraces) {
#pragma omp atomic
        g_retraces += retraces;
    }
    return frame_tail(P, gx, gy, cand, radiance);
}

static uint32_t g_frames_in_flight = 0u;  // > 1: the k_trace / k_merge / k_fix structure instead of one fused pass per frame
static uint64_t g_retraced = 0;           // pixel-frames whose sun-direction prediction failed

template <class Pending>
static float frame_pixel_any(const FrameParams &P, uint32_t gx, uint32_t gy, Pending &pend) {
    return g_sample_lanes > 1u ? frame_pixel_lanes(P, gx, gy, g_sample_lanes, pend) : frame_pixel(P, gx, gy, pend);
}

extern "C" {

int emul_trace_batch(const float *heights, uint32_t w, uint32_t h, float origin_x, float origin_z, float spacing_x,
                     float spacing_z, float exaggeration, float inv_two_r_prime, uint32_t curvature_enabled,
                     const float *rays, uint32_t n, int32_t any_hit, int32_t apply_curvature, uint32_t *out_hit,
                     float *out_t, float *out_normal) {
    try {
        HostTables t = build_tables_host(heights, w, h, exaggeration);
        TerrainDev T{};
        t.attach(T);
        T.origin_x = origin_x;
        T.origin_z = origin_z;
        T.spacing_x = spacing_x;
        T.spacing_z = spacing_z;
        T.inv_spacing_x = 1.0f / spacing_x;
        T.inv_spacing_z = 1.0f / spacing_z;
        T.inv_two_r_prime = inv_two_r_prime;
        T.curvature_enabled = curvature_enabled;
#pragma omp parallel for schedule(dynamic, 64)
        for (long i = 0; i < (long)n; i++) {
            const float *r = rays + 8 * (size_t)i;
            ArrayPending pend;
            RayCtx rc = make_ray(T, V3{r[0], r[1], r[2]}, r[3], V3{r[4], r[5], r[6]}, r[7], apply_curvature != 0);
            TraceHit hit;
            const uint32_t slices = ((uint32_t)any_hit >> 4) & 15u, slice_level = ((uint32_t)any_hit >> 8) & 15u;
            if ((any_hit & 3) == 2 && slices > 1u) {
                // the any-hit ray as `slices` geometric parameter slices of its root interval, each started at a
                // node of level `slice_level` located from the position -- what the frame kernel's ray sharing
                // does across lanes (march_shared / deal), here one after the other: the OR is the answer
                float lo, hi;
                march_root_interval(T, rc, lo, hi);
                hit.hit = false;
                hit.t = rc.tmin;
                hit.n = V3{0.0f, 0.0f, 0.0f};
                const bool curved = rc.c2 != 0.0f || rc.has_vertex;
                const float base = f_max(lo, 1e-3f * f_max(hi, 1e-30f));
                const float lg = log2f(f_max(hi, base) / base) / (float)slices;
                for (uint32_t k = 0; k < slices && !(lo > hi); k++) {
                    const float begin = k == 0u ? lo : base * exp2f(lg * (float)k);
                    const float stop = k + 1u == slices ? 3.0e38f : base * exp2f(lg * (float)(k + 1u));
                    MarchState m = march_begin(T, rc, (any_hit & 4) != 0);
                    if (k > 0u) {
                        const uint32_t top = T.mip_count - 1u;
                        m.level = slice_level < top ? slice_level : top;
                        m.t_cur = begin;
                        march_locate(T, rc, begin, m.level, m.nx, m.nz);
                        m.unverified_start = true;
                        m.marching = begin <= hi;
                    }
                    if (m.marching) march_fetch(T, m, pend);  // (what march_terrain_from / march_shared do after placing a lane)
                    uint32_t queued = 0u;
                    TraceHit res;
                    res.hit = false;
                    res.t = rc.tmax;
                    res.n = V3{0.0f, 0.0f, 0.0f};
                    for (;;) {
                        if (m.marching) {
                            if (curved) march_step<true, true>(T, rc, m, queued, pend, true, stop);
                            else march_step<false, true>(T, rc, m, queued, pend, true, stop);
                        }
                        if (pend.flush_now(queued, m.marching)) march_drain(T, rc, true, m, queued, res, pend);
                        if (!pend.any(m.marching || queued != 0u)) break;
                    }
                    if (res.hit) hit.hit = true;
                }
            } else if ((any_hit & 3) >= 2) {  // stackless march: 2 any hit, 3 closest; +4 start in the origin cell
                hit = march_ray(T, rc, (any_hit & 3) == 2, (any_hit & 4) != 0, pend);
            } else {
                hit = trace_terrain(T, rc, any_hit != 0, pend);
            }
            out_hit[i] = hit.hit ? 1u : 0u;
            if (out_t) out_t[i] = hit.t;
            if (out_normal) {
                out_normal[3 * i] = hit.hit ? hit.n.x : 0.0f;
                out_normal[3 * i + 1] = hit.hit ? hit.n.y : 0.0f;
                out_normal[3 * i + 2] = hit.hit ? hit.n.z : 0.0f;
            }
        }
    } catch (const Failure &f) {
        return f.status;
    }
    return 0;
}

// terrain any-hit / closest march over a ray batch in WAVES of 64 lanes with the ray sharing live
// (f3d_kernels.hip k_ray_batch on the host): mode 2 any / 3 closest, +4 start in the origin's cell;
// share_below = the sharing threshold (0 = default).  stats_out[0] = wave exchanges, [1] = rays that were dealt.
int emul_trace_batch_wave(const float *heights, uint32_t w, uint32_t h, float origin_x, float origin_z, float spacing_x,
                          float spacing_z, float exaggeration, float inv_two_r_prime, uint32_t curvature_enabled,
                          const float *rays, uint32_t n, int32_t mode, int32_t apply_curvature, uint32_t share_below,
                          uint32_t *out_hit, float *out_t, float *out_normal, uint64_t *stats_out) {
    try {
        HostTables t = build_tables_host(heights, w, h, exaggeration);
        TerrainDev T{};
        t.attach(T);
        T.origin_x = origin_x;
        T.origin_z = origin_z;
        T.spacing_x = spacing_x;
        T.spacing_z = spacing_z;
        T.inv_spacing_x = 1.0f / spacing_x;
        T.inv_spacing_z = 1.0f / spacing_z;
        T.inv_two_r_prime = inv_two_r_prime;
        T.curvature_enabled = curvature_enabled;
        const long waves = ((long)n + 63) / 64;
        uint64_t exchanges = 0, deals = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : exchanges, deals)
        for (long wv = 0; wv < waves; wv++) {
            Wave wave;
            const uint32_t first = (uint32_t)wv * 64u, count = n - first < 64u ? n - first : 64u;
            const uint64_t mask = count == 64u ? ~0ull : ((1ull << count) - 1ull);
            wave.run(mask, [&](int lane) {
                const uint32_t i = first + (uint32_t)lane;
                const float *r = rays + 8 * (size_t)i;
                WavePending pend;
                pend.wave = &wave;
                pend.me = lane;
                pend.share_below = share_below ? (share_below < 64u ? share_below : 64u) : kShareBelow;
                RayCtx rc = make_ray(T, V3{r[0], r[1], r[2]}, r[3], V3{r[4], r[5], r[6]}, r[7], apply_curvature != 0);
                const TraceHit hit = march_ray(T, rc, (mode & 3) == 2, (mode & 4) != 0, pend);
                out_hit[i] = hit.hit ? 1u : 0u;
                if (out_t) out_t[i] = hit.t;
                if (out_normal) {
                    out_normal[3 * i] = hit.hit ? hit.n.x : 0.0f;
                    out_normal[3 * i + 1] = hit.hit ? hit.n.y : 0.0f;
                    out_normal[3 * i + 2] = hit.hit ? hit.n.z : 0.0f;
                }
            });
            exchanges += wave.exchanges;
            deals += wave.deals;
        }
        if (stats_out) {
            stats_out[0] = exchanges;
            stats_out[1] = deals;  // lane-calls of march_deal
        }
    } catch (const Failure &f) {
        return f.status;
    }
    return 0;
}

// levels_out in the reference layout (see f3d_build_minmax_mips)
int emul_build_minmax_mips(const float *heights, uint32_t w, uint32_t h, float *levels_out, uint32_t *dims_out,
                           uint64_t *total_floats) {
    try {
        HostTables t = build_tables_host(heights, w, h, 1.0f);
        uint64_t off = 0;
        for (uint32_t l = 0; l < t.L.levels; l++) {
            if (dims_out) {
                dims_out[2 * l] = t.L.level_w[l];
                dims_out[2 * l + 1] = t.L.level_h[l];
            }
            for (uint32_t y = 0; y < t.L.level_h[l]; y++)
                for (uint32_t x = 0; x < t.L.level_w[l]; x++) {
                    float mn = INFINITY, mx = -INFINITY;
                    if (l == 0) {
                        if (x < t.L.cell_w && y < t.L.cell_h) {
                            const LeafRec &r = t.leaves[tiled_index(x, y, t.L.tiles_x[0])];
                            mn = min4(r);
                            mx = max4(r);
                        }
                    } else {
                        const NodeRec &r = t.nodes[t.L.node_offset[l] + tiled_index(x, y, t.L.tiles_x[l])];
                        mn = r.mn;
                        mx = r.mx;
                    }
                    if (levels_out) {
                        levels_out[off] = mn;
                        levels_out[off + 1] = mx;
                    }
                    off += 2;
                }
        }
        if (total_floats) *total_floats = off;
        return (int)t.L.levels;
    } catch (const Failure &f) {
        return -f.status;
    }
}

// Full render with the product's per-pixel code; fixed-frame or converging, like the C ABI.
// state dumps: accum (P x 4: rgb + welford mean), m2 (P), res (P x 4 u32 words, final temporal output)
int emul_render(const f3d_terrain_ref_desc *d, uint32_t row_begin, uint32_t row_end, uint8_t *rgba, float *albedo,
                float *normal, float *depth, uint32_t *frames_out, float *variance_out, int32_t *converged_out,
                float *accum_dump, float *m2_dump, uint32_t *res_dump, char *err, size_t errlen) {
    try {
        validate_desc(*d);
        validate_scene(*d);
        FrameParams P{};
        const bool require_valid = fill_uniforms(*d, P);
        HostTables t = build_tables_host(d->heights, d->dem_width, d->dem_height, d->exaggeration);
        t.attach(P.terrain);
        t.attach_horizon(P.terrain);
        std::vector<float> env4, mesh4;
        MeshBvh bvh;
        std::vector<Bvh4Node> bvh4;
        if (d->env_map) {
            env4 = pad_rgb_to_rgba(d->env_map, (size_t)d->env_width * d->env_height, 1.0f);
            P.env.texels = (const float4 *)env4.data();
            P.env.width = d->env_width;
            P.env.height = d->env_height;
        }
        if (d->mesh_vertices) {
            mesh4 = pad_rgb_to_rgba(d->mesh_vertices, d->mesh_vertex_count, 0.0f);
            P.mesh.vertices = (const float4 *)mesh4.data();
            P.mesh.indices = d->mesh_indices;
            P.mesh.vertex_count = d->mesh_vertex_count;
            P.mesh.index_count = d->mesh_index_count;
            P.mesh.traversal_mode = 0u;
            if (g_use_bvh) {
                bvh = build_mesh_bvh(d->mesh_vertices, d->mesh_vertex_count, d->mesh_indices, d->mesh_index_count);
                P.mesh.bvh_nodes = bvh.nodes.data();
                P.mesh.bvh_tris = (const float4 *)bvh.tris.data();
                P.mesh.bvh_node_count = (uint32_t)bvh.nodes.size();
                if (g_use_bvh == 2) bvh4 = collapse_bvh4(bvh);
                if (!bvh4.empty()) {
                    P.mesh.bvh4_nodes = bvh4.data();
                    P.mesh.bvh4_node_count = (uint32_t)bvh4.size();
                }
                if (g_use_bvh == 2) t.attach_mesh_grid(P.terrain, d->mesh_vertices, d->mesh_vertex_count, d->mesh_indices, d->mesh_index_count);
            }
        }
        if (row_end == 0) row_end = d->height;
        P.row_begin = row_begin;
        P.row_end = row_end;
        const uint32_t W = d->width, rows = row_end - row_begin;
        const size_t px = (size_t)rows * W, res_n = (size_t)(rows + 2 * kHaloRows) * W;
        std::vector<PackedReservoir> res[2] = {std::vector<PackedReservoir>(res_n), std::vector<PackedReservoir>(res_n)};
        std::vector<float4> accum(px), gbuf(px);
        std::vector<float> m2(px, 0.0f), dep(px);
        memset(accum.data(), 0, px * sizeof(float4));
        memset(res[0].data(), 0, res_n * sizeof(PackedReservoir));
        memset(res[1].data(), 0, res_n * sizeof(PackedReservoir));
        P.accum_mean = accum.data();
        P.welford_m2 = m2.data();
        P.gbuffer_n = gbuf.data();
        std::vector<uint2> starts(getenv("F3D_EMUL_NO_PRIMARY_START") ? 0 : px);  // f3d_cone.h certificates
        P.primary_start = starts.empty() ? nullptr : starts.data();
        std::vector<float2> sun_clear(getenv("F3D_EMUL_NO_SUN_CLEAR") ? 0 : px);
        P.sun_clear = sun_clear.empty() ? nullptr : sun_clear.data();
#pragma omp parallel for schedule(dynamic, 4)
        for (long y = row_begin; y < (long)row_end; y++) {
            ArrayPending pend;
            for (uint32_t x = 0; x < W; x++) gbuffer_pixel(P, x, (uint32_t)y, gbuf.data(), dep.data(), pend);
        }
        if (getenv("F3D_EMUL_START_STATS") && P.primary_start) {  // how far do the certificates of f3d_cone.h reach?
            size_t none = 0, sky = 0, some = 0;
            double frac = 0.0, lvl = 0.0;
            for (size_t i = 0; i < px; i++) {
                const float t = f_from_bits(starts[i].x);
                if (t == 0.0f) none++;
                else if (t > 1e37f) sky++;
                else {
                    some++;
                    lvl += starts[i].y;
                    if (gbuf[i].w != 0.0f) frac += t / dep[i];
                }
            }
            fprintf(stderr, "certificates: none %zu, whole ray %zu, partial %zu (mean t_clear / depth %.3f, mean level %.2f)\n", none, sky, some,
                    some ? frac / (double)some : 0.0, some ? lvl / (double)some : 0.0);
            if (P.sun_clear) {
                size_t hits = 0, with = 0, sectors = 0;
                double mean_from = 0.0;
                for (size_t i = 0; i < px; i++) {
                    if (gbuf[i].w == 0.0f) continue;
                    hits++;
                    if (sun_clear[i].x < 1e30f) {
                        with++;
                        mean_from += sun_clear[i].x;
                    }
                }
                fprintf(stderr, "sun certificates: %zu of %zu hit pixels (mean clear_from %.1f); IBL sectors with a horizon: %.2f of 8\n", with, hits,
                        with ? mean_from / (double)with : 0.0, hits ? (double)sectors / (double)hits : 0.0);
            }
        }
        uint32_t frames = 0;
        float variance = INFINITY;
        bool converged = false;
        // frames in flight on the host: the same batches (2, 2, 4, 8, ... frames, never across a convergence window), the
        // same prediction flags, the same three passes as f3d_host.hip / f3d_kernels.hip
        const bool in_flight = g_frames_in_flight > 1u;
        std::vector<float4> records;
        std::vector<uint2> head_flags(in_flight ? px : 0);
        uint32_t batch_end = 0u;
        if (in_flight) {
            P.same_sun = (f_bits(P.light.wi.x) == f_bits(P.light.wi_reuse.x) && f_bits(P.light.wi.y) == f_bits(P.light.wi_reuse.y) &&
                          f_bits(P.light.wi.z) == f_bits(P.light.wi_reuse.z)) ? 1u : 0u;
            if (getenv("F3D_EMUL_FORCE_PREDICTION")) P.same_sun = 0u;  // exercise prediction + re-trace even when the directions agree
            for (size_t lp = 0; lp < px; lp++) {  // k_trace_init
                const float4 g = gbuf[lp];
                head_flags[lp] = uint2{0u, (g.w != 0.0f && dot(V3{g.x, g.y, g.z}, P.light.wi) > 0.0f) ? kHeadPrevValid : 0u};
            }
            P.head = head_flags.data();
        }
        while (frames < d->max_frames) {
            P.frame_index = frames;
            P.res_out = res[frames & 1u].data();
            P.res_in = res[(frames & 1u) ^ 1u].data();
            float vmax_m2 = 0.0f;
            bool nonfinite = false;
            if (in_flight) {
                if (frames >= batch_end) {  // trace the next batch: every pixel, every frame of the batch
                    uint32_t stop = (frames / kWelfordWindow + 1u) * kWelfordWindow;
                    if (stop > d->max_frames) stop = d->max_frames;
                    uint32_t ramp = 2u;
                    while (ramp < g_frames_in_flight && ramp * 2u <= frames) ramp *= 2u;
                    if (frames < 2u) ramp = 2u - frames;
                    const uint32_t n = std::max(1u, std::min(std::min(g_frames_in_flight, stop - frames), ramp));
                    records.assign((size_t)n * P.spp * px * 2u, float4{0.0f, 0.0f, 0.0f, 0.0f});
                    P.trace = records.data();
                    P.trace_first = frames;
                    batch_end = frames + n;
#pragma omp parallel for schedule(dynamic, 4)
                    for (long y = row_begin; y < (long)row_end; y++) {
                        ArrayPending pend;
                        for (uint32_t x = 0; x < W; x++) {
                            const size_t lp = (size_t)(y - row_begin) * W + x;
                            for (uint32_t f = frames; f < batch_end; f++)
                                trace_pixel(P, f, x, (uint32_t)y, f > 0u && (head_flags[lp].y & kHeadPrevValid) != 0u,
                                            records.data() + 2u * ((size_t)(f - frames) * P.spp * px + lp), px, pend);
                        }
                    }
                }
                std::vector<uint32_t> fix_list;
#pragma omp parallel for schedule(dynamic, 4) reduction(max : vmax_m2) reduction(|| : nonfinite)
                for (long y = row_begin; y < (long)row_end; y++) {  // k_merge
                    for (uint32_t x = 0; x < W; x++) {
                        const size_t lp = (size_t)(y - row_begin) * W + x;
                        const float4 *rec = records.data() + 2u * ((size_t)(frames - P.trace_first) * P.spp * px + lp);
                        const FrameHead h = frame_head<true>(P, x, (uint32_t)y);
                        bool redo = false;
                        if (P.same_sun == 0u) {
                            redo = merge_mispredicted(P, h, rec, px);
                            if (getenv("F3D_EMUL_FORCE_PREDICTION") && (lp % 97u) == (frames % 97u)) redo = true;  // and some re-traces for no reason
                            if (frames > 0u) head_flags[lp].y = h.prev_valid ? kHeadPrevValid : 0u;
                        }
                        if (redo) {
#pragma omp critical
                            fix_list.push_back((uint32_t)lp);
                        } else {
                            const float v = merge_pixel(P, x, (uint32_t)y, h, rec, px);
                            if (!f_finite(v)) nonfinite = true;
                            else vmax_m2 = f_max(vmax_m2, f_max(v, 0.0f));
                        }
                    }
                }
                g_retraced += fix_list.size();
                for (uint32_t lp : fix_list) {  // k_fix
                    ArrayPending pend;
                    const float v = fix_pixel(P, lp % W, row_begin + lp / W, records.data() + 2u * ((size_t)(frames - P.trace_first) * P.spp * px + lp), px, pend);
                    if (!f_finite(v)) nonfinite = true;
                    else vmax_m2 = f_max(vmax_m2, f_max(v, 0.0f));
                }
            }
            // F3D_EMUL_RAYLOG=<file>: dump the per-ray step log of the LAST frame
            const char *log_path = getenv("F3D_EMUL_RAYLOG");
            const bool logging = log_path && log_path[0] && frames + 1 == d->max_frames;
            std::vector<std::vector<RayLog>> pixel_logs(logging ? px : 0);
#pragma omp parallel for schedule(dynamic, 4) reduction(max : vmax_m2) reduction(|| : nonfinite)
            for (long y = row_begin; y < (long)(in_flight ? row_begin : row_end); y++) {
                ArrayPending pend;
                for (uint32_t x = 0; x < W; x++) {
                    if (logging) pend.log = &pixel_logs[(size_t)(y - row_begin) * W + x];
                    const float v = frame_pixel_any(P, x, (uint32_t)y, pend);
                    if (!f_finite(v)) nonfinite = true;
                    else vmax_m2 = f_max(vmax_m2, f_max(v, 0.0f));
                }
            }
            if (logging) {
                if (FILE *f = fopen(log_path, "wb")) {
                    const uint32_t hdr[4] = {W, rows, P.spp, 0u};
                    fwrite(hdr, 4, 4, f);
                    for (const auto &pl : pixel_logs) {
                        const uint32_t n = (uint32_t)pl.size();
                        fwrite(&n, 4, 1, f);
                        if (n) fwrite(pl.data(), sizeof(RayLog), n, f);
                    }
                    fclose(f);
                }
            }
            frames++;
            if (frames % kWelfordWindow == 0u || frames == d->max_frames) {
                const uint32_t n_window = ((frames - 1u) % kWelfordWindow) + 1u;
                if (n_window >= 2u) {
                    if (nonfinite) fail(F3D_STATUS_RENDER, "terrain PT produced non-finite variance (NaN in accumulation)");
                    variance = f_max(0.0f, vmax_m2 / ((float)n_window - 1.0f));
                    if (frames >= d->min_frames && variance < d->variance_threshold) {
                        converged = true;
                        break;
                    }
                }
            }
        }
        *frames_out = frames;
        *variance_out = variance;
        *converged_out = converged ? 1 : 0;
        if (!converged) fail(F3D_STATUS_RENDER, "terrain PT did not converge");
        P.res_in = res[(frames - 1u) & 1u].data();
        uint32_t flags_or = 0;
        for (uint32_t y = row_begin; y < row_end; y++)
            for (uint32_t x = 0; x < W; x++) flags_or |= resolve_pixel(P, frames, x, y, rgba, albedo, normal);
        memcpy(depth, dep.data(), px * sizeof(float));
        if (flags_or & 2u) fail(F3D_STATUS_RENDER, "terrain PT reservoir bookkeeping produced non-finite values");
        if (require_valid && !(flags_or & 1u)) fail(F3D_STATUS_RENDER, "no valid reservoirs");
        if (accum_dump) memcpy(accum_dump, accum.data(), px * sizeof(float4));
        if (m2_dump) memcpy(m2_dump, m2.data(), px * sizeof(float));
        if (res_dump) memcpy(res_dump, res[(frames - 1u) & 1u].data() + (size_t)kHaloRows * W, px * sizeof(PackedReservoir));
    } catch (const Failure &f) {
        if (err && errlen) snprintf(err, errlen, "%s", f.message.c_str());
        return f.status;
    }
    return 0;
}

// ---- strip session on the host: the subset of f3d_session_* the row-strip driver
// (forge3d_amd/distributed.py) needs, so its halo exchange / all-reduce / gather logic can be
// tested with world_size-2 gloo on the CPU.  Reservoir buffers are caller-owned.
struct EmulSession {
    FrameParams P{};
    HostTables tables;
    std::vector<float> env4, mesh4;
    std::vector<uint32_t> mesh_idx;
    MeshBvh bvh;
    std::vector<Bvh4Node> bvh4;
    std::vector<float4> accum, gbuf;
    std::vector<uint2> starts;
    std::vector<float2> sun_clear;
    std::vector<float> m2, depth;
    PackedReservoir *res[2] = {nullptr, nullptr};
    uint32_t rows = 0, width = 0;
    bool require_valid = false;
};

void *emul_session_create(const f3d_terrain_ref_desc *d, uint32_t row_begin, uint32_t row_end, void *res0, void *res1,
                          char *err, size_t errlen) {
    EmulSession *s = new EmulSession();
    try {
        validate_desc(*d);
        validate_scene(*d);
        s->require_valid = fill_uniforms(*d, s->P);
        s->tables = build_tables_host(d->heights, d->dem_width, d->dem_height, d->exaggeration);
        s->tables.attach(s->P.terrain);
        s->tables.attach_horizon(s->P.terrain);
        if (d->env_map) {
            s->env4 = pad_rgb_to_rgba(d->env_map, (size_t)d->env_width * d->env_height, 1.0f);
            s->P.env.texels = (const float4 *)s->env4.data();
            s->P.env.width = d->env_width;
            s->P.env.height = d->env_height;
        }
        if (d->mesh_vertices) {
            s->mesh4 = pad_rgb_to_rgba(d->mesh_vertices, d->mesh_vertex_count, 0.0f);
            s->mesh_idx.assign(d->mesh_indices, d->mesh_indices + d->mesh_index_count);
            s->P.mesh.vertices = (const float4 *)s->mesh4.data();
            s->P.mesh.indices = s->mesh_idx.data();
            s->P.mesh.vertex_count = d->mesh_vertex_count;
            s->P.mesh.index_count = d->mesh_index_count;
            s->P.mesh.traversal_mode = 0u;
            if (g_use_bvh) {
                s->bvh = build_mesh_bvh(d->mesh_vertices, d->mesh_vertex_count, d->mesh_indices, d->mesh_index_count);
                s->P.mesh.bvh_nodes = s->bvh.nodes.data();
                s->P.mesh.bvh_tris = (const float4 *)s->bvh.tris.data();
                s->P.mesh.bvh_node_count = (uint32_t)s->bvh.nodes.size();
                if (g_use_bvh == 2) s->bvh4 = collapse_bvh4(s->bvh);
                if (!s->bvh4.empty()) {
                    s->P.mesh.bvh4_nodes = s->bvh4.data();
                    s->P.mesh.bvh4_node_count = (uint32_t)s->bvh4.size();
                }
                if (g_use_bvh == 2)
                    s->tables.attach_mesh_grid(s->P.terrain, d->mesh_vertices, d->mesh_vertex_count, d->mesh_indices, d->mesh_index_count);
            }
        }
        if (row_end == 0) row_end = d->height;
        s->P.row_begin = row_begin;
        s->P.row_end = row_end;
        s->rows = row_end - row_begin;
        s->width = d->width;
        const size_t px = (size_t)s->rows * s->width;
        s->accum.assign(px, float4{0, 0, 0, 0});
        s->gbuf.resize(px);
        s->m2.assign(px, 0.0f);
        s->depth.resize(px);
        s->res[0] = (PackedReservoir *)res0;
        s->res[1] = (PackedReservoir *)res1;
        s->P.accum_mean = s->accum.data();
        s->P.welford_m2 = s->m2.data();
        s->P.gbuffer_n = s->gbuf.data();
        s->starts.assign(px, uint2{0u, 0u});
        s->P.primary_start = s->starts.data();
        s->sun_clear.assign(px, float2{3.0e38f, 0.0f});
        s->P.sun_clear = s->sun_clear.data();
        for (uint32_t y = row_begin; y < row_end; y++) {
            ArrayPending pend;
            for (uint32_t x = 0; x < s->width; x++) gbuffer_pixel(s->P, x, y, s->gbuf.data(), s->depth.data(), pend);
        }
    } catch (const Failure &f) {
        if (err && errlen) snprintf(err, errlen, "%s", f.message.c_str());
        delete s;
        return nullptr;
    }
    return s;
}

// one frame (part 0), or its edge rows (part 1: the first and last kHaloRows pixel rows, what a neighbouring
// strip needs as halo) / the interior (part 2, after part 1); stats[0] = max m2 bits, stats[1] = nonfinite
// (same record as the device writes; part 2 merges into what part 1 left)
void emul_session_frame_part(void *h, uint32_t frame, uint32_t part, int32_t collect, uint32_t *stats) {
    EmulSession *s = (EmulSession *)h;
    FrameParams &P = s->P;
    P.frame_index = frame;
    P.res_out = s->res[frame & 1u];
    P.res_in = s->res[(frame & 1u) ^ 1u];
    const uint32_t rows = P.row_end - P.row_begin;
    float vmax = 0.0f;
    bool bad = false;
    for (uint32_t y = P.row_begin; y < P.row_end; y++) {
        const uint32_t r = y - P.row_begin;
        const bool edge = rows <= 2u * kHaloRows || r < kHaloRows || r + kHaloRows >= rows;
        if ((part == 1u && !edge) || (part == 2u && edge)) continue;
        ArrayPending pend;
        for (uint32_t x = 0; x < s->width; x++) {
            const float v = frame_pixel_any(P, x, y, pend);
            if (!f_finite(v)) bad = true;
            else vmax = f_max(vmax, f_max(v, 0.0f));
        }
    }
    if (collect && stats) {
        if (part == 2u) {
            stats[0] = f_bits(f_max(f_from_bits(stats[0]), vmax));
            stats[1] |= bad ? 1u : 0u;
        } else {
            stats[0] = f_bits(vmax);
            stats[1] = bad ? 1u : 0u;
        }
    }
}

void emul_session_frame(void *h, uint32_t frame, int32_t collect, uint32_t *stats) {
    emul_session_frame_part(h, frame, 0u, collect, stats);
}

int emul_session_resolve(void *h, uint32_t frames, uint8_t *rgba, float *albedo, float *normal, float *depth,
                         int32_t *any_valid) {
    EmulSession *s = (EmulSession *)h;
    FrameParams P = s->P;
    P.res_in = s->res[(frames - 1u) & 1u];
    uint32_t flags = 0;
    for (uint32_t y = P.row_begin; y < P.row_end; y++)
        for (uint32_t x = 0; x < s->width; x++) flags |= resolve_pixel(P, frames, x, y, rgba, albedo, normal);
    memcpy(depth, s->depth.data(), s->depth.size() * sizeof(float));
    if (any_valid) *any_valid = (flags & 1u) != 0u;
    return (flags & 2u) ? F3D_STATUS_RENDER : 0;
}

void emul_session_destroy(void *h) { delete (EmulSession *)h; }

// debugging aid: the far-horizon record of the block that holds cell (cx, cz): out[0..7] slopes, [8] rho, [9] stop distance,
// [10] block level, [11..12] block centre x z, [13] the height no origin of the block lies below, [14..15] block size x z
int emul_horizon_blocks(const f3d_terrain_ref_desc *d, uint32_t n, const uint32_t *cells, float *out_all) {
    try {
        FrameParams P{};
        (void)fill_uniforms(*d, P);
        HostTables t = build_tables_host(d->heights, d->dem_width, d->dem_height, d->exaggeration);
        t.attach(P.terrain);
        const TerrainDev &T = P.terrain;
        const uint32_t level = horizon_block_level(T.cell_w, T.cell_h);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t cx = cells[2 * i], cz = cells[2 * i + 1];
            float *out = out_all + 16 * (size_t)i;
            if (cx >= T.cell_w || cz >= T.cell_h) return 1;
            horizon_block_build(T, level, cx >> level, cz >> level, out);
            out[8] = horizon_block_rho(T, level);
            out[9] = ibl_stop_distance(out[8], f_max(T.spacing_x, T.spacing_z));
            out[10] = (float)level;
            horizon_block_frame(T, level, cx >> level, cz >> level, out[11], out[12], out[13]);
            out[14] = T.spacing_x * (float)(1u << level);
            out[15] = T.spacing_z * (float)(1u << level);
        }
        return 0;
    } catch (const Failure &) {
        return 1;
    }
}

// The claim behind k_head's shortcut for tiles whose neighbourhood holds no reservoir sample (csrc/f3d_frame.h
// head_neighbourhood_empty): for a pixel whose [-3, +reach_hi]^2 neighbourhood (clamped to the image) has m == 0 everywhere, frame_head
// parks {0, the pixel's own light-type bit, 0, its own target pdf} and returns "no usable history" -- whatever else the
// records hold.  res: (height + 2 * kHaloRows) rows of `width` packed reservoirs (rows of the halo included, strip = image);
// gbuffer: per pixel {n, hit flag}.  Returns the number of pixels for which the claim applies and frame_head says
// otherwise (0 = the claim holds); *applies = how many pixels it applied to.  reach_hi = kSpatialReachHi (4) is the window
// the kernel uses; 3 is round 4's window, kept callable so that the test can show the constructed u == 1.0 case break it.
uint32_t emul_head_shortcut_mismatches(uint32_t width, uint32_t height, uint32_t frame, const void *res, const float *gbuffer,
                                       const float *wi_reuse, uint32_t seed_hi, uint32_t seed_lo, uint32_t *applies, uint32_t reach_hi) {
    FrameParams P{};
    P.cam.width = width;
    P.cam.height = height;
    P.cam.seed_hi = seed_hi;
    P.cam.seed_lo = seed_lo;
    P.row_begin = 0;
    P.row_end = height;
    P.frame_index = frame;
    P.light.wi_reuse = V3{wi_reuse[0], wi_reuse[1], wi_reuse[2]};
    P.gbuffer_n = reinterpret_cast<const float4 *>(gbuffer);
    const PackedReservoir *in = reinterpret_cast<const PackedReservoir *>(res);
    std::vector<PackedReservoir> out((size_t)(height + 2u * kHaloRows) * width);
    P.res_in = in;
    P.res_out = out.data();
    uint32_t bad = 0u, n = 0u;
    for (uint32_t gy = 0; gy < height; gy++)
        for (uint32_t gx = 0; gx < width; gx++) {
            bool empty = true;
            for (int dy = -(int)kSpatialReachLo; dy <= (int)reach_hi && empty; dy++)
                for (int dx = -(int)kSpatialReachLo; dx <= (int)reach_hi; dx++) {
                    const int qx = std::min(std::max((int)gx + dx, 0), (int)width - 1), qy = std::min(std::max((int)gy + dy, 0), (int)height - 1);
                    if ((in[reservoir_index(P, (uint32_t)qx, (uint32_t)qy)].m_lt & ~kLightTypeBit) != 0u) {
                        empty = false;
                        break;
                    }
                }
            if (!empty) continue;
            n++;
            const FrameHead h = frame_head(P, gx, gy);
            const size_t ri = reservoir_index(P, gx, gy);
            const PackedReservoir self = in[ri], parked = out[ri];
            const uint2 rec = pack_head(h);
            const bool centre_hit = gbuffer[4u * ((size_t)gy * width + gx) + 3u] != 0.0f;
            const bool same = f_bits(parked.w_sum) == 0u && parked.m_lt == self.m_lt && f_bits(parked.weight) == 0u &&
                              f_bits(parked.target_pdf) == f_bits(self.target_pdf) && rec.x == f_bits(1.0f) &&
                              rec.y == (centre_hit ? kHeadCentreHit : 0u);
            if (!same) bad++;
        }
    if (applies) *applies = n;
    return bad;
}

// debugging aid: the sun-ray certificate of one pixel: {clear_from, centre depth, centre origin xyz}
int emul_sun_clear(const f3d_terrain_ref_desc *d, uint32_t gx, uint32_t gy, float *out) {
    try {
        FrameParams P{};
        (void)fill_uniforms(*d, P);
        HostTables t = build_tables_host(d->heights, d->dem_width, d->dem_height, d->exaggeration);
        t.attach(P.terrain);
        P.row_begin = 0;
        P.row_end = d->height;
        ArrayPending pend;
        const V3 rd = camera_dir(P.cam, gx, gy, 0.0f, 0.0f);
        const SurfaceHit hit = closest_hit(P, P.cam.origin, 1e-3f, rd, 1e30f, pend);
        out[0] = 3.0e38f;
        out[1] = 0.0f;
        if (hit.kind != 0u) {
            const V3 o = along(hit.p, 1e-3f, hit.n);
            out[0] = sun_clear_from(P, o, hit.t);
            out[1] = hit.t;
            out[2] = o.x;
            out[3] = o.y;
            out[4] = o.z;
            out[5] = P.light.wi.x;
            out[6] = P.light.wi.y;
            out[7] = P.light.wi.z;
        }
        return 0;
    } catch (const Failure &) {
        return 1;
    }
}

// debugging aid: the certificate of one pixel (build with -DF3D_CONE_DEBUG for the walk)
int emul_primary_start(const f3d_terrain_ref_desc *d, uint32_t gx, uint32_t gy, float *t_clear, uint32_t *level) {
    try {
        FrameParams P{};
        (void)fill_uniforms(*d, P);
        HostTables t = build_tables_host(d->heights, d->dem_width, d->dem_height, d->exaggeration);
        t.attach(P.terrain);
        P.row_begin = 0;
        P.row_end = d->height;
        const PrimaryStart ps = primary_start(P, gx, gy);
        *t_clear = ps.t_clear;
        *level = ps.level;
        return 0;
    } catch (const Failure &) {
        return 1;
    }
}

void emul_set_use_bvh(int32_t form) { g_use_bvh = form < 0 ? 0 : (form > 2 ? 2 : form); }  // 0 sweep, 1 binary walk, 2 four wide
// FNV-1a over the node and triangle arrays of the mesh BVH built with / without worker threads
uint64_t emul_bvh_fingerprint(const float *verts, uint32_t nverts, const uint32_t *idx, uint32_t nidx, int32_t parallel,
                              uint32_t *node_count) {
    const MeshBvh bvh = build_mesh_bvh(verts, nverts, idx, nidx, parallel ? 1u : 0xFFFFFFFFu);
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    };
    mix(bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode));
    mix(bvh.tris.data(), bvh.tris.size() * sizeof(float));
    if (node_count) *node_count = (uint32_t)bvh.nodes.size();
    return h;
}
// records of the 4-wide form of the mesh BVH (0: the tree is too deep for the walk's per-level words -> binary walk)
uint32_t emul_bvh4_nodes(const float *verts, uint32_t nverts, const uint32_t *idx, uint32_t nidx, uint32_t *binary_nodes) {
    const MeshBvh bvh = build_mesh_bvh(verts, nverts, idx, nidx);
    if (binary_nodes) *binary_nodes = (uint32_t)bvh.nodes.size();
    return (uint32_t)collapse_bvh4(bvh).size();
}
// sample lanes of the frame emulation (1 = frame_pixel; 2, 4, 8 = the frame_lanes mirror)
void emul_set_frames_in_flight(uint32_t n) { g_frames_in_flight = n; }
uint64_t emul_take_retraced() {
    const uint64_t r = g_retraced;
    g_retraced = 0;
    return r;
}
void emul_set_sample_lanes(uint32_t lanes) { g_sample_lanes = (lanes == 2u || lanes == 4u || lanes == 8u) ? lanes : 1u; }
uint64_t emul_take_retraces() {
    const uint64_t r = g_retraces;
    g_retraces = 0;
    return r;
}

// The multi-bounce PBR tracer's per-pixel path code (f3d_wf_path.h) on the host: same validation and scene
// preparation as f3d_wavefront_render, the prepared arrays read in place, one OpenMP iteration per pixel.
int emul_wavefront_render(const f3d_wf_scene *scene, uint32_t width, uint32_t height, uint32_t first_frame, uint32_t frame_count,
                          float *accum, uint64_t *vertices_out, char *err, size_t errlen) {
    try {
        wf::validate_scene(*scene, width, height, frame_count);
        wf::PreparedScene prep = wf::prepare_scene(*scene, width, height);
        std::vector<wf::BlasDev> blas(prep.bvh.size());
        for (size_t m = 0; m < prep.bvh.size(); m++)
            blas[m] = wf::BlasDev{prep.bvh[m].nodes.data(), reinterpret_cast<const float4 *>(prep.bvh[m].tris.data()),
                                  (uint32_t)prep.bvh[m].nodes.size(), 0u};
        wf::SceneDev S = prep.S;
        S.spheres = prep.spheres.data();
        S.mats = prep.mats.data();
        S.blas = blas.data();
        S.inst = prep.inst.data();
        S.dir = prep.dir.data();
        S.area = prep.area.data();
        S.hair = prep.hair.data();
        HostTables terrain;  // the heightfield primitive: the terrain tracer's tables, its placement from prepare_scene
        if (scene->terrain) {
            terrain = build_tables_host(scene->terrain->heights, scene->terrain->dem_width, scene->terrain->dem_height, scene->terrain->exaggeration);
            terrain.attach(S.terrain);
        }
        const int64_t pixels = (int64_t)width * height;
        uint64_t vertices = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : vertices)
        for (int64_t p = 0; p < pixels; p++) {
            V3 acc{accum[4 * p], accum[4 * p + 1], accum[4 * p + 2]};
            ArrayPending pend;
            // (a call takes at most kMaxFramesPerCall frames -- the lane state's frame counter; frames are independent of each other)
            for (uint32_t done = 0u; done < frame_count; done += wf::kMaxFramesPerCall) {
                wf::SoloWave<ArrayPending> wave{&pend, (uint32_t)p, (uint32_t)width, {}};
                vertices += wf::trace_frames(S, first_frame + done, std::min(wf::kMaxFramesPerCall, frame_count - done), wave, [&](uint32_t, V3 total) { acc = acc + total; });
            }
            accum[4 * p] = acc.x;
            accum[4 * p + 1] = acc.y;
            accum[4 * p + 2] = acc.z;
        }
        if (vertices_out) *vertices_out = vertices;
        return 0;
    } catch (const Failure &f) {
        return report(f, err, errlen);
    }
}

// The smoke transport solver's per-voxel code (f3d_smoke_sim.h) on the host, the passes in the order f3d_smoke_sim.hip
// launches them (SmokeVolume::step, sim.rs:47-139).
int emul_smoke_step(f3d_smoke_state *st, const f3d_smoke_step_settings *settings, const f3d_smoke_emitter *emitters, uint32_t emitter_count,
                    uint32_t steps) {
    using namespace f3d::smoke;
    SimGrid G{st->dims[0], st->dims[1], st->dims[2], st->voxel_size[0], st->voxel_size[1], st->voxel_size[2], st->origin[0], st->origin[1], st->origin[2],
              st->sparse_threshold, st->time_seconds, st->frame_index};
    SimFields F{st->density, st->temperature, st->fuel, st->soot, st->humidity, st->emission_rate, st->particle_age, st->velocity, st->pressure};
    SimSettings S;
    static_assert(sizeof(SimSettings) == sizeof(f3d_smoke_step_settings) && sizeof(SimEmitter) == sizeof(f3d_smoke_emitter), "layouts differ");
    memcpy(&S, settings, sizeof(S));
    const size_t n = (size_t)G.nx * G.ny * G.nz;
    std::vector<float> tmp_a(n), tmp_b(n), vec_a(3 * n), curl(3 * n), div(n), rows(4 * (size_t)G.ny * G.nz), slabs(4 * (size_t)G.nz);
    float sums[4] = {0, 0, 0, 0};
    auto each = [&](auto &&fn) {
        for (uint32_t z = 0; z < G.nz; z++)
            for (uint32_t y = 0; y < G.ny; y++)
                for (uint32_t x = 0; x < G.nx; x++) fn(x, y, z);
    };
    auto sum = [&](uint32_t kind) {
        for (uint32_t z = 0; z < G.nz; z++) {
            for (uint32_t y = 0; y < G.ny; y++) rows[(size_t)z * G.ny + y] = sim_sum_row(G, F.density, kind, y, z);
            slabs[z] = sim_sum_seq(rows.data() + (size_t)z * G.ny, G.ny);
        }
        return sim_sum_seq(slabs.data(), G.nz);
    };
    auto project = [&](uint32_t iterations) {
        each([&](uint32_t x, uint32_t y, uint32_t z) { sim_divergence(G, F.velocity, div.data(), x, y, z); });
        std::fill(F.pressure, F.pressure + n, 0.0f);
        float *cur = F.pressure, *next = tmp_a.data();
        for (uint32_t it = 0; it < iterations; it++) {
            each([&](uint32_t x, uint32_t y, uint32_t z) { sim_jacobi(G, cur, div.data(), next, x, y, z); });
            std::swap(cur, next);
        }
        if (cur != F.pressure) memcpy(F.pressure, cur, n * sizeof(float));
        each([&](uint32_t x, uint32_t y, uint32_t z) { sim_subtract_gradient(G, F.pressure, F.velocity, x, y, z); });
    };
    auto advect = [&](float *field) {
        each([&](uint32_t x, uint32_t y, uint32_t z) { sim_advect_predict(G, field, F.velocity, tmp_a.data(), S.dt, x, y, z); });
        if (S.mac_cormack) {
            each([&](uint32_t x, uint32_t y, uint32_t z) { sim_advect_correct(G, field, F.velocity, tmp_a.data(), tmp_b.data(), S.dt, x, y, z); });
            memcpy(field, tmp_b.data(), n * sizeof(float));
        } else {
            memcpy(field, tmp_a.data(), n * sizeof(float));
        }
    };
    for (uint32_t step = 0; step < steps; step++) {
        std::fill(F.emission_rate, F.emission_rate + n, 0.0f);
        for (uint32_t e = 0; e < emitter_count; e++)
            if (G.time_seconds >= emitters[e].start_time && G.time_seconds <= emitters[e].end_time) {
                SimEmitter E;
                memcpy(&E, &emitters[e], sizeof(E));
                each([&](uint32_t x, uint32_t y, uint32_t z) { sim_emit(G, F, E, S.dt, x, y, z); });
            }
        each([&](uint32_t x, uint32_t y, uint32_t z) { sim_forces(G, F, S, x, y, z); });
        memcpy(vec_a.data(), F.velocity, 3 * n * sizeof(float));
        each([&](uint32_t x, uint32_t y, uint32_t z) { sim_advect_vector(G, vec_a.data(), F.velocity, S.dt, x, y, z); });
        if (S.diffusion > 0.0f) {
            memcpy(vec_a.data(), F.velocity, 3 * n * sizeof(float));
            for (uint32_t c = 0; c < 3u; c++)
                each([&](uint32_t x, uint32_t y, uint32_t z) { sim_diffuse(G, vec_a.data(), F.velocity, S.diffusion * S.dt, 3u, c, x, y, z); });
        }
        if (S.vorticity > 0.0f) {
            each([&](uint32_t x, uint32_t y, uint32_t z) { sim_curl(G, F.velocity, curl.data(), tmp_a.data(), x, y, z); });
            each([&](uint32_t x, uint32_t y, uint32_t z) { sim_confine(G, curl.data(), tmp_a.data(), F.velocity, S.vorticity, S.dt, x, y, z); });
        }
        project(std::max(1u, S.pressure_iterations));
        each([&](uint32_t x, uint32_t y, uint32_t z) { sim_boundary(G, F, S, x, y, z); });
        if (S.turbulence_strength > 0.0f) {
            sums[0] = sum(1u);
            sums[1] = sum(2u);
            sums[2] = sum(3u);
            each([&](uint32_t x, uint32_t y, uint32_t z) { sim_lane_shear(G, F, S, sums, x, y, z); });
        }
        if (S.mass_conservation) sums[3] = sum(0u);
        advect(F.density);
        if (S.mass_conservation) {
            sums[0] = sum(0u);
            if (sums[3] > 0.0f && sums[0] > 1.0e-12f)
                for (size_t i = 0; i < n; i++) F.density[i] *= sums[3] / sums[0];
        }
        advect(F.temperature);
        advect(F.fuel);
        advect(F.soot);
        advect(F.humidity);
        each([&](uint32_t x, uint32_t y, uint32_t z) { sim_subgrid(G, F, S, x, y, z); });
        if (S.diffusion > 0.0f) {
            float *fields[5] = {F.density, F.temperature, F.fuel, F.soot, F.humidity};
            for (float *f : fields) {
                each([&](uint32_t x, uint32_t y, uint32_t z) { sim_diffuse(G, f, tmp_a.data(), S.diffusion * S.dt, 1u, 0u, x, y, z); });
                memcpy(f, tmp_a.data(), n * sizeof(float));
            }
        }
        for (size_t i = 0; i < n; i++) sim_decay(G, F, S, i);
        project(std::max(1u, S.pressure_iterations / 2u));
        each([&](uint32_t x, uint32_t y, uint32_t z) { sim_boundary(G, F, S, x, y, z); });
        G.time_seconds += S.dt;
        G.frame_index += 1u;
    }
    st->time_seconds = G.time_seconds;
    st->frame_index = G.frame_index;
    return 0;
}

// The composite pass's per-pixel code (f3d_composite.h) on the host, parameters as f3d_smoke_composite fills them.
int emul_composite(const f3d_composite_desc *d, uint8_t *out) {
    using namespace f3d::composite;
    Params P{};
    P.mode = d->mode;
    P.width = d->width;
    P.height = d->height;
    P.has_layer = d->layer ? 1u : 0u;
    P.layer_width = d->layer ? d->layer_width : 0u;
    P.layer_height = d->layer ? d->layer_height : 0u;
    P.offset_x = d->offset_x;
    P.offset_y = d->offset_y;
    P.base_alpha = d->base_alpha;
    P.layer_alpha = d->layer_alpha;
    P.max_alpha = d->max_alpha;
    P.max_alpha_fraction = (float)((double)d->max_alpha / 255.0);
    for (uint32_t y = 0; y < P.height; y++)
        for (uint32_t x = 0; x < P.width; x++)
            reinterpret_cast<uint32_t *>(out)[(size_t)y * P.width + x] =
                pixel(P, reinterpret_cast<const uint32_t *>(d->base), reinterpret_cast<const uint32_t *>(d->layer), x, y);
    return 0;
}

// The AETHER acceptance reference's device code (f3d_aether_ref.h) on the host, in the three passes of f3d_aether_ref.hip:
// stream positions, one call per (pixel, sample, wavelength) path, ordered fold.
int emul_aether_reference(const f3d_aether_ref_desc *d, f3d_aether_ref_out *out, char *err, size_t errlen) {
    using namespace f3d::aref;
    try {
        validate_ref_desc(*d);
        check_ref_terrain(*d);
        const size_t pixels = (size_t)d->width * d->height;
        out->variance = 0.0f;
        out->terrain_primary_hits = 0;
        if (!d->enabled) {
            std::fill(out->mean_xyz, out->mean_xyz + 3 * pixels, 0.0f);
            std::fill(out->linear_rgb, out->linear_rgb + 3 * pixels, 0.0f);
            out->converged = 1;
            return 0;
        }
        HostTables t = build_tables_host(d->heights, d->dem_width, d->dem_height, d->exaggeration);
        RefScene S{};
        t.attach(S.terrain);
        fill_ref_scene(*d, S);
        const size_t samples = pixels * d->spp;
        std::vector<uint32_t> states(samples), hits(samples);
        std::vector<float> values(samples * kWavelengths), accum(4 * pixels), welford(2 * pixels);
        for (size_t p = 0; p < pixels; p++) {
            uint32_t state = pixel_seed(S, (uint32_t)(p % d->width), (uint32_t)(p / d->width));
            for (uint32_t s = 0; s < d->spp; s++) {
                states[p * d->spp + s] = state;
                rng_skip(state, kDrawsPerSample);
            }
        }
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t path = 0; path < (int64_t)(samples * kWavelengths); path++) {
            const uint32_t w = (uint32_t)(path % kWavelengths);
            const size_t ps = (size_t)path / kWavelengths;
            const uint32_t pixel = (uint32_t)(ps / d->spp);
            ArrayPending pend;
            bool primary = false;
            values[path] = sample_path(S, pixel % d->width, pixel / d->width, states[ps], w, w == 0u, primary, pend);
            if (w == 0u) hits[ps] = primary ? 1u : 0u;
        }
        for (size_t p = 0; p < pixels; p++)
            fold_pixel(values.data() + p * d->spp * kWavelengths, hits.data() + p * d->spp, d->spp, accum.data() + 4 * p, welford.data() + 2 * p);
        finalize_ref(*d, accum.data(), welford.data(), *out);
        return 0;
    } catch (const Failure &f) {
        return report(f, err, errlen);
    }
}

// div_known (csrc/f3d_div_known.h) against the division, for EVERY significand of the dividend at the given binary exponents
// (the identity is invariant under scaling by powers of two as long as nothing leaves the normal range), both signs:
// returns the number of dividends whose quotients differ in any bit.
uint64_t emul_div_known_mismatches(float d, const int32_t *exponents, uint32_t n_exponents) {
    const float r = 1.0f / d;
    uint64_t bad = 0;
    for (uint32_t e = 0; e < n_exponents; e++) {
#pragma omp parallel for reduction(+ : bad)
        for (int64_t m = 0; m < (int64_t)1 << 23; m++) {
            const uint32_t bits = ((uint32_t)(exponents[e] + 127) << 23) | (uint32_t)m;
            float a;
            memcpy(&a, &bits, sizeof(a));
            for (int sign = 0; sign < 2; sign++) {
                const float x = sign ? -a : a;
                const volatile float want = x / d;
                const float got = f3d::div_known(x, d, r);
                float w = want;
                if (memcmp(&w, &got, sizeof(float)) != 0) bad++;
            }
        }
    }
    return bad;
}
int32_t emul_div_known_divisor(float d) { return f3d::div_known_divisor(d) ? 1 : 0; }

}  // extern "C"
