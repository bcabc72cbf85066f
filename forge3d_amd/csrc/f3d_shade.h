// forge3d_amd/csrc/f3d_shade.h
// Per-pixel work of the terrain path tracer: camera rays, mesh/terrain closest and any
// hit, sun + IBL shading, the ReSTIR reservoir chain and progressive accumulation.
//
// The reference runs three dispatches per frame over 80-byte reservoirs:
//   main_terrain (hybrid_terrain_traversal.wgsl:445-610) -> pt_restir_temporal
//   (pt_restir_temporal.wgsl:54-109) -> pt_restir_spatial (pt_restir_spatial.wgsl:158-222).
// Here ONE kernel per frame does, for its pixel:  spatial reuse of the previous frame's
// temporal output (the only cross-pixel step, so it moves to the head of the next frame),
// M-clamp, the spp camera samples, candidate generation and the temporal merge -- with the
// merged history held in registers and 16-byte packed reservoirs in HBM (f3d_scene.h).
// Frame-0 AOVs come from the G-buffer pass: the reference's AOV centre ray (:583-609) and
// its G-buffer centre ray (:619-644) are the same ray.
#pragma once

#include "f3d_aether.h"
#include "f3d_march.h"
#include "f3d_trace.h"

namespace f3d {

struct SurfaceHit {
    float t;
    V3 p, n;
    uint32_t kind;  // 0 miss, 1 terrain (reference hit_type 3), 2 mesh (hit_type 0)
};

// (ray_triangle, the sweep's triangle test, lives in f3d_trace.h: the march tests a cell's triangles with it too)

// intersect_mesh, hybrid_traversal.wgsl:137-172: the reference sweeps every triangle
// (its BVH buffer is bound but never read); triangle data is wave-uniform here.
F3D_HD bool mesh_sweep(const MeshDev &M, V3 o, float tmin, V3 d, float tmax, float &t_best, V3 &n_best) {
    bool any = false;
    t_best = tmax;
    if (M.index_count < 3u) return false;
    for (uint32_t tri = 0u; tri + 2u < M.index_count; tri += 3u) {
        uint32_t i0 = M.indices[tri], i1 = M.indices[tri + 1u], i2 = M.indices[tri + 2u];
        if (i0 >= M.vertex_count || i1 >= M.vertex_count || i2 >= M.vertex_count) continue;
        float4 a = M.vertices[i0], b = M.vertices[i1], c = M.vertices[i2];
        float t;
        V3 n;
        if (ray_triangle(o, tmin, d, tmax, V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, V3{c.x, c.y, c.z}, t, n) &&
            t < t_best) {
            t_best = t;
            n_best = n;
            any = true;
        }
    }
    return any;
}

// The same answer through the threaded BVH of f3d_bvh.h: every triangle whose (padded) leaf box the
// ray touches is put through the sweep's own ray_triangle; the sweep keeps the first triangle in
// index order among those with the smallest t (`t < t_best` is strict), hence the tie rule below.
// ANY: stop at the first accepted triangle (occlusion rays only need existence).
#if defined(F3D_MESH_STATS) && defined(__HIPCC__)  // diagnostics build: what a WAVE pays for the walk
static __device__ unsigned long long g_mesh_stats[8];  // [0] wave iterations, [1] lane iterations, [2] wave leaf blocks, [3] lane leaf blocks, [4] walks (waves), [5] walks (lanes)
__device__ __forceinline__ void mesh_stat(int slot) {
    const unsigned long long m = __ballot(true);
    if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) {
        atomicAdd(&g_mesh_stats[slot], 1ull);
        atomicAdd(&g_mesh_stats[slot + 1], (unsigned long long)__popcll(m));
    }
}
#if defined(__HIP_DEVICE_COMPILE__)
#define F3D_MESH_STAT(slot) mesh_stat(slot)
#else
#define F3D_MESH_STAT(slot) (void)0
#endif
#elif defined(F3D_MESH_STATS_HOST)  // the host emulator's count of the same events (tools/experiments/bvh4_order.py)
extern unsigned long long g_host_mesh_stats[8];
#define F3D_MESH_STAT(slot) (void)__atomic_fetch_add(&g_host_mesh_stats[slot], 1ull, __ATOMIC_RELAXED)
#else
#define F3D_MESH_STAT(slot) (void)0
#endif
// Shape of the loop (round 3; gpurun logs in profiles/r03_variant_ab.log): the walk is bound by the scattered
// record loads of the 64 lanes (their latency: half of a wave's iterations hold a lane that misses the L2), not by arithmetic.
// The whole 32-byte record is requested at once (left to itself the compiler fetches the leaf word after the box test: a
// second latency on every entered node); there is ONE divergent region per node -- the triangles of an entered leaf, met in 7 % of a wave's iterations -- and the successor is
// a select (no `continue` / `return` inside the loops: each of those is a lane-mask merge per node); "found one" lives in
// best_tri, not in a boolean carried through the regions; an occlusion ray that has its answer leaves by setting node past
// the end.
template <bool ANY>
F3D_HD bool mesh_bvh(const MeshDev &M, V3 o, float tmin, V3 d, float tmax, float &t_best, V3 &n_best) {
    const float ix = (d.x < 0.0f ? -1.0f : 1.0f) / f_max(f_abs(d.x), 1e-12f);
    const float iy = (d.y < 0.0f ? -1.0f : 1.0f) / f_max(f_abs(d.y), 1e-12f);
    const float iz = (d.z < 0.0f ? -1.0f : 1.0f) / f_max(f_abs(d.z), 1e-12f);
    bool any = false;
    uint32_t best_tri = 0xFFFFFFFFu;
    t_best = tmax;
    uint32_t node = 0u;
    F3D_MESH_STAT(4);
    // plane parameters as one fma each: plane * (1/d) - o * (1/d).  In space units the rounding is |o| * 2^-24 where
    // (plane - o) * (1/d) has |plane - o| * 2^-24 -- both far inside the padding of the boxes (f3d_bvh.h), and the test only
    // has to be conservative: the triangles decide.
    const float oix = o.x * ix, oiy = o.y * iy, oiz = o.z * iz;
    const float4 *nodes = reinterpret_cast<const float4 *>(M.bvh_nodes);
    while (node < M.bvh_node_count) {
        const float4 lo = nodes[2u * node];
        float4 hi = nodes[2u * node + 1u];
        F3D_OPAQUE(hi.w);  // the whole record at once: the leaf word is not fetched after the box test
        F3D_MESH_STAT(0);
        const float ax = f_fma(lo.x, ix, -oix), bx = f_fma(hi.x, ix, -oix);
        const float ay = f_fma(lo.y, iy, -oiy), by = f_fma(hi.y, iy, -oiy);
        const float az = f_fma(lo.z, iz, -oiz), bz = f_fma(hi.z, iz, -oiz);
        const float enter = f_max(f_max(f_min(ax, bx), f_min(ay, by)), f_max(f_min(az, bz), tmin));
        const float exit = f_min(f_min(f_max(ax, bx), f_max(ay, by)), f_min(f_max(az, bz), t_best));
        const bool inside = enter <= exit * 1.00001f;  // conservative: boxes are padded, ties are kept
        const uint32_t leaf = f_bits(hi.w);
        // into the first child (the next record), or past the subtree: a missed box, or a leaf once its triangles are done
        uint32_t next = (inside & (leaf == 0u)) ? node + 1u : f_bits(lo.w);
        if (inside & (leaf != 0u)) {
            F3D_MESH_STAT(2);
            const uint32_t first = leaf >> 3, count = leaf & 7u;
            for (uint32_t k = 0u; k < count; k++) {
                const float4 a = M.bvh_tris[3u * (first + k)], b = M.bvh_tris[3u * (first + k) + 1u],
                             c = M.bvh_tris[3u * (first + k) + 2u];
                float t;
                V3 n;
                if (ray_triangle(o, tmin, d, tmax, V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, V3{c.x, c.y, c.z}, t, n)) {
                    const uint32_t tri = f_bits(a.w);
                    // (best_tri doubles as the "found one" flag: a boolean carried through the divergent regions costs the
                    // wave a lane-mask merge per region and node)
                    if (ANY ? best_tri == 0xFFFFFFFFu : (t < t_best || (t == t_best && tri < best_tri))) {
                        t_best = t;
                        n_best = n;
                        best_tri = tri;
                    }
                }
            }
            if (ANY && best_tri != 0xFFFFFFFFu) next = M.bvh_node_count;  // existence is all an occlusion ray asks for
        }
        node = next;
    }
    any = best_tri != 0xFFFFFFFFu;
    return any;
}

// The same answer through the 4-wide form of the tree (f3d_bvh.h collapse_bvh4, f3d_scene.h Bvh4Node): a visited node
// tests its four CHILDREN's boxes from one 128-byte record, so the chain of dependent loads is as long as the number of
// nodes a ray enters, not the number of boxes it tests (round 4; the binary walk above measured ~45 dependent loads a ray
// on the 600 000-triangle stand-in, each ~1 300 cycles of a wave's time).  Children wait their turn in one word per LEVEL
// -- (first_child << 8) | (how many - 1) << 6 | their slot numbers, nearest first -- in the lane's column (Stack::stack_put / stack_get:
// LDS rows on the device), and `open` says which levels hold one; no other stack.  The order of visits is immaterial to
// the answer -- the sweep's "smallest t, lowest index among equal t", or existence (ANY) -- and is NEAREST FIRST: the
// entered inner children are sorted by their entry parameter (five min / max pairs on integer keys), so that a near
// child's triangles shorten t_best before the far children's boxes are tested (-4.3 % node visits and +2.7 % on the
// configs[3] stand-in, 4 001 / 4 010 -> 4 116 / 4 114 Msamples/s; -DF3D_BVH4_SLOT_ORDER is the slot-order walk it replaced,
// whose level word was (first_child << 4) | mask of the inner slots still to visit).
template <bool ANY, class Stack>
F3D_HD bool mesh_bvh4(const MeshDev &M, V3 o, float tmin, V3 d, float tmax, float &t_best, V3 &n_best, Stack &stk) {
    const float ix = (d.x < 0.0f ? -1.0f : 1.0f) / f_max(f_abs(d.x), 1e-12f);
    const float iy = (d.y < 0.0f ? -1.0f : 1.0f) / f_max(f_abs(d.y), 1e-12f);
    const float iz = (d.z < 0.0f ? -1.0f : 1.0f) / f_max(f_abs(d.z), 1e-12f);
    const float oix = o.x * ix, oiy = o.y * iy, oiz = o.z * iz;  // plane * (1/d) - o * (1/d): see mesh_bvh
    uint32_t best_tri = 0xFFFFFFFFu;
    t_best = tmax;
    const float4 *nodes = reinterpret_cast<const float4 *>(M.bvh4_nodes);
    constexpr uint32_t kDone = 0xFFFFFFFFu;
    // nearest first is for the rays whose t_best shrinks; an occlusion ray only asks whether there is a triangle at all and mostly
    // finds none -- -DF3D_BVH4_ANY_SLOT_ORDER (A/B, round 6) walks its children in slot order, without the sort
#if defined(F3D_BVH4_SLOT_ORDER)
    constexpr bool kNearestFirst = false;
#elif defined(F3D_BVH4_ANY_SLOT_ORDER)
    constexpr bool kNearestFirst = !ANY;
#else
    constexpr bool kNearestFirst = true;
#endif
    uint32_t node = 0u, level = 0u, open = 0u;
    while (node != kDone) {
        const float4 *rec = nodes + 8u * node;
        const float4 lox = rec[0], hix = rec[1], loy = rec[2], hiy = rec[3], loz = rec[4], hiz = rec[5];
        const float4 leaf = rec[6], meta = rec[7];
        F3D_MESH_STAT(0);
        uint32_t hit = 0u;
        float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f, e3 = 0.0f;
#define F3D_BVH4_KEEP(S, v) if (kNearestFirst) e##S = v;
#define F3D_BVH4_SLOT(S, C)                                                                                       \
        {                                                                                                         \
            const float ax = f_fma(lox.C, ix, -oix), bx = f_fma(hix.C, ix, -oix);                                 \
            const float ay = f_fma(loy.C, iy, -oiy), by = f_fma(hiy.C, iy, -oiy);                                 \
            const float az = f_fma(loz.C, iz, -oiz), bz = f_fma(hiz.C, iz, -oiz);                                 \
            const float enter = f_max(f_max(f_min(ax, bx), f_min(ay, by)), f_max(f_min(az, bz), tmin));           \
            const float exit = f_min(f_min(f_max(ax, bx), f_max(ay, by)), f_min(f_max(az, bz), t_best));          \
            hit |= (enter <= exit * 1.00001f) ? (1u << S) : 0u; /* conservative: boxes are padded, ties are kept; an empty slot (both planes at +inf, f3d_bvh.h) never passes: enter = +inf or exit = -inf */ \
            F3D_BVH4_KEEP(S, enter)                                                                               \
        }
        F3D_BVH4_SLOT(0, x)
        F3D_BVH4_SLOT(1, y)
        F3D_BVH4_SLOT(2, z)
        F3D_BVH4_SLOT(3, w)
#undef F3D_BVH4_SLOT
#undef F3D_BVH4_KEEP
        const uint32_t first_child = f_bits(meta.x), inner_slots = (1u << f_bits(meta.y)) - 1u;
        uint32_t inner = hit & inner_slots, leaves = hit & ~inner_slots;
        while (leaves != 0u) {  // the divergent region of the walk: the triangles of the leaf children the ray enters
            const uint32_t s = (uint32_t)__builtin_ctz(leaves);
            leaves &= leaves - 1u;
            const uint32_t word = f_bits(s == 0u ? leaf.x : (s == 1u ? leaf.y : (s == 2u ? leaf.z : leaf.w)));
            const uint32_t first = word >> 3, count = word & 7u;
            F3D_MESH_STAT(2);
            for (uint32_t k = 0u; k < count; k++) {
                const float4 a = M.bvh_tris[3u * (first + k)], b = M.bvh_tris[3u * (first + k) + 1u], c = M.bvh_tris[3u * (first + k) + 2u];
                float t;
                V3 n;
                if (ray_triangle(o, tmin, d, tmax, V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, V3{c.x, c.y, c.z}, t, n)) {
                    const uint32_t tri = f_bits(a.w);
                    if (ANY ? best_tri == 0xFFFFFFFFu : (t < t_best || (t == t_best && tri < best_tri))) {
                        t_best = t;
                        n_best = n;
                        best_tri = tri;
                    }
                }
            }
        }
        if (ANY && best_tri != 0xFFFFFFFFu) {  // existence is all an occlusion ray asks for
            inner = 0u;
            open = 0u;
        }
        uint32_t next = kDone;
        if constexpr (kNearestFirst) {
        // the next node: the NEAREST inner child entered (the others wait, nearest first, in this level's word:
        // (first_child << 8) | (how many - 1) << 6 | three 2-bit slot numbers), or the next waiting child of the deepest level
        // that holds one.  A near child's triangles shorten t_best before the far children's boxes are tested.
        if (inner != 0u) {
            // sort keys of the entered inner children: the entry parameter (>= tmin >= 0: its bits order as integers; only an order is at stake) with
            // the slot number in the two lowest bits; a child not entered sorts last.  (Testing the entries again against a
            // t_best the leaf children's triangles have just shortened saves 0.02 % of the visits: not done.)
            uint32_t k0 = (inner & 1u) ? ((f_bits(e0) & ~3u) | 0u) : 0xFFFFFFFFu;
            uint32_t k1 = (inner & 2u) ? ((f_bits(e1) & ~3u) | 1u) : 0xFFFFFFFFu;
            uint32_t k2 = (inner & 4u) ? ((f_bits(e2) & ~3u) | 2u) : 0xFFFFFFFFu;
            uint32_t k3 = (inner & 8u) ? ((f_bits(e3) & ~3u) | 3u) : 0xFFFFFFFFu;
            {   // five compare-exchanges: min / max on the integer keys
                uint32_t a = k0 < k1 ? k0 : k1, b = k0 < k1 ? k1 : k0, c = k2 < k3 ? k2 : k3, e = k2 < k3 ? k3 : k2;
                k0 = a < c ? a : c;
                const uint32_t m1 = a < c ? c : a, m2 = b < e ? b : e;
                k3 = b < e ? e : b;
                k1 = m1 < m2 ? m1 : m2;
                k2 = m1 < m2 ? m2 : m1;
            }
            const uint32_t more = (uint32_t)__builtin_popcount(inner) - 1u;
            if (more != 0u) {
                stk.stack_put(level, (first_child << 8) | ((more - 1u) << 6) | (k1 & 3u) | ((k2 & 3u) << 2) | ((k3 & 3u) << 4));
                open |= 1u << level;
            }
            next = first_child + (k0 & 3u);
            level = level + 1u;
        } else if (open != 0u) {
            const uint32_t at = 31u - (uint32_t)__builtin_clz(open);
            const uint32_t word = stk.stack_get(at);
            const uint32_t left = (word >> 6) & 3u;
            if (left != 0u) stk.stack_put(at, (word & ~0xFFu) | ((left - 1u) << 6) | ((word & 63u) >> 2));
            else open &= ~(1u << at);
            next = (word >> 8) + (word & 3u);
            level = at + 1u;
        }
        } else {
        // the next node: the first inner child entered (its siblings wait in this level's word), or the next waiting sibling
        // of the deepest level that holds one, or nothing
        if (inner != 0u) {
            const uint32_t rest = inner & (inner - 1u);
            if (rest != 0u) {
                stk.stack_put(level, (first_child << 4) | rest);
                open |= 1u << level;
            }
            next = first_child + (uint32_t)__builtin_ctz(inner);
            level = level + 1u;
        } else if (open != 0u) {
            const uint32_t at = 31u - (uint32_t)__builtin_clz(open);
            const uint32_t word = stk.stack_get(at);
            const uint32_t rest = (word & 15u) & ((word & 15u) - 1u);
            if (rest != 0u) stk.stack_put(at, (word & ~15u) | rest);
            else open &= ~(1u << at);
            next = (word >> 4) + (uint32_t)__builtin_ctz(word & 15u);
            level = at + 1u;
        }
        }
        node = next;
    }
    return best_tri != 0xFFFFFFFFu;
}

template <class Stack>
F3D_HD bool mesh_closest(const MeshDev &M, V3 o, float tmin, V3 d, float tmax, float &t_best, V3 &n_best, Stack &stk) {
    if (M.bvh4_nodes) return mesh_bvh4<false>(M, o, tmin, d, tmax, t_best, n_best, stk);
    if (M.bvh_nodes) return mesh_bvh<false>(M, o, tmin, d, tmax, t_best, n_best);
    return mesh_sweep(M, o, tmin, d, tmax, t_best, n_best);
}
// Is any triangle of the mesh hit (occlusion rays)?
template <class Stack>
F3D_HD bool mesh_any(const MeshDev &M, V3 o, float tmin, V3 d, float tmax, float &t, V3 &n, Stack &stk) {
    if (M.bvh4_nodes) return mesh_bvh4<true>(M, o, tmin, d, tmax, t, n, stk);
    if (M.bvh_nodes) return mesh_bvh<true>(M, o, tmin, d, tmax, t, n);
    return mesh_sweep(M, o, tmin, d, tmax, t, n);
}

// intersect_hybrid, hybrid_traversal.wgsl:175-201 (closest hit, curvature off)
// t_clear / level: a certificate for camera rays (f3d_cone.h); t_clear = 0 starts the march at the root.
template <class Pending>
F3D_HD SurfaceHit closest_hit(const FrameParams &P, V3 o, float tmin, V3 d, float tmax, Pending &pend, float t_clear = 0.0f,
                              uint32_t start_level = 0u) {
    SurfaceHit best;
    best.kind = 0u;
    best.t = tmax;
    best.p = V3{0, 0, 0};
    best.n = V3{0, 0, 0};
#if defined(F3D_CLOSEST_TERRAIN_FIRST) && !defined(F3D_MESH_FIRST) && !defined(F3D_TRAVERSAL_DESCENT)
    // A/B (round 6, MEASURED SLOWER: configs[3] stand-in 33.06 ms a frame against 32.51, 88 B of scratch against 48; bit-identical).
    // The reference asks the mesh first and then the terrain with tmax = the mesh's hit (below).  A ray that meets no triangle walks
    // the tree along its WHOLE length to prove it -- on, underground, to the far side of the mesh's box -- and almost every camera
    // ray of a city on a DEM is such a ray.  Here the terrain goes first too (the occlusion rays' order since round 3),
    // with the ray's own tmax, and the mesh is asked for hits up to the END of the terrain's hit leaf only:
    //   * no triangle before that: the reference's terrain march (tmax = a mesh hit beyond the leaf, or the ray's own) visits the
    //     leaves up to the hit leaf with the same intervals and returns this very hit, which is nearer than any triangle;
    //   * a triangle before that: the reference's order from there on -- the terrain again with tmax = that hit (a leaf cut by
    //     tmax is solved over the cut interval: only marching it again gives those bits), the nearer of the two.
    // One copy of the march: a loop of at most two passes, left after the first by the lanes without such a triangle.
    if (Pending::kMesh && P.mesh.traversal_mode == 0u) {
        float tm = tmax;
#pragma unroll 1
        for (uint32_t pass = 0u;; pass++) {
            const RayCtx r = make_ray(P.terrain, o, tmin, d, tm, false);
            uint32_t cell = 0u;
            const TraceHit th = march_terrain_from<false, false>(P.terrain, r, false, march_begin_at(P.terrain, r, t_clear, start_level), pend, 3.0e38f, &cell);
            if (th.hit && th.t < best.t) {
                best.kind = 1u;
                best.t = th.t;
                best.n = th.n;
                best.p = along(o, th.t, d);
            }
            if (pass != 0u) break;
            float bound = tmax;
            if (th.hit) {
                float lo, hi;
                march_leaf_interval(P.terrain, r, cell & 0xFFFFu, cell >> 16, lo, hi);
                bound = f_min(f_from_bits(f_bits(hi) + 1u), tmax);  // (hi > tmin > 0: the next float up -- a triangle AT the leaf's end counts as before it)
            }
            float t;
            V3 n;
            if (!mesh_closest(P.mesh, o, tmin, d, bound, t, n, pend)) break;
            best.kind = 2u;
            best.t = t;
            best.n = n;
            best.p = along(o, t, d);
            tm = t;
        }
        return best;
    }
#endif
#if !defined(F3D_TIMING_NO_MESH_CLOSEST)  // timing experiment only (wrong image)
    if (Pending::kMesh && P.mesh.traversal_mode == 0u) {
#else
    if (false) {
#endif
        float t;
        V3 n;
        if (mesh_closest(P.mesh, o, tmin, d, tmax, t, n, pend) && t < best.t) {
            best.kind = 2u;
            best.t = t;
            best.n = n;
            best.p = along(o, t, d);
        }
    }
    RayCtx r = make_ray(P.terrain, o, tmin, d, best.t, false);
#if defined(F3D_TRAVERSAL_DESCENT)
    TraceHit th = trace_terrain(P.terrain, r, false, pend);  // the reference-shaped sorted descent
#else
    // camera rays enter the footprint from outside: the march starts at the root, or where the pixel's certificate ends
    TraceHit th = march_terrain_from<false, false>(P.terrain, r, false, march_begin_at(P.terrain, r, t_clear, start_level), pend);
#endif
    if (th.hit && th.t < best.t) {
        best.kind = 1u;
        best.t = th.t;
        best.n = th.n;
        best.p = along(o, th.t, d);
    }
    return best;
}

// intersect_hybrid_optimized + intersect_shadow_ray / intersect_ibl_occlusion_ray,
// hybrid_traversal.wgsl:204-259 (any hit; early_exit 0.01; max_distance 1e30)
// The answer is an OR: any accepted triangle makes the reference return true whatever the terrain does (:204-259 -- it takes
// the closest mesh hit, returns at once when that is nearer than its early-exit distance and otherwise keeps hit = true),
// and without one the terrain is asked with the ray's own tmax.  So the order is free, and the terrain goes FIRST: the
// rays with the longest mesh walks are the flat ones that run through the building layer for kilometres -- which are the
// ones the terrain stops.  On the S4 stand-in (600 000 triangles) the longest walk of a wave of IBL rays halves
// (DESIGN.md 9.1).  -DF3D_MESH_FIRST is the A/B switch for the reference's order.
template <class Pending>
F3D_HD bool occluded(const FrameParams &P, V3 o, float tmin, V3 d, float tmax, bool apply_curvature,
                     Pending &pend, float terrain_tmax = 1e30f) {
#if defined(F3D_MESH_FIRST)
    if (Pending::kMesh && P.mesh.traversal_mode == 0u) {
        float t;
        V3 n;
        if (mesh_any(P.mesh, o, tmin, d, tmax, t, n, pend)) return t < 1e30f;
    }
#endif
    // terrain_tmax: a certificate that no terrain lies beyond it on this ray (f3d_cone.h sun_clear_from): the march stops
    // after the node that contains it (sun rays only: the curved instantiation carries the stop rule)
    RayCtx r = make_ray(P.terrain, o, tmin, d, tmax, apply_curvature);
    // A/B (round 6, -DF3D_MESH_FUSED; MEASURED SLOWER, not in the shipped library): the kernels compiled for scenes with a mesh
    // march BOTH at once (f3d_march.h FUSE, f3d_meshgrid.h) -- the mesh is a second band of every node of the terrain's pyramid,
    // and a cell whose mesh band the ray meets has its triangles put through the sweep's own test by the drain: no second
    // traversal.  The certificates that promise "no TERRAIN beyond" do not hold for the mesh: no stop parameter, and the climbing
    // ray leaves only above both the terrain's and the mesh's highest point.  Without a grid (a mesh that reaches beyond the DEM,
    // lists that explode) the second band is the terrain's own and the tree is walked as before.  Bit-identical (1 000 scenes on
    // the emulator against the oracle's sweep, the device's mesh tests); on the configs[3] stand-in 36.4 ms a frame against 32.4:
    // a second band per step would cost 0.4 ms, but rays inside the 60 m building layer descend to the cells wherever a node holds
    // a building at their height (+7.7 ms of steps, as much as the tree walks they replace) and a cell's triangles have no box
    // of their own in front of them (+5 ms).
#if defined(F3D_MESH_FUSED) && !defined(F3D_MESH_FIRST) && !defined(F3D_TRAVERSAL_DESCENT)
    constexpr bool kFuse = Pending::kMesh;
#else
    constexpr bool kFuse = false;
#endif
    const bool grid = kFuse && P.terrain.mesh_cell_start != nullptr;
    if (grid) {
        if (r.y_exit < 3.0e38f) r.y_exit = f_max(r.y_exit, P.terrain.mesh_top);
        terrain_tmax = 3.0e38f;
    }
#if defined(F3D_TRAVERSAL_DESCENT)
    TraceHit th = trace_terrain(P.terrain, r, true, pend);  // the reference-shaped sorted descent
#else
    // occlusion rays start on the surface: the march starts in the origin's cell.  Only sun
    // rays carry the curvature policy (apply_curvature is a compile-time constant per call site).
    // (with the curvature policy switched off for the whole render, c2 = 0 and fma(t*t, 0, y) == y: the
    // curved instantiation then computes the flat answers exactly, so there is no third copy of the march)
    TraceHit th = apply_curvature ? march_terrain<true, kFuse>(P.terrain, r, true, true, pend, terrain_tmax)
                                  : march_terrain<false, kFuse>(P.terrain, r, true, true, pend, terrain_tmax);
#endif
    bool hit = th.hit && th.t < tmax && th.t < 1e30f;
#if !defined(F3D_MESH_FIRST) && !defined(F3D_TIMING_NO_MESH_ANY)
    if (Pending::kMesh && P.mesh.traversal_mode == 0u) {
        float t;
        V3 n;
        // (accepted triangles have t < tmax; `t < 1e30` is the reference's own last word on the mesh hit)
        if (!hit && !grid && mesh_any(P.mesh, o, tmin, d, tmax, t, n, pend)) hit = t < 1e30f;
    }
#endif
    return hit;
}

// terrain_env_radiance, hybrid_terrain_traversal.wgsl:392-405
F3D_HD V3 env_radiance(const EnvDev &E, V3 dir) {
    if (E.width == 0u || E.height == 0u) return V3{E.intensity, E.intensity, E.intensity};
    uint32_t width = E.width, height = E.height;
    F3D_OPAQUE_UNIFORM(width);  // (the map's constants are formed here, not in front of the sample loop: f3d_math.h)
    F3D_OPAQUE_UNIFORM(height);
    V3 d = normalize(dir);
    float uu = atan2_det(d.z, d.x) / (2.0f * kPi) + 0.5f;
    float vv = acos_det(f_clamp(d.y, -1.0f, 1.0f)) / kPi;
    uint32_t px = sat_u32(uu * (float)width), py = sat_u32(vv * (float)height);
    px = px < width - 1u ? px : width - 1u;
    py = py < height - 1u ? py : height - 1u;
    float4 t = E.texels[(size_t)py * width + px];
    return V3{t.x * E.intensity, t.y * E.intensity, t.z * E.intensity};
}

// terrain_tent_offset, :409-414
F3D_HD float tent_offset(float u) {
    if (u < 0.5f) return f_sqrt(2.0f * u) - 1.0f;
    return 1.0f - f_sqrt(2.0f * (1.0f - u));
}

// terrain_cosine_dir, :421-431
F3D_HD V3 cosine_dir(V3 n, float u1, float u2) {
    float sign = n.z < 0.0f ? -1.0f : 1.0f;
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    V3 t = V3{1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x};
    V3 bt = V3{b, sign + n.y * n.y * a, -n.y};
    float rr = f_sqrt(u1);
    float sn, cs;
    sincos_turn(u2, sn, cs);
    return normalize(combine(rr * cs, t, rr * sn, bt, f_sqrt(f_max(0.0f, 1.0f - u1)), n));
}

// Camera ray through pixel (gx, gy) + sub-pixel offset, :481-485
F3D_HD V3 camera_dir(const CameraDev &C, uint32_t gx, uint32_t gy, float jx, float jy) {
    float ndc_x = (((float)gx + 0.5f + jx) / (float)C.width) * 2.0f - 1.0f;
    float ndc_y = (1.0f - ((float)gy + 0.5f + jy) / (float)C.height) * 2.0f - 1.0f;
    V3 rd = normalize(V3{ndc_x * C.half_w, ndc_y * C.half_h, -1.0f});
    return normalize(combine(rd.x, C.right, rd.y, C.up, rd.z, neg(C.forward)));
}

}  // namespace f3d
#include "f3d_cone.h"  // primary_start: where the camera rays of a pixel may start
namespace f3d {

// ---- reservoirs -------------------------------------------------------------------
struct Reservoir {  // register form of PackedReservoir
    float w_sum;
    uint32_t m;
    float weight;
    float target_pdf;
    bool directional;  // sample.light_type == 1
};
F3D_HD Reservoir unpack(const PackedReservoir &p) {
    return Reservoir{p.w_sum, p.m_lt & ~kLightTypeBit, p.weight, p.target_pdf, (p.m_lt & kLightTypeBit) != 0u};
}
F3D_HD PackedReservoir pack(const Reservoir &r) {
    return PackedReservoir{r.w_sum, r.m | (r.directional ? kLightTypeBit : 0u), r.weight, r.target_pdf};
}
F3D_HD Reservoir empty_reservoir() { return Reservoir{0.0f, 0u, 0.0f, 0.0f, false}; }

// Every division of the reservoir arithmetic is a * (1 / b), the reciprocal rounded once.  Which of two
// weights that are both exactly 1 in real arithmetic wins the temporal pass of frame 1 depends on it, and with it the start of a
// 513-frame relaxation of the reuse weight: this form reproduces the reference's golden (12.6 % of those ties go to `prev`, the
// golden implies 13 %), IEEE division (0 %) is 4 % too bright on every sun-lit pixel.  The oracle spells the same (restir_div).
F3D_HD float restir_div(float a, float b) { return a * (1.0f / b); }

// Address of image pixel (x, y) in a strip-local reservoir buffer (halo rows included).
F3D_HD size_t reservoir_index(const FrameParams &P, uint32_t x, uint32_t y) {
    return (size_t)(y + kHaloRows - P.row_begin) * P.cam.width + x;
}

// pt_restir_spatial (pt_restir_spatial.wgsl:158-222 with consider_candidate :45-111)
// specialised to the light table the driver binds (render_terrain.rs:756-781): one
// directional light of importance 1 => p_sel = 1; one zeroed area light, never selected.
// `frame` is the frame whose spatial pass this is.  n_raw = G-buffer normal record.
// PREFETCH (the pixel-parallel passes: k_head, k_merge, k_resolve): which neighbour comes next depends on the ones
// before it only through ONE bit each -- did the candidate draw its selection number -- so a lane would wait for
// nine dependent loads in a row.  Validity is spatially coherent, so the walk is predicted with every neighbour
// behaving as the pixel itself did, the eight predicted records are fetched at once, and the real walk takes a
// record from that set whenever it is the one it wants (else it loads it: the result is the same either way).
template <bool PREFETCH = false>
F3D_HD Reservoir spatial_reuse(const FrameParams &P, const PackedReservoir *res, uint32_t gx, uint32_t gy,
                               uint32_t frame, V3 n_raw) {
    const uint32_t W = P.cam.width, H = P.cam.height;
    const uint32_t idx = gy * W + gx;
    uint32_t seed = (P.cam.seed_hi ^ frame) + idx * 1664525u + 1013904223u;
    const Reservoir self = unpack(res[reservoir_index(P, gx, gy)]);
    const bool facing = f_max(dot(normalize(n_raw), P.light.wi_reuse), 0.0f) > 0.0f;

    bool chosen_directional = self.directional;
    float chosen_pdf = self.target_pdf;
    float wsum = 0.0f;
    uint32_t m_total = 0u;
    auto consider = [&](const Reservoir &r) F3D_LAMBDA {  // returns whether the candidate drew a number
        if (r.m == 0u) return false;
        if (!r.directional) return false;  // light_type 0: no sample ever stored
        if (!facing) return false;
        const float p_curr = 1.0f;
        if (r.target_pdf <= 0.0f) return false;
        const float w = r.w_sum * restir_div(p_curr, f_max(r.target_pdf, 1e-6f));
        if (w <= 0.0f) return false;
        wsum = wsum + w;
        const float u = rng_next(seed);
        if (u < restir_div(w, wsum)) {
            chosen_directional = true;
            chosen_pdf = p_curr;
        }
        return true;
    };
    auto neighbour = [&](uint32_t &stream, size_t &index) F3D_LAMBDA {  // false: the offset (0, 0) is skipped
        const int rx = (int)f_floor(rng_next(stream) * 7.0f) - 3;
        const int ry = (int)f_floor(rng_next(stream) * 7.0f) - 3;
        if (rx == 0 && ry == 0) return false;
        int qx = (int)gx + rx, qy = (int)gy + ry;
        qx = qx < 0 ? 0 : (qx > (int)W - 1 ? (int)W - 1 : qx);
        qy = qy < 0 ? 0 : (qy > (int)H - 1 ? (int)H - 1 : qy);
        index = reservoir_index(P, (uint32_t)qx, (uint32_t)qy);
        return true;
    };
    const bool self_drew = consider(self);
    m_total += self.m;
    constexpr size_t kNone = ~(size_t)0;
    PackedReservoir ahead[PREFETCH ? 8 : 1];
    size_t ahead_index[PREFETCH ? 8 : 1];
    if (PREFETCH) {
        uint32_t stream = seed;
#pragma unroll
        for (uint32_t i = 0u; i < 8u; i++) {
            ahead_index[i] = kNone;
            size_t index;
            if (!neighbour(stream, index)) continue;
            ahead_index[i] = index;
            ahead[i] = res[index];
            if (self_drew) (void)rng_next(stream);
        }
    }
#pragma unroll
    for (uint32_t i = 0u; i < 8u; i++) {
        size_t index;
        if (!neighbour(seed, index)) continue;
        const Reservoir rn = unpack((PREFETCH && ahead_index[PREFETCH ? i : 0u] == index) ? ahead[PREFETCH ? i : 0u] : res[index]);
        (void)consider(rn);
        m_total += rn.m;
    }
    Reservoir out;
    out.directional = chosen_directional;
    out.target_pdf = chosen_pdf;
    out.w_sum = wsum;
    out.m = m_total;
    out.weight = (out.w_sum > 0.0f && out.target_pdf > 0.0f) ? restir_div(out.w_sum, (float)out.m * out.target_pdf) : 0.0f;
    return out;
}

// pt_restir_temporal.wgsl:54-109
F3D_HD Reservoir temporal_merge(const Reservoir &rp, const Reservoir &rc) {
    const bool pv = rp.m > 0u && rp.weight > 0.0f && rp.target_pdf > 0.0f;
    const bool cv = rc.m > 0u && rc.weight > 0.0f && rc.target_pdf > 0.0f;
    if (!pv) return rc;
    if (!cv) return rp;
    Reservoir ro = (rp.weight > rc.weight) ? rp : rc;
    ro.m = rp.m + rc.m;
    ro.w_sum = rp.w_sum + rc.w_sum;
    ro.weight = (ro.w_sum > 0.0f && ro.target_pdf > 0.0f) ? restir_div(ro.w_sum, (float)ro.m * ro.target_pdf) : 0.0f;
    return ro;
}

// ---- one accumulation frame of one pixel ---------------------------------------------
struct FrameHead {
    bool centre_hit;  // the G-buffer's centre ray hit a surface (hit-flag prediction of frame_lanes)
    bool prev_valid;  // the merged history carries a usable sun sample (:463-465)
    float reuse_w;    // clamp(prev.weight, 0, 4) or 1
    uint32_t rng;
};

// Everything main_terrain does before its sample loop (:452-471), with the previous frame's
// spatial reuse pass evaluated lazily for this pixel.  The merged, M-clamped history is needed
// again only by the temporal merge at the very end of the frame, so it is parked in this pixel's
// slot of the OUTPUT reservoir buffer (which the same lane overwrites in frame_tail) instead of
// occupying five registers across the whole sample loop.
template <bool PREFETCH = false>
F3D_HD FrameHead frame_head(const FrameParams &P, uint32_t gx, uint32_t gy) {
    const size_t lp = (size_t)(gy - P.row_begin) * P.cam.width + gx;  // strip-local pixel
    const float4 g = P.gbuffer_n[lp];
    Reservoir prev = empty_reservoir();
    if (P.frame_index > 0u) prev = spatial_reuse<PREFETCH>(P, P.res_in, gx, gy, P.frame_index - 1u, V3{g.x, g.y, g.z});
    // M-clamp, hybrid_terrain_traversal.wgsl:452-462
    if (prev.m > kRestirMCap) {
        const float scale = restir_div((float)kRestirMCap, (float)prev.m);
        prev.w_sum = prev.w_sum * scale;
        prev.m = kRestirMCap;
        if (prev.target_pdf > 0.0f) prev.weight = restir_div(prev.w_sum, (float)prev.m * prev.target_pdf);
    }
    P.res_out[reservoir_index(P, gx, gy)] = pack(prev);
    FrameHead h;
    h.centre_hit = g.w != 0.0f;
    h.prev_valid = P.frame_index > 0u && prev.m > 0u && prev.weight > 0.0f && prev.target_pdf > 0.0f &&
                   prev.directional;
    h.reuse_w = h.prev_valid ? f_clamp(prev.weight, 0.0f, 4.0f) : 1.0f;
    h.rng = P.cam.seed_hi ^ (gx * 1664525u) ^ (gy * 1013904223u) ^ (P.frame_index * 92837111u) ^ P.cam.seed_lo;
    return h;
}

// FrameHead as an 8-byte record (the sample-lane frame runs frame_head in its own pixel-parallel
// kernel, f3d_kernels.hip k_head, instead of once per sample lane); the stream seed is recomputed.
constexpr uint32_t kHeadPrevValid = 1u, kHeadCentreHit = 2u;
F3D_HD uint2 pack_head(const FrameHead &h) {
    return uint2{f_bits(h.reuse_w), (h.prev_valid ? kHeadPrevValid : 0u) | (h.centre_hit ? kHeadCentreHit : 0u)};
}
F3D_HD FrameHead unpack_head(const FrameParams &P, uint32_t gx, uint32_t gy, uint2 rec) {
    FrameHead h;
    h.centre_hit = (rec.y & kHeadCentreHit) != 0u;
    h.prev_valid = (rec.y & kHeadPrevValid) != 0u;
    h.reuse_w = f_from_bits(rec.x);
    h.rng = P.cam.seed_hi ^ (gx * 1664525u) ^ (gy * 1013904223u) ^ (P.frame_index * 92837111u) ^ P.cam.seed_lo;
    return h;
}

// ---- one camera sample, cut where the samples of a pixel depend on one another -----------
// Samples of a pixel-frame are chained only through (a) the RNG stream -- a sample draws 2 numbers
// for its jitter and 2 more for the IBL direction IF its primary ray hit (:477-478, :537-538), so
// the state at the start of sample s depends on the hit flags of samples < s -- and (b) two
// order-sensitive accumulations: the radiance sum and the candidate reservoir (:500-512).  The
// tracing in between is independent.  frame_pixel below walks the samples one after the other;
// the sample-lane form of the frame kernel (f3d_kernels.hip) traces several samples of a pixel on
// neighbouring lanes and replays (b) in sample order.
struct PrimaryHit {
    SurfaceHit hit;
    V3 rd;
    uint32_t rng;    // stream state after the two jitter draws
    float sun_tmax;  // the sample's sun ray meets no terrain beyond this parameter (1e30: no certificate)
    uint32_t cert;   // strip-local pixel whose certificates cover this sample's hit point; 0xFFFFFFFF: none
};

template <class Pending>
F3D_HD PrimaryHit sample_primary(const FrameParams &P, uint32_t gx, uint32_t gy, uint32_t rng, Pending &pend) {
    PrimaryHit ph;
    const float jx = tent_offset(rng_next(rng)) * 0.5f;
    const float jy = tent_offset(rng_next(rng)) * 0.5f;
    ph.rd = camera_dir(P.cam, gx, gy, jx, jy);
    uint2 start = uint2{0u, 0u};
#if !defined(F3D_NO_PRIMARY_START)  // A/B builds (tools/build_variant.sh)
    if (P.primary_start) start = P.primary_start[(size_t)(gy - P.row_begin) * P.cam.width + gx];
#endif
    ph.hit = closest_hit(P, P.cam.origin, 1e-3f, ph.rd, 1e30f, pend, f_from_bits(start.x), start.y);
    ph.sun_tmax = 1e30f;
    ph.cert = 0xFFFFFFFFu;
#if !defined(F3D_NO_SUN_CLEAR)  // A/B builds
    if (P.sun_clear && ph.hit.kind != 0u) {
        const uint32_t lp = (gy - P.row_begin) * P.cam.width + gx;
        const float2 c = P.sun_clear[lp];
        float sx = P.terrain.spacing_x;
        F3D_OPAQUE_UNIFORM(sx);
        const float cell = f_min(sx, P.terrain.spacing_z);
        if (c.y > 0.0f && f_abs(ph.hit.t - c.y) <= sun_depth_slack(c.y, pixel_cone_delta(P.cam), cell)) {
            ph.cert = lp;  // the sample's hit point is within the radius the pixel's certificates allow for
            if (c.x < 1e30f) ph.sun_tmax = c.x;
        }
    }
#endif
    ph.rng = rng;
    return ph;
}

struct SampleOut {
    V3 a, b;           // radiance = (radiance + a) + b: hit -> (sun, ibl); miss -> (env, 0)
    float target_pdf;  // candidate weight of the hit (:500-512); 0 = no candidate
};

// The IBL-occlusion ray of a sample and what its verdict scales: ibl = b0 * (occluded ? 0 : 1).
struct IblRay {
    bool valid;  // false: the primary ray missed, there is no IBL ray
    V3 o, d;     // origin (hit point lifted off the surface), cosine-weighted direction
    V3 b0;       // albedo * env(d)
    float key;   // cos(normal, d): small = grazing = a long march (scheduling hint only)
    float t_stop;  // no terrain beyond this parameter (f3d_cone.h ibl_stop); 3e38: no certificate
};

// What the shading of a sample needs from its two occlusion rays, with everything else already evaluated (:486-545):
// radiance = (Y * vis_sun) * reuse_w + b0 * vis_ibl.  The fused kernels trace the two rays on the spot; the wavefront
// kernels (f3d_kernels.hip k_wf_primary / k_wf_occl) put them into queues.
struct ShadeSetup {
    IblRay q;        // the IBL ray (valid = the primary ray hit); q.o is the origin of BOTH occlusion rays
    bool need_sun;   // the surface faces the sun: a shadow ray decides vis_sun (else the sun term is 0)
    V3 sun_dir;      // direction of the shadow ray (the candidate's or the reservoir's: equal up to the last bit)
    V3 y;            // (albedo * light colour) * max(n . sun_dir, 0): the sun term before visibility and reuse weight
};

// Everything of a sample's shading except the two occlusion rays; draws u1, u2 from `rng` on a hit; o.a = the miss
// radiance on a miss, o.target_pdf = the candidate weight.
F3D_HD ShadeSetup sample_shade_setup(const FrameParams &P, bool prev_valid, const PrimaryHit &ph, uint32_t &rng, SampleOut &o) {
    ShadeSetup su;
    IblRay &q = su.q;
    q.valid = false;
    q.o = q.d = q.b0 = V3{0.0f, 0.0f, 0.0f};
    q.key = 2.0f;
    q.t_stop = 3.0e38f;
    su.need_sun = false;
    su.sun_dir = su.y = V3{0.0f, 0.0f, 0.0f};
    o.a = o.b = V3{0.0f, 0.0f, 0.0f};
    o.target_pdf = 0.0f;
    if (ph.hit.kind == 0u) {
        o.a = env_radiance(P.env, ph.rd);
        return su;
    }
    const V3 n = ph.hit.n;
    V3 albedo = P.light.albedo;
    F3D_OPAQUE_UNIFORM(albedo.x);  // (render constants: their products are formed per sample, not held across the kernel)
    F3D_OPAQUE_UNIFORM(albedo.y);
    F3D_OPAQUE_UNIFORM(albedo.z);
    if (ph.hit.kind != 1u) albedo = V3{0.7f, 0.7f, 0.8f};
    const V3 so = along(ph.hit.p, 1e-3f, n);
    // candidate generation for this hit (:500-512)
    o.target_pdf = luminance((albedo * P.light.color) * f_max(dot(n, P.light.wi), 0.0f));
    // sun through the merged reservoir, :517-532
    // (component selects: `cond ? P.light.wi_reuse : P.light.wi` became a select of two kernarg ADDRESSES and a vector load)
    uint32_t from_reservoir = prev_valid ? 1u : 0u;
    F3D_OPAQUE(from_reservoir);  // (per lane but the same in every round: selected here, not kept in three registers)
    su.sun_dir = V3{from_reservoir ? P.light.wi_reuse.x : P.light.wi.x, from_reservoir ? P.light.wi_reuse.y : P.light.wi.y,
                    from_reservoir ? P.light.wi_reuse.z : P.light.wi.z};
    const float nd = f_max(dot(n, su.sun_dir), 0.0f);
    if (nd > 0.0f) {
        su.need_sun = true;
        su.y = (albedo * P.light.color) * nd;
    }
    // one cosine-weighted IBL sample, :537-545
    const float u1 = rng_next(rng);
    const float u2 = rng_next(rng);
    const V3 ei = cosine_dir(n, u1, u2);
    q.valid = true;
    q.o = so;
    q.d = ei;
    q.b0 = albedo * env_radiance(P.env, ei);
    q.key = dot(n, ei);
#if !defined(F3D_NO_IBL_STOP)  // A/B builds
    if (ph.hit.kind == 1u) q.t_stop = ibl_stop(P.terrain, so, ei);  // (a mesh hit may lie below the terrain: no certificate)
#endif
    return su;
}

// First half of the shading of a sample whose primary hit is known (:486-536): candidate, sun term
// (with its shadow ray), and the IBL ray to trace; draws u1, u2 from `rng` on a hit.
// reuse_w(): the reuse weight of the frame head, asked for AFTER the shadow ray (the sample-lane kernels keep it in the
// lane's LDS column, f3d_frame.h, instead of in a register across the march).
template <class Pending, class ReuseW>
F3D_HD IblRay sample_shade_sun(const FrameParams &P, bool prev_valid, ReuseW reuse_w, const PrimaryHit &ph, uint32_t &rng,
                               SampleOut &o, Pending &pend) {
    const ShadeSetup su = sample_shade_setup(P, prev_valid, ph, rng, o);
    if (su.need_sun) {
        float vis = 1.0f;
#if defined(F3D_MODEL_HINT_SUN)  // scheduling-model builds of the emulator only
        pend.hint(F3D_MODEL_HINT_SUN);
#endif
        float sun_stop = ph.sun_tmax;
#if defined(F3D_SUN_HORIZON)  // A/B (round 6): the DEM-block far horizons (f3d_cone.h ibl_stop, built with F3D_IBL_HORIZON=1) for the
        // sun rays too -- a curved sun ray lies above the straight ray the horizon bounds, so the stop stays conservative
        if (ph.hit.kind == 1u) sun_stop = f_min(sun_stop, ibl_stop(P.terrain, su.q.o, su.sun_dir));
#endif
#if !defined(F3D_TIMING_NO_SHADOW)  // timing experiment only (wrong image): tools/gpu_build_ab.sh
        if (P.light.shadows_enabled != 0u && occluded(P, su.q.o, 1e-3f, su.sun_dir, 1e30f, true, pend, sun_stop)) vis = 0.0f;
#endif
        o.a = (su.y * vis) * reuse_w();
    }
#if defined(F3D_MODEL_HINT)  // scheduling-model builds of the emulator only (tools/march_model.py predictor)
    if (su.q.valid) pend.hint(F3D_MODEL_HINT);
#endif
    return su.q;
}
template <class Pending>
F3D_HD IblRay sample_shade_sun(const FrameParams &P, const FrameHead &h, const PrimaryHit &ph, uint32_t &rng,
                               SampleOut &o, Pending &pend) {
    const float w = h.reuse_w;
    return sample_shade_sun(P, h.prev_valid, [w]() F3D_LAMBDA { return w; }, ph, rng, o, pend);
}

// The verdict of an IBL ray (intersect_ibl_occlusion_ray, hybrid_traversal.wgsl:250-259).
template <class Pending>
F3D_HD bool ibl_occluded(const FrameParams &P, V3 o, V3 d, Pending &pend, float t_stop = 3.0e38f) {
#if defined(F3D_TIMING_NO_IBL)  // timing experiment only (wrong image): tools/gpu_build_ab.sh, profiles/README.md
    return false;
#else
    return occluded(P, o, 1e-3f, d, 1e30f, false, pend, t_stop);
#endif
}

// Shading of one sample on one lane: both halves back to back.
template <class Pending>
F3D_HD SampleOut sample_shade(const FrameParams &P, const FrameHead &h, const PrimaryHit &ph, uint32_t &rng,
                              Pending &pend) {
    SampleOut o;
    const IblRay q = sample_shade_sun(P, h, ph, rng, o, pend);
    if (q.valid) o.b = q.b0 * (ibl_occluded(P, q.o, q.d, pend, q.t_stop) ? 0.0f : 1.0f);
    return o;
}

// The order-sensitive part of a sample: candidate reservoir (:500-512) and radiance sum.  A miss
// adds (env, +0): x + (+0) == x for every x that can occur here (sums of non-negative terms).
F3D_HD void accumulate_sample(Reservoir &cand, V3 &radiance, V3 a, V3 b, float target_pdf) {
    if (target_pdf > 0.0f) {
        cand.directional = true;
        cand.w_sum = cand.w_sum + target_pdf;
        cand.m = cand.m + 1u;
        cand.target_pdf = target_pdf;
    }
    radiance = (radiance + a) + b;
}

// Everything after the sample loop (:549-574) plus the temporal pass; returns Welford m2.
F3D_HD float frame_tail(const FrameParams &P, uint32_t gx, uint32_t gy, Reservoir cand, V3 radiance) {
    const size_t lp = (size_t)(gy - P.row_begin) * P.cam.width + gx;
    const float fspp = (float)P.spp;
    radiance = V3{radiance.x / fspp, radiance.y / fspp, radiance.z / fspp};
    if (cand.m > 0u && cand.w_sum > 0.0f && cand.target_pdf > 0.0f)
        cand.weight = restir_div(cand.w_sum, (float)cand.m * cand.target_pdf);
    const size_t ri = reservoir_index(P, gx, gy);
    const Reservoir prev = unpack(P.res_out[ri]);  // parked by frame_head
    P.res_out[ri] = pack(temporal_merge(prev, cand));

    float4 am = P.accum_mean[lp];
    float wf_m2 = P.welford_m2[lp];
    float wf_mean = am.w;
    const V3 acc = V3{am.x + radiance.x, am.y + radiance.y, am.z + radiance.z};
    const float count = (float)(P.frame_index + 1u);  // accum.a: one per frame, exact
    const uint32_t phase = P.frame_index % kWelfordWindow;
    if (phase == 0u) {
        wf_mean = 0.0f;
        wf_m2 = 0.0f;
    }
    const float mean_lum = luminance(V3{acc.x / count, acc.y / count, acc.z / count});
    const float k = (float)phase + 1.0f;
    const float delta = mean_lum - wf_mean;
    const float mean = wf_mean + delta / k;
    const float m2 = f_fma(delta, mean_lum - mean, wf_m2);
    P.accum_mean[lp] = float4{acc.x, acc.y, acc.z, mean};
    P.welford_m2[lp] = m2;
    return m2;
}

// ---- frames in flight: the per-pixel pieces (f3d_kernels.hip k_trace / k_merge / k_fix) -----------
// A frame's records: two float4 per (sample, pixel) -- {a without the reuse weight, target pdf} and {b, code} with
// code = (hit ? 1 : 0) + (the head was PREDICTED to read the reservoir's sun direction ? 2 : 0).
F3D_HD float trace_code(bool hit, bool predicted_valid) { return (hit ? 1.0f : 0.0f) + (predicted_valid ? 2.0f : 0.0f); }
F3D_HD bool trace_code_hit(float code) { return code == 1.0f || code == 3.0f; }

// What k_trace leaves for one pixel-frame, one sample after the other on one lane (the sample-lane form is
// f3d_kernels.hip trace_lanes): frame_pixel's sample loop with reuse_w = 1 and the predicted sun-direction choice.
template <class Pending>
F3D_HD void trace_pixel(const FrameParams &P, uint32_t frame, uint32_t gx, uint32_t gy, bool predicted_valid, float4 *rec,
                        size_t pixels, Pending &pend) {
    FrameHead h;
    h.centre_hit = false;
    h.prev_valid = predicted_valid;
    h.reuse_w = 1.0f;
    h.rng = P.cam.seed_hi ^ (gx * 1664525u) ^ (gy * 1013904223u) ^ (frame * 92837111u) ^ P.cam.seed_lo;
    uint32_t rng = h.rng;
    for (uint32_t s = 0u; s < P.spp; s++) {
        const PrimaryHit ph = sample_primary(P, gx, gy, rng, pend);
        rng = ph.rng;
        const SampleOut o = sample_shade(P, h, ph, rng, pend);
        rec[2u * (size_t)s * pixels] = float4{o.a.x, o.a.y, o.a.z, o.target_pdf};
        rec[2u * (size_t)s * pixels + 1u] = float4{o.b.x, o.b.y, o.b.z, trace_code(ph.hit.kind != 0u, predicted_valid)};
    }
}

// Was the sun direction of this pixel-frame predicted wrong (for a sample that used it)?
F3D_HD bool merge_mispredicted(const FrameParams &P, const FrameHead &h, const float4 *rec, size_t pixels) {
    bool redo = false;
    for (uint32_t s = 0u; s < P.spp; s++) {
        const float code = rec[2u * (size_t)s * pixels + 1u].w;
        if (trace_code_hit(code) && (code == 3.0f) != h.prev_valid) redo = true;
    }
    return redo;
}

// The ordered half of a pixel-frame whose records are right: the samples through accumulate_sample with the sun term
// multiplied by the real reuse weight (what sample_shade_sun does at that point in k_frame), then the tail.
F3D_HD float merge_pixel(const FrameParams &P, uint32_t gx, uint32_t gy, const FrameHead &h, const float4 *rec, size_t pixels) {
    V3 radiance = V3{0.0f, 0.0f, 0.0f};
    Reservoir cand = empty_reservoir();
    for (uint32_t s = 0u; s < P.spp; s++) {
        const float4 r0 = rec[2u * (size_t)s * pixels], r1 = rec[2u * (size_t)s * pixels + 1u];
        V3 a = V3{r0.x, r0.y, r0.z};
        if (trace_code_hit(r1.w)) a = a * h.reuse_w;
        accumulate_sample(cand, radiance, a, V3{r1.x, r1.y, r1.z}, r0.w);
    }
    return frame_tail(P, gx, gy, cand, radiance);
}

// Both for a sample count known at compile time (k_merge, spp = 8: the strip driver's case).  A merge wave is nothing but
// memory latency -- the head's reservoirs, then 2 records per sample that k_trace left in HBM -- and with the count in a
// register the loop waits for every sample's records in turn (8 round trips); unrolled, all of a pixel-frame's record loads
// are in flight at once.  `redo` = the pixel-frame was mispredicted (nothing is merged then).  Same operations, same order.
template <uint32_t N>
F3D_HD float merge_pixel_n(const FrameParams &P, uint32_t gx, uint32_t gy, const FrameHead &h, const float4 *rec, size_t pixels, bool check, bool &redo) {
    float4 r0[N], r1[N];
#pragma unroll
    for (uint32_t s = 0u; s < N; s++) {
        r0[s] = rec[2u * (size_t)s * pixels];
        r1[s] = rec[2u * (size_t)s * pixels + 1u];
    }
    redo = false;
    if (check) {
#pragma unroll
        for (uint32_t s = 0u; s < N; s++)
            if (trace_code_hit(r1[s].w) && (r1[s].w == 3.0f) != h.prev_valid) redo = true;
    }
    if (redo) return 0.0f;
    V3 radiance = V3{0.0f, 0.0f, 0.0f};
    Reservoir cand = empty_reservoir();
#pragma unroll
    for (uint32_t s = 0u; s < N; s++) {
        V3 a = V3{r0[s].x, r0[s].y, r0[s].z};
        if (trace_code_hit(r1[s].w)) a = a * h.reuse_w;
        accumulate_sample(cand, radiance, a, V3{r1[s].x, r1[s].y, r1[s].z}, r0[s].w);
    }
    return frame_tail(P, gx, gy, cand, radiance);
}

// A mispredicted pixel-frame again: head (idempotent), the pixel's primary and sun rays with the direction the real
// head reads -- frame_pixel's sample loop, the IBL terms taken from the records -- and the tail.
template <class Pending>
F3D_HD float fix_pixel(const FrameParams &P, uint32_t gx, uint32_t gy, const float4 *rec, size_t pixels, Pending &pend) {
    const FrameHead h = frame_head(P, gx, gy);
    uint32_t rng = h.rng;
    V3 radiance = V3{0.0f, 0.0f, 0.0f};
    Reservoir cand = empty_reservoir();
    for (uint32_t s = 0u; s < P.spp; s++) {
        const PrimaryHit ph = sample_primary(P, gx, gy, rng, pend);
        rng = ph.rng;
        SampleOut o;
        (void)sample_shade_sun(P, h, ph, rng, o, pend);
        const float4 r1 = rec[2u * (size_t)s * pixels + 1u];
        accumulate_sample(cand, radiance, o.a, V3{r1.x, r1.y, r1.z}, o.target_pdf);
    }
    return frame_tail(P, gx, gy, cand, radiance);
}

// The sample loop of main_terrain (:476-548): primary, sun-shadow and IBL-occlusion rays per sample,
// one sample after the other on one lane.
// (A per-lane ray state machine with ballot-gated shading transitions was measured at 0.5-0.7x of
// this nested form on MI355X and removed -- profiles/README.md.)
template <class Pending>
F3D_HD float frame_pixel(const FrameParams &P, uint32_t gx, uint32_t gy, Pending &pend) {
    const FrameHead h = frame_head(P, gx, gy);
    uint32_t rng = h.rng;
    V3 radiance = V3{0.0f, 0.0f, 0.0f};
    Reservoir cand = empty_reservoir();
    for (uint32_t s = 0u; s < P.spp; s++) {
        const PrimaryHit ph = sample_primary(P, gx, gy, rng, pend);
        rng = ph.rng;
        const SampleOut o = sample_shade(P, h, ph, rng, pend);
        accumulate_sample(cand, radiance, o.a, o.b, o.target_pdf);
    }
    return frame_tail(P, gx, gy, cand, radiance);
}

// ---- G-buffer + frame-0 AOVs: unjittered centre ray (:583-609, :619-644) ---------------
template <class Pending>
F3D_HD void gbuffer_pixel(const FrameParams &P, uint32_t gx, uint32_t gy, float4 *gbuffer_n, float *depth,
                          Pending &pend) {
    const size_t lp = (size_t)(gy - P.row_begin) * P.cam.width + gx;
    const V3 rd = camera_dir(P.cam, gx, gy, 0.0f, 0.0f);
    if (P.primary_start) {
        const PrimaryStart ps = primary_start(P, gx, gy);
        P.primary_start[lp] = uint2{f_bits(ps.t_clear), ps.level};
    }
    const SurfaceHit hit = closest_hit(P, P.cam.origin, 1e-3f, rd, 1e30f, pend);
    if (P.sun_clear) {
        float2 c = float2{3.0e38f, 0.0f};
        if (hit.kind != 0u) c = float2{sun_clear_from(P, along(hit.p, 1e-3f, hit.n), hit.t), hit.t};
        P.sun_clear[lp] = c;
    }
    if (hit.kind != 0u) {
        gbuffer_n[lp] = float4{hit.n.x, hit.n.y, hit.n.z, (float)hit.kind};
        depth[lp] = hit.t;
    } else {
        gbuffer_n[lp] = float4{0.0f, 0.0f, 1.0f, 0.0f};
        depth[lp] = f_from_bits(0x7fc00000u);
    }
}

// ---- final resolve: last spatial pass + validity (render_terrain.rs:1313-1337),
// Reinhard -> RGBA16F -> u8 (hybrid_kernel.wgsl:109-112, render_terrain.rs:1358-1366),
// AOVs through RGBA16F (render_terrain.rs:438-447, :1367-1393). flags: bit0 valid, bit1 bad.
// aether != null && aether->enabled: the AETHER aerial-perspective post (f3d_aether.h) replaces the plain Reinhard.
F3D_HD uint32_t resolve_pixel(const FrameParams &P, uint32_t frames, uint32_t gx, uint32_t gy, uint8_t *rgba,
                              float *albedo, float *normal, const AetherDev *aether = nullptr,
                              const float *depth = nullptr) {
    const size_t lp = (size_t)(gy - P.row_begin) * P.cam.width + gx;
    const float4 g = P.gbuffer_n[lp];
    const Reservoir r = spatial_reuse<true>(P, P.res_in, gx, gy, frames - 1u, V3{g.x, g.y, g.z});
    uint32_t flags = 0u;
    if (!(f_finite(r.w_sum) && f_finite(r.weight) && f_finite(r.target_pdf))) flags |= 2u;
    if (r.m > 0u && r.weight > 0.0f && r.target_pdf > 0.0f) flags |= 1u;

    const float4 am = P.accum_mean[lp];
    const float count = (float)frames;
    const V3 mean = V3{am.x / count, am.y / count, am.z / count};
    float ldr[3];
    if (aether && aether->enabled != 0u) {
        // the unjittered pixel ray as prometheus_aerial.wgsl:113-121 forms it
        const float ndc_x = (((float)gx + 0.5f) / (float)P.cam.width) * 2.0f - 1.0f;
        const float ndc_y = (1.0f - ((float)gy + 0.5f) / (float)P.cam.height) * 2.0f - 1.0f;
        const float aspect = (float)P.cam.width / (float)P.cam.height;
        const float sx = ndc_x * P.cam.half_h * aspect, sy = ndc_y * P.cam.half_h;
        const V3 ray = normalize(V3{P.cam.right.x * sx + P.cam.up.x * sy + P.cam.forward.x,
                                    P.cam.right.y * sx + P.cam.up.y * sy + P.cam.forward.y,
                                    P.cam.right.z * sx + P.cam.up.z * sy + P.cam.forward.z});
        const V3 c = aether_resolve(*aether, mean, depth[lp], g.w != 0.0f, ray, P.light.wi, P.cam.origin.y);
        ldr[0] = c.x;
        ldr[1] = c.y;
        ldr[2] = c.z;
    } else {
        const V3 e = mean * P.cam.exposure;
        ldr[0] = e.x / (1.0f + e.x);
        ldr[1] = e.y / (1.0f + e.y);
        ldr[2] = e.z / (1.0f + e.z);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float v = round_to_half(ldr[c]);
        rgba[4 * lp + c] = (uint8_t)(f_clamp(v, 0.0f, 1.0f) * 255.0f + 0.5f);
    }
    rgba[4 * lp + 3] = 255;

    const uint32_t kind = (uint32_t)g.w;
    const V3 a = kind == 1u ? P.light.albedo : (kind == 2u ? V3{0.7f, 0.7f, 0.8f} : V3{0.0f, 0.0f, 0.0f});
    const V3 n = kind != 0u ? V3{g.x, g.y, g.z} : V3{0.0f, 0.0f, 0.0f};
    albedo[3 * lp + 0] = round_to_half(a.x);
    albedo[3 * lp + 1] = round_to_half(a.y);
    albedo[3 * lp + 2] = round_to_half(a.z);
    normal[3 * lp + 0] = round_to_half(n.x);
    normal[3 * lp + 1] = round_to_half(n.y);
    normal[3 * lp + 2] = round_to_half(n.z);
    return flags;
}

}  // namespace f3d
