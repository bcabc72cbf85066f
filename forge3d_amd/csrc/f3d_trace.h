// forge3d_amd/csrc/f3d_trace.h
// Min-max quadtree ray traversal of the heightfield, re-designed for CDNA4.
//
// Same results as the reference's `terrain_trace`
// (src/shaders/hybrid_terrain_traversal.wgsl:254-372) -- same nodes culled, same leaves
// solved over the same [t_lo, t_hi] in the same order, hence the same hit t and normal --
// but a different machine:
//
//  * A node is tested when it is PUSHED, not when it is popped: visiting a node loads its
//    four children with one 32-byte (nodes) or 64-byte (leaf corners) vector fetch, applies
//    to each child exactly the tests the reference applies when it pops that child
//    (cell range :284, slab/t-interval :288-297 with the current best t, height band
//    :301-304), and only survivors are kept.  One dependent memory round trip per visited
//    node instead of one per child, and rejected children never cost a stack slot.
//    When a closer hit shrinks best-t after a node was queued, the slab test is redone at
//    pop time; the band test need not be (children are bounded by their parent in both
//    interval and height range, so they all reject -- see DESIGN.md "Traversal
//    equivalence").
//  * Level-1 nodes are "fat leaves": their (up to) four cells are solved in the
//    reference's near-to-far order straight from the 64-byte corner record, so level 0
//    never touches the stack and the reference's second fetch of the same four heights
//    for the normal (:239-248) disappears.
//  * No per-lane node stack.  Siblings that still have to be visited are a 2-bit child
//    code each; per level there are at most three, stored as one word in a
//    [level][lane] LDS column (bank = lane, conflict-free) with the remaining counts of
//    all levels in one register.  Node coordinates are recomputed from the current path.
//    (The reference's 64-entry private u32 stack would be 16 KiB per wave.)
#pragma once

#include "f3d_build.h"
#include "f3d_scene.h"

namespace f3d {

struct RayCtx {
    V3 o, d;
    float tmin, tmax;
    float inv_x, inv_z;  // terrain_safe_inv of d.x / d.z (:88-91)
    float c2;            // dot(d.xz, d.xz) * inv_two_r_prime when the curvature policy is on, else 0
    float vertex;        // parameter of the parabola's minimum (:120)
    bool has_vertex;
    // March only (f3d_march.h): the height above which this ray can stop -- the maximum of the whole terrain for a ray
    // that only climbs (d.y > 0; the curvature policy lifts it further, c2 >= 0), 3e38 otherwise.  Heights are monotone
    // in t (correctly rounded fma), so once the ray is above it every cell still ahead fails its band test (:301-304).
    float y_exit;
};

F3D_HD float safe_inv(float d) {
    float ad = f_max(f_abs(d), 1e-12f);
    return d < 0.0f ? -1.0f / ad : 1.0f / ad;
}

F3D_HD RayCtx make_ray(const TerrainDev &T, V3 o, float tmin, V3 d, float tmax, bool apply_curvature) {
    RayCtx r;
    r.o = o;
    r.d = d;
    r.tmin = tmin;
    r.tmax = tmax;
    r.inv_x = safe_inv(d.x);
    r.inv_z = safe_inv(d.z);
    bool curved = apply_curvature && T.curvature_enabled != 0u;
    float hd2 = dot2(d.x, d.z, d.x, d.z);
    r.c2 = curved ? hd2 * T.inv_two_r_prime : 0.0f;
    r.has_vertex = curved && r.c2 > 0.0f;
    r.vertex = r.has_vertex ? -d.y / (2.0f * r.c2) : 0.0f;
#if !defined(F3D_NO_ASCEND_EXIT)  // A/B builds (tools/build_variant.sh)
    r.y_exit = (d.y > 0.0f && r.c2 >= 0.0f && T.bands) ? T.bands[T.band_offset[T.mip_count - 1u]].mx : 3.0e38f;
#else
    r.y_exit = 3.0e38f;
#endif
    return r;
}

// terrain_curved_height (:95-103)
F3D_HD float height_at(const RayCtx &r, float t) { return f_fma(t * t, r.c2, f_fma(t, r.d.y, r.o.y)); }

// terrain_curved_height_range + the band rejection (:108-127, :301-304)
F3D_HD bool band_rejects(const RayCtx &r, float t0, float t1, float mn, float mx) {
    float y0 = height_at(r, t0), y1 = height_at(r, t1);
    float lo = f_min(y0, y1);
    if (r.has_vertex && r.vertex >= t0 && r.vertex <= t1) lo = f_min(lo, height_at(r, r.vertex));
    float hi = f_max(y0, y1);
    return lo > mx || hi < mn;
}

F3D_HD float plane_at(float origin, uint32_t cell, float spacing) { return f_fma((float)cell, spacing, origin); }

template <class T>
F3D_HD T pick4(uint32_t code, T a0, T a1, T a2, T a3) {
    T lo = (code & 1u) ? a1 : a0;
    T hi = (code & 1u) ? a3 : a2;
    return (code & 2u) ? hi : lo;
}

// ray_triangle_intersect, hybrid_traversal.wgsl:86-132
F3D_HD bool ray_triangle(V3 o, float tmin, V3 d, float tmax, V3 v0, V3 v1, V3 v2, float &t_out, V3 &n_out) {
    V3 e1 = v1 - v0, e2 = v2 - v0;
    V3 h = cross(d, e2);
    float a = dot(e1, h);
    if (f_abs(a) < 1e-7f) return false;
    float f = 1.0f / a;
    V3 s = o - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = cross(s, e1);
    float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = f * dot(e2, q);
    if (t > tmin && t < tmax) {
        t_out = t;
        n_out = normalize(cross(e1, e2));
        return true;
    }
    return false;
}

// Exact ray / bilinear patch solve (terrain_leaf_intersect, :167-235) on a corner record.
F3D_HD bool leaf_solve(const TerrainDev &T, const RayCtx &r, const LeafRec &h, uint32_t cx, uint32_t cz, float t0,
                       float t1, bool any_hit, float &t_hit) {
    float tm = 0.5f * (t0 + t1);
    float fx = -(float)cx, fz = -(float)cz;
    float dv[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float t = (i == 0) ? t0 : (i == 1) ? tm : t1;
        float px = f_fma(t, r.d.x, r.o.x);
        float pz = f_fma(t, r.d.z, r.o.z);
        float u = f_clamp(f_fma(px - T.origin_x, T.inv_spacing_x, fx), 0.0f, 1.0f);
        float v = f_clamp(f_fma(pz - T.origin_z, T.inv_spacing_z, fz), 0.0f, 1.0f);
        float hh = mix(mix(h.h00, h.h10, u), mix(h.h01, h.h11, u), v);
        dv[i] = height_at(r, t) - hh;
    }
    float c = dv[0];
    float a = 2.0f * dv[2] + 2.0f * dv[0] - 4.0f * dv[1];
    float b = dv[2] - dv[0] - a;
    float s_hit = 1e30f;
    if (any_hit && c <= 0.0f) {
        s_hit = 0.0f;
    } else if (f_abs(a) < 1e-12f) {
        if (f_abs(b) > 1e-12f) {
            float s = -c / b;
            if (s >= 0.0f && s <= 1.0f) s_hit = s;
        }
    } else {
        float four_ac = 4.0f * a * c;
        float disc = f_fma(b, b, -four_ac);
        if (disc >= 0.0f) {
            float sq = f_sqrt(disc);
            float q = -0.5f * (b + (b >= 0.0f ? sq : -sq));
            float r0 = q / a;
            float r1 = (f_abs(q) < 1e-30f) ? 1e30f : c / q;
            float lo = f_min(r0, r1), hi = f_max(r0, r1);
            if (lo >= 0.0f && lo <= 1.0f) s_hit = lo;
            else if (hi >= 0.0f && hi <= 1.0f) s_hit = hi;
        }
    }
    if (s_hit <= 1.0f) {
        float t = f_fma(s_hit, t1 - t0, t0);
        if (t > r.tmin && t < r.tmax) {
            t_hit = t;
            return true;
        }
    }
    return false;
}

// terrain_normal_at (:239-248) from the corner record already in registers.
F3D_HD V3 leaf_normal(const TerrainDev &T, const LeafRec &h, V3 p, uint32_t cx, uint32_t cz) {
    float u = f_clamp(f_fma(p.x - T.origin_x, T.inv_spacing_x, -(float)cx), 0.0f, 1.0f);
    float v = f_clamp(f_fma(p.z - T.origin_z, T.inv_spacing_z, -(float)cz), 0.0f, 1.0f);
    float dh_du = mix(h.h10 - h.h00, h.h11 - h.h01, v);
    float dh_dv = mix(h.h01 - h.h00, h.h11 - h.h10, u);
    return normalize(V3{-dh_du * T.inv_spacing_x, 1.0f, -dh_dv * T.inv_spacing_z});
}

// The four children of node (level, nx, nz): slab entry/exit per child and whether the
// reference would have queued it (cell range :332, clipped interval :342-344).
struct ChildSlabs {
    float enter[4], exit[4];
    float key[4];  // ct_lo, the reference's sort key (:345)
    bool queued[4];
};

F3D_HD ChildSlabs child_slabs(const TerrainDev &T, const RayCtx &r, uint32_t level, uint32_t nx, uint32_t nz,
                              float t_lo, float t_hi) {
    const uint32_t cl = level - 1u;
    const uint32_t x0 = (2u * nx) << cl, xm = (2u * nx + 1u) << cl, x1 = (2u * nx + 2u) << cl;
    const uint32_t z0 = (2u * nz) << cl, zm = (2u * nz + 1u) << cl, z1 = (2u * nz + 2u) << cl;
    const bool right_ok = xm < T.cell_w, lower_ok = zm < T.cell_h;
    const uint32_t xmc = xm < T.cell_w ? xm : T.cell_w, x1c = x1 < T.cell_w ? x1 : T.cell_w;
    const uint32_t zmc = zm < T.cell_h ? zm : T.cell_h, z1c = z1 < T.cell_h ? z1 : T.cell_h;
    const float tx0 = (plane_at(T.origin_x, x0, T.spacing_x) - r.o.x) * r.inv_x;
    const float txm = (plane_at(T.origin_x, xmc, T.spacing_x) - r.o.x) * r.inv_x;
    const float tx1 = (plane_at(T.origin_x, x1c, T.spacing_x) - r.o.x) * r.inv_x;
    const float tz0 = (plane_at(T.origin_z, z0, T.spacing_z) - r.o.z) * r.inv_z;
    const float tzm = (plane_at(T.origin_z, zmc, T.spacing_z) - r.o.z) * r.inv_z;
    const float tz1 = (plane_at(T.origin_z, z1c, T.spacing_z) - r.o.z) * r.inv_z;
    const float ex_lo[2] = {f_min(tx0, txm), f_min(txm, tx1)}, ex_hi[2] = {f_max(tx0, txm), f_max(txm, tx1)};
    const float ez_lo[2] = {f_min(tz0, tzm), f_min(tzm, tz1)}, ez_hi[2] = {f_max(tz0, tzm), f_max(tzm, tz1)};
    ChildSlabs k;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int ix = c & 1, iz = c >> 1;
        k.enter[c] = f_max(ex_lo[ix], ez_lo[iz]);
        k.exit[c] = f_min(ex_hi[ix], ez_hi[iz]);
        const float ct_lo = f_max(k.enter[c], t_lo), ct_hi = f_min(k.exit[c], t_hi);
        k.key[c] = ct_lo;
        k.queued[c] = (ix == 0 || right_ok) && (iz == 0 || lower_ok) && !(ct_lo > ct_hi);
    }
    return k;
}

// Visit order of the surviving children: the reference sorts by descending ct_lo with a
// stable insertion sort and pushes in that order (:351-369), so they POP by ascending
// ct_lo, and for equal keys the child scanned LATER pops first.  Returns the 2-bit child
// codes in pop order packed from bit 0, and the count.
F3D_HD uint32_t visit_order(const float key[4], const bool ok[4], uint32_t &count) {
    float k0 = ok[0] ? key[0] : __builtin_inff(), k1 = ok[1] ? key[1] : __builtin_inff();
    float k2 = ok[2] ? key[2] : __builtin_inff(), k3 = ok[3] ? key[3] : __builtin_inff();
    // b_ij (i < j): child j is visited before child i
    uint32_t b01 = k1 <= k0, b02 = k2 <= k0, b03 = k3 <= k0, b12 = k2 <= k1, b13 = k3 <= k1, b23 = k3 <= k2;
    uint32_t r0 = b01 + b02 + b03;
    uint32_t r1 = (1u - b01) + b12 + b13;
    uint32_t r2 = (2u - b02 - b12) + b23;
    uint32_t r3 = 3u - b03 - b13 - b23;
    count = (uint32_t)ok[0] + (uint32_t)ok[1] + (uint32_t)ok[2] + (uint32_t)ok[3];
    // invalid children have key +inf; among equal +inf keys the later index ranks first, so
    // invalid entries may interleave only among themselves at ranks >= count ... unless a
    // valid key is +inf too, which cannot happen (ct_lo <= ct_hi <= tmax < inf).
    (void)r0;
    return (1u << (2u * r1)) | (2u << (2u * r2)) | (3u << (2u * r3));
}

struct TraceHit {
    float t;
    V3 n;
    bool hit;
};

// Resumable traversal: trace_begin applies the root tests, every trace_step visits one node
// (inner node or fat leaf).  The frame kernel interleaves steps of rays in different
// phases (primary / sun shadow / IBL) across the lanes of a wave.
struct TraceState {
    TraceHit res;
    uint32_t level, nx, nz;  // node being visited / last visited (path for sibling decoding)
    uint32_t remaining;      // 2 bits per level: siblings still queued
    float t_lo, t_hi;        // clipped interval of the node to visit
    bool have;               // (level, nx, nz) with [t_lo, t_hi] is ready to visit
    bool done;
};

template <class Pending>
F3D_HD void trace_begin(const TerrainDev &T, const RayCtx &r, bool any_hit, TraceState &st, Pending &pend) {
    pend.note(2 | (any_hit ? 1 : 0) | (r.c2 != 0.0f ? 4 : 0));  // statistics hook: a new ray starts
    st.res.hit = false;
    st.res.t = r.tmax;
    st.res.n = V3{0.0f, 0.0f, 0.0f};
    st.done = true;
    st.have = false;
    st.remaining = 0u;
    const uint32_t top = T.mip_count - 1u;
    st.level = top;
    st.nx = 0u;
    st.nz = 0u;
    // ---- root: the reference pops it first and applies :284-304 ----
    {
        const float ax = (plane_at(T.origin_x, 0u, T.spacing_x) - r.o.x) * r.inv_x;
        const float bx = (plane_at(T.origin_x, T.cell_w, T.spacing_x) - r.o.x) * r.inv_x;
        const float az = (plane_at(T.origin_z, 0u, T.spacing_z) - r.o.z) * r.inv_z;
        const float bz = (plane_at(T.origin_z, T.cell_h, T.spacing_z) - r.o.z) * r.inv_z;
        st.t_lo = f_max(f_max(f_min(ax, bx), f_min(az, bz)), r.tmin);
        st.t_hi = f_min(f_min(f_max(ax, bx), f_max(az, bz)), f_min(r.tmax, st.res.t));
        if (st.t_lo > st.t_hi) return;
    }
    if (top == 0u) {  // 2x2 DEM: the root is the single cell
        const LeafRec h = T.leaves[0];
        if (band_rejects(r, st.t_lo, st.t_hi, min4(h), max4(h))) return;
        float t;
        if (leaf_solve(T, r, h, 0u, 0u, st.t_lo, st.t_hi, any_hit, t) && t < st.res.t) {
            st.res.hit = true;
            st.res.t = t;
            st.res.n = leaf_normal(T, h, along(r.o, t, r.d), 0u, 0u);
        }
        return;
    }
    const NodeRec root = T.nodes[T.node_offset[top]];
    if (band_rejects(r, st.t_lo, st.t_hi, root.mn, root.mx)) return;
    st.have = true;
    st.done = false;
}

// MODE 0: visit whatever node is next.  MODE 1: inner nodes only (the caller guarantees the
// next node is not a fat leaf, or that a sibling has to be popped first).  MODE 2: the fat
// leaf that is ready (st.have && st.level == 1).  Modes 1/2 let trace_terrain run all lanes
// of a wave through their inner-node descents together and then through their leaf solves
// together ("while-while"), instead of serialising the two divergent bodies every step.
template <int MODE = 0, class Pending>
F3D_HD void trace_step(const TerrainDev &T, const RayCtx &r, bool any_hit, TraceState &st, Pending &pend) {
    if (MODE != 2 && !st.have) {
        if (st.remaining == 0u) {
            st.done = true;
            return;
        }
        // deepest level with queued siblings
        const uint32_t l = (uint32_t)__builtin_ctz(st.remaining) >> 1;
        const uint32_t left = (st.remaining >> (2u * l)) & 3u;
        const uint32_t word = pend.get(l);
        const uint32_t total = word >> 6;
        const uint32_t code = (word >> (2u * (total - left))) & 3u;
        st.remaining -= 1u << (2u * l);
        const uint32_t up = l + 1u - st.level;  // levels between the current node and the parent
        st.nx = ((st.nx >> up) << 1) | (code & 1u);
        st.nz = ((st.nz >> up) << 1) | (code >> 1);
        st.level = l;
        // pop-time slab/interval test with the CURRENT best t (:288-297)
        const uint32_t cx0 = st.nx << l, cz0 = st.nz << l;
        uint32_t cx1 = (st.nx + 1u) << l, cz1 = (st.nz + 1u) << l;
        cx1 = cx1 < T.cell_w ? cx1 : T.cell_w;
        cz1 = cz1 < T.cell_h ? cz1 : T.cell_h;
        const float ax = (plane_at(T.origin_x, cx0, T.spacing_x) - r.o.x) * r.inv_x;
        const float bx = (plane_at(T.origin_x, cx1, T.spacing_x) - r.o.x) * r.inv_x;
        const float az = (plane_at(T.origin_z, cz0, T.spacing_z) - r.o.z) * r.inv_z;
        const float bz = (plane_at(T.origin_z, cz1, T.spacing_z) - r.o.z) * r.inv_z;
        st.t_lo = f_max(f_max(f_min(ax, bx), f_min(az, bz)), r.tmin);
        st.t_hi = f_min(f_min(f_max(ax, bx), f_max(az, bz)), f_min(r.tmax, st.res.t));
        if (st.t_lo > st.t_hi) return;  // culled; next step pops again
        if (MODE == 1 && l == 1u) {     // a fat leaf surfaced: leave it for the leaf phase
            st.have = true;
            return;
        }
    }
    st.have = false;
    const uint32_t level = st.level, nx = st.nx, nz = st.nz;
    const uint32_t cl = level - 1u;
    const ChildSlabs k = child_slabs(T, r, level, nx, nz, st.t_lo, st.t_hi);

    if (MODE != 1 && (MODE == 2 || cl == 0u)) {
        // ---- fat leaf: solve the queued cells near-to-far (:306-318) ----
        pend.note(1);
        const uint32_t g = child_group_index(nx, nz, T.tiles_x[0]);
        const LeafRec h0 = T.leaves[g], h1 = T.leaves[g + 1u], h2 = T.leaves[g + 2u], h3 = T.leaves[g + 3u];
        uint32_t count;
        const uint32_t order = visit_order(k.key, k.queued, count);
        for (uint32_t i = 0u; i < count; i++) {
            const uint32_t code = (order >> (2u * i)) & 3u;
            const float en = pick4(code, k.enter[0], k.enter[1], k.enter[2], k.enter[3]);
            const float ex = pick4(code, k.exit[0], k.exit[1], k.exit[2], k.exit[3]);
            const float lo = f_max(en, r.tmin);
            const float hi = f_min(ex, f_min(r.tmax, st.res.t));
            if (lo > hi) continue;
            LeafRec h;
            h.h00 = pick4(code, h0.h00, h1.h00, h2.h00, h3.h00);
            h.h10 = pick4(code, h0.h10, h1.h10, h2.h10, h3.h10);
            h.h01 = pick4(code, h0.h01, h1.h01, h2.h01, h3.h01);
            h.h11 = pick4(code, h0.h11, h1.h11, h2.h11, h3.h11);
            if (band_rejects(r, lo, hi, min4(h), max4(h))) continue;
            const uint32_t cx = 2u * nx + (code & 1u), cz = 2u * nz + (code >> 1);
            float t;
            if (leaf_solve(T, r, h, cx, cz, lo, hi, any_hit, t) && t < st.res.t) {
                st.res.hit = true;
                st.res.t = t;
                st.res.n = leaf_normal(T, h, along(r.o, t, r.d), cx, cz);
                if (any_hit) {
                    st.done = true;
                    return;
                }
            }
        }
        return;
    }

    // ---- inner node: test the four children now, keep the survivors (:320-369) ----
    pend.note(0);
    uint32_t level_offset, level_tiles_x;
    pend.level_entry(T, cl, level_offset, level_tiles_x);
    const uint32_t g = level_offset + child_group_index(nx, nz, level_tiles_x);
    // four (min, max) records = 32 contiguous, 32-byte aligned bytes: two 16-byte loads
    const float4 *pair = reinterpret_cast<const float4 *>(T.nodes + g);
    const float4 q0 = pair[0], q1 = pair[1];
    const float cmn[4] = {q0.x, q0.z, q1.x, q1.z}, cmx[4] = {q0.y, q0.w, q1.y, q1.w};
    bool keep[4];
    const float cap = f_min(r.tmax, st.res.t);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float lo = f_max(k.enter[c], r.tmin), hi = f_min(k.exit[c], cap);
        keep[c] = k.queued[c] && !(lo > hi) && !band_rejects(r, lo, hi, cmn[c], cmx[c]);
    }
    uint32_t count;
    const uint32_t order = visit_order(k.key, keep, count);
    if (count == 0u) return;
    // nearest survivor is visited next; the rest wait in the level's pending word
    const uint32_t first = order & 3u;
    if (count > 1u) {
        pend.put(cl, (order >> 2) | ((count - 1u) << 6));
        st.remaining |= (count - 1u) << (2u * cl);
    }
    st.t_lo = f_max(pick4(first, k.enter[0], k.enter[1], k.enter[2], k.enter[3]), r.tmin);
    st.t_hi = f_min(pick4(first, k.exit[0], k.exit[1], k.exit[2], k.exit[3]), cap);
    st.nx = 2u * nx + (first & 1u);
    st.nz = 2u * nz + (first >> 1);
    st.level = cl;
    st.have = true;
}

// LDS (device) or array (host) column holding the pending-sibling word of each level.
// put/get are only called with 1 <= level < kMaxLevels.
template <class Pending>
F3D_HD TraceHit trace_terrain(const TerrainDev &T, const RayCtx &r, bool any_hit, Pending &pend) {
    TraceState st;
    trace_begin(T, r, any_hit, st, pend);
#if defined(F3D_TRACE_WHILE_WHILE)
    // Measured on MI355X (profiles/README.md): batching the leaf solves ("while-while") is
    // 1.5x SLOWER than visiting whatever comes next -- lanes holding a leaf idle through the
    // other lanes' descents -- so it is kept only as an A/B switch.
    while (!st.done) {
        while (!st.done && !(st.have && st.level == 1u)) trace_step<1>(T, r, any_hit, st, pend);
        if (!st.done) trace_step<2>(T, r, any_hit, st, pend);
    }
#elif defined(F3D_TRACE_LEAF_GATE)
    // Leaf gating (A/B switch, measured SLOWER: 2312 -> 1603 Msamples/s as the quorum goes
    // 1 -> 32, profiles/README.md): a lane holding a fat leaf waits until `leaf_quorum` lanes
    // of its wave hold one too, or no lane has inner-node work left.
    while (!st.done) {
        const bool at_leaf = st.have && st.level == 1u;
        const bool open = pend.leaf_gate(at_leaf);
        if (!at_leaf) trace_step<1>(T, r, any_hit, st, pend);
        else if (open) trace_step<2>(T, r, any_hit, st, pend);
    }
#else
    while (!st.done) trace_step<0>(T, r, any_hit, st, pend);
#endif
    return st.res;
}

}  // namespace f3d
