// forge3d_amd/csrc/f3d_march.h
// Occlusion (any-hit) rays: stackless min-max march along the ray.
//
// The reference answers "is anything in the way?" with the same sorted quadtree descent it
// uses for closest hits (`terrain_trace(ray, any_hit = true, ...)`,
// hybrid_terrain_traversal.wgsl:254-372, called by intersect_shadow_ray /
// intersect_ibl_occlusion_ray, hybrid_traversal.wgsl:248-259).  For an any-hit ray best-t never
// changes before the function returns, so every node and every leaf is judged by a test that
// depends on the node and the ray only -- the answer is the OR over all leaves of
// "passes its slab interval, its height band and the leaf solve", whatever the visiting order.
// That freedom is used here: two thirds of all rays (sun shadow + IBL occlusion) and ~80 % of the
// traversal steps are any-hit.
//
// The march keeps ONE current node (level, x, z) -- the node of that level the ray is in -- and
// applies to it exactly the reference's per-node tests with the node's own slab interval
// (:288-304):
//   * band test fails  -> nothing in this node can be hit: step across its far boundary to the
//                         neighbour at the same level, and move one level UP whenever that
//                         crossing also leaves the parent (bigger steps while the ray is clear);
//   * band test passes -> level > 0: go DOWN into the child the ray is in (chosen by comparing
//                         the ray parameter with the child boundary's plane parameter);
//                         level 0: solve the leaf (:167-235); a hit ends the ray.
// No stack, no sorting, no four-children expansion: a step costs one 8-byte (min,max) fetch (or one
// 16-byte leaf record) and a few dozen VALU instructions, against ~200 for a descent step.
//
// Completeness (every leaf the descent would accept is reached): a node is skipped only when the
// reference's own band test for that node rejects it (which rejects every leaf inside it, because
// children are bounded by their parent in interval and height range); lateral moves follow the
// ray's exit boundary, so consecutive nodes tile the ray's path.  Two measure-zero deviations from
// the reference's enumeration are accepted and documented in DESIGN.md: a ray that leaves a node
// EXACTLY through a corner (both axis parameters equal in f32) skips the two cells it touches in
// that single point, and the node containing the ray's start is located from its slab
// parameters.  tests/ compare the boolean against the oracle on 75 000 proof rays and whole
// images bit for bit.
#pragma once

#include "f3d_trace.h"

namespace f3d {

// (min,max)*exaggeration of node (level, x, z); level 0 comes from the corner record.
template <class Pending>
F3D_HD void node_band(const TerrainDev &T, uint32_t level, uint32_t x, uint32_t z, Pending &pend, float &mn, float &mx,
                      LeafRec &leaf) {
    if (level == 0u) {
        leaf = T.leaves[tiled_index(x, z, T.tiles_x[0])];
        mn = min4(leaf);
        mx = max4(leaf);
    } else {
        uint32_t offset, tiles_x;
        pend.level_entry(T, level, offset, tiles_x);
        const NodeRec r = T.nodes[offset + tiled_index(x, z, tiles_x)];
        mn = r.mn;
        mx = r.mx;
    }
}

template <class Pending>
F3D_HD bool terrain_occluded_march(const TerrainDev &T, const RayCtx &r, Pending &pend) {
    const uint32_t top = T.mip_count - 1u;
    pend.note(3 | (r.c2 != 0.0f ? 4 : 0));  // statistics hook: a new any-hit ray starts
    // root slab interval (:288-297 for the root node)
    float t_cur;
    {
        const float ax = (plane_at(T.origin_x, 0u, T.spacing_x) - r.o.x) * r.inv_x;
        const float bx = (plane_at(T.origin_x, T.cell_w, T.spacing_x) - r.o.x) * r.inv_x;
        const float az = (plane_at(T.origin_z, 0u, T.spacing_z) - r.o.z) * r.inv_z;
        const float bz = (plane_at(T.origin_z, T.cell_h, T.spacing_z) - r.o.z) * r.inv_z;
        const float lo = f_max(f_max(f_min(ax, bx), f_min(az, bz)), r.tmin);
        const float hi = f_min(f_min(f_max(ax, bx), f_max(az, bz)), r.tmax);
        if (lo > hi) return false;
        t_cur = lo;
    }
    const bool x_forward = !(r.d.x < 0.0f), z_forward = !(r.d.z < 0.0f);
    // Start in the CELL the ray is in at t_cur instead of walking down from the root: for a
    // secondary ray (origin on the surface) the ~11 levels above its cell would all pass their
    // band tests anyway.  The cell is located from the position and then validated by its own
    // slab interval in the first iteration; if it does not contain t_cur (position rounded across
    // a cell boundary) the march restarts from the root.
    uint32_t level = 0u, nx, nz;
    {
        const float fx = f_floor((f_fma(t_cur, r.d.x, r.o.x) - T.origin_x) * T.inv_spacing_x);
        const float fz = f_floor((f_fma(t_cur, r.d.z, r.o.z) - T.origin_z) * T.inv_spacing_z);
        nx = sat_u32(fx);
        nz = sat_u32(fz);
        nx = nx < T.cell_w - 1u ? nx : T.cell_w - 1u;
        nz = nz < T.cell_h - 1u ? nz : T.cell_h - 1u;
    }
    bool unverified_start = true;
    for (;;) {
        pend.note(0);
        // node extent in cells, clamped at ragged edges (:282-286)
        const uint32_t cx0 = nx << level, cz0 = nz << level;
        uint32_t cx1 = (nx + 1u) << level, cz1 = (nz + 1u) << level;
        cx1 = cx1 < T.cell_w ? cx1 : T.cell_w;
        cz1 = cz1 < T.cell_h ? cz1 : T.cell_h;
        const float tx0 = (plane_at(T.origin_x, cx0, T.spacing_x) - r.o.x) * r.inv_x;
        const float tx1 = (plane_at(T.origin_x, cx1, T.spacing_x) - r.o.x) * r.inv_x;
        const float tz0 = (plane_at(T.origin_z, cz0, T.spacing_z) - r.o.z) * r.inv_z;
        const float tz1 = (plane_at(T.origin_z, cz1, T.spacing_z) - r.o.z) * r.inv_z;
        const float x_out = f_max(tx0, tx1), z_out = f_max(tz0, tz1);
        if (unverified_start) {
            unverified_start = false;
            const float enter = f_max(f_min(tx0, tx1), f_min(tz0, tz1));
            if (!(enter <= t_cur && t_cur <= f_min(x_out, z_out))) {  // not the cell the ray is in
                level = top;
                nx = 0u;
                nz = 0u;
                continue;
            }
        }
        const float lo = f_max(f_max(f_min(tx0, tx1), f_min(tz0, tz1)), r.tmin);
        const float hi = f_min(f_min(x_out, z_out), r.tmax);
        bool skip = lo > hi;  // the ray misses this node altogether (:297)
        float mn, mx;
        LeafRec leaf{};
        if (!skip) {
            node_band(T, level, nx, nz, pend, mn, mx, leaf);
            skip = band_rejects(r, lo, hi, mn, mx);  // :301-304
        }
        if (!skip) {
            if (level == 0u) {
                float t;
                if (leaf_solve(T, r, leaf, nx, nz, lo, hi, true, t) && t < r.tmax) return true;
                skip = true;  // leaf done: move on along the ray
            } else {
                // descend into the child the ray is in at t_cur: it has crossed the child boundary
                // plane iff that plane's parameter is <= t_cur
                const uint32_t cl = level - 1u;
                const uint32_t xm = (2u * nx + 1u) << cl, zm = (2u * nz + 1u) << cl;
                uint32_t ix = x_forward ? 0u : 1u, iz = z_forward ? 0u : 1u;  // entry-side child
                if (xm < T.cell_w) {
                    const float txm = (plane_at(T.origin_x, xm, T.spacing_x) - r.o.x) * r.inv_x;
                    if (txm <= t_cur) ix ^= 1u;
                } else {
                    ix = 0u;  // the far half is outside the cell grid
                }
                if (zm < T.cell_h) {
                    const float tzm = (plane_at(T.origin_z, zm, T.spacing_z) - r.o.z) * r.inv_z;
                    if (tzm <= t_cur) iz ^= 1u;
                } else {
                    iz = 0u;
                }
                nx = 2u * nx + ix;
                nz = 2u * nz + iz;
                level = cl;
                continue;
            }
        }
        // ---- step across the exit boundary of this node ----
        const float t_exit = f_min(x_out, z_out);
        if (!(t_exit < r.tmax)) return false;
        const bool cross_x = x_out <= z_out, cross_z = z_out <= x_out;
        const uint32_t px = nx, pz = nz;
        if (cross_x) {
            if (x_forward) {
                nx = nx + 1u;
                if ((nx << level) >= T.cell_w) return false;
            } else {
                if (nx == 0u) return false;
                nx = nx - 1u;
            }
        }
        if (cross_z) {
            if (z_forward) {
                nz = nz + 1u;
                if ((nz << level) >= T.cell_h) return false;
            } else {
                if (nz == 0u) return false;
                nz = nz - 1u;
            }
        }
        t_cur = f_max(t_cur, t_exit);
        // leaving the parent as well: continue one level up (the parent-level neighbour is new)
        if (level < top && ((nx >> 1) != (px >> 1) || (nz >> 1) != (pz >> 1))) {
            nx >>= 1;
            nz >>= 1;
            level = level + 1u;
        }
    }
}

}  // namespace f3d
