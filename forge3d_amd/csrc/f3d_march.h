// forge3d_amd/csrc/f3d_march.h
// Stackless min-max march: the traversal the frame kernel uses for every ray.
//
// The reference answers both "what does this ray hit first?" and "is anything in the way?" with a
// sorted quadtree descent over an explicit stack (`terrain_trace`,
// hybrid_terrain_traversal.wgsl:254-372).  Its RESULT, however, is a function of per-node and
// per-leaf tests that depend on the node and the ray only:
//   * a node is entered iff its own slab interval [max(enter,tmin), min(exit,tmax,best)] is not
//     empty (:288-297) and the ray's height range over it meets the node's (min,max) band (:301-304);
//   * a leaf inside entered nodes is solved over its own interval (:167-235);
//   * any-hit rays return at the first hit (best-t never changes before that, so the answer is the
//     OR over all leaves, whatever the order); closest-hit rays visit leaves near-to-far (children are
//     sorted by entry parameter, :351-369), so the first hit along the ray is final -- a later leaf
//     starts where the earlier one ended and `t < best` is strict.
// So any enumeration that walks the nodes ALONG THE RAY, applies those same tests with the same
// plane parameters, and stops at the first hit, returns the same hit/t/normal.  The march does that
// with ONE current node (level, x, z) and no stack:
//   band test fails -> step across the node's exit boundary to the neighbour of the same level, and
//                      one level UP whenever that crossing also leaves the parent;
//   band test passes -> level > 0: DOWN into the child the ray is in (the child boundary's plane
//                      parameter against the current ray parameter); level 0: solve the leaf.
// A step costs one 8-byte (min,max) fetch from a row-major per-level table and a few dozen VALU
// instructions; the descent's four-children expansion, sorting network and LDS sibling lists are gone.
// Secondary rays start in the cell their origin is in (validated by that cell's own slab interval)
// instead of walking ~11 levels down from the root.
//
// Corners.  A ray that passes EXACTLY through a lattice corner (equal f32 plane parameters: diagonal
// rays over square DEMs, the unjittered centre ray of a camera on the diagonal, a 45-degree sun) touches
// the two cells beside the corner in that single point; the reference visits them with the zero-length
// interval [T, T] (:288-297 keeps lo == hi).  Such a visit can only answer an ANY-hit ray (the closest-hit
// solve finds no root on a zero-length interval: a = b = 0), and it does: `c <= 0 -> hit` (:197-201) fires
// for rays that run below the surface there.  The march detects every exact corner passage -- leaving a
// node through its own corner (x_out == z_out), or entering a node on the mid-plane of the node it
// descends (t_mid == t_cur with the entry plane's parameter equal too) -- and queues a TIE entry; the
// drain judges the two side cells by their own band test and the zero-length solve (march_drain).
// The start cell is located from the ray position and validated by its own slab interval.  tests/
// compare hit, t and normal with the oracle on 75 000 proof rays, on adversarial lattice-aligned ray
// sets (tests/test_adversarial_march.py) and whole renders bit for bit.
#pragma once

#include "f3d_trace.h"

namespace f3d {

template <bool CURVED>
F3D_HD float march_height(const RayCtx &r, float t) {
    const float lin = f_fma(t, r.d.y, r.o.y);
    // non-curved rays have c2 == 0: fma(t*t, 0, lin) == lin exactly, so the term is dropped
    return CURVED ? f_fma(t * t, r.c2, lin) : lin;
}

template <bool CURVED>
F3D_HD bool march_band_rejects(const RayCtx &r, float t0, float t1, float mn, float mx) {
#if !defined(F3D_PACKED_F32)
    const float y0 = march_height<CURVED>(r, t0), y1 = march_height<CURVED>(r, t1);
#else
    const F2 tt = f2(t0, t1);
    F2 yy = fma2(tt, f2(r.d.y, r.d.y), f2(r.o.y, r.o.y));                // march_height at both ends at once
    if (CURVED) yy = fma2(tt * tt, f2(r.c2, r.c2), yy);
    const float y0 = yy.x, y1 = yy.y;
#endif
    float lo = f_min(y0, y1);
    if (CURVED) {
        if (r.has_vertex && r.vertex >= t0 && r.vertex <= t1) lo = f_min(lo, march_height<true>(r, r.vertex));
    }
    return (lo > mx) | (f_max(y0, y1) < mn);
}

// Leaf solves are DEFERRED.  The solve (~170 VALU instructions with its divisions and square
// root) is wanted by ~5 % of the lane-steps, but in a 64-lane wave some lane wants it in almost every
// iteration, so an in-line solve makes every iteration pay for it at 1-3 active lanes.  Instead a lane
// that reaches a leaf whose band test passes appends (cell, lo, hi) to a small per-lane FIFO and keeps
// marching as if the leaf had missed; the wave drains the FIFOs together when enough lanes have
// something queued (Ctx::flush_now, a ballot), when a FIFO is full, or when nobody marches any more.
// This is exact: before the first hit every leaf is judged with best-t = tmax, so its verdict does
// not depend on when it is evaluated; any-hit rays need the OR, closest-hit rays the FIRST queued
// leaf (ray order = FIFO order) that hits.  The price is a few extra march steps for rays whose hit
// is sitting in the FIFO.
#ifndef F3D_LEAF_FIFO
#define F3D_LEAF_FIFO 4
#endif
constexpr uint32_t kLeafFifo = F3D_LEAF_FIFO;  // entries per lane (A/B: 2, 3, 6, 8 -- profiles/README.md)
// A step queues at most two entries (a leaf AND the corner it leaves through) and the wave drains as soon
// as one lane holds kLeafFifo: storage for one more.
constexpr uint32_t kLeafFifoRows = kLeafFifo + 1u;
// An entry is {cell, lo, hi}.  -DF3D_FIFO_WORDS=1 (round 4, measured and NOT adopted) keeps the cell only -- or the corner
// of a TIE -- and lets the drain form the interval again, by the expressions the step used on the same integers (a TIE's
// parameter T is the plane parameter of its corner's x line: every site that queues one has just compared that very
// value for equality with T).  Bit-identical (the emulator hands the drain NaNs for lo / hi in that build), 4 352 bytes of
// LDS a wave instead of 6 144, i.e. room for 7 and 8 waves per SIMD (LDS is handed out in 1 280-byte pieces:
// tools/experiments/lds_granule.hip) -- but the ~20 instructions per drained leaf cost more than the waves bring:
// 8 680 (3 words, 6 waves) / 8 450 (1 word, 6) / 8 610 (1 word, 7) / 8 330 (1 word, 8) Msamples/s on one box.
#ifndef F3D_FIFO_WORDS
#define F3D_FIFO_WORDS 3
#endif
constexpr uint32_t kFifoWords = F3D_FIFO_WORDS;
static_assert(kFifoWords == 1u || kFifoWords == 3u, "F3D_FIFO_WORDS: 1 (cell only) or 3 (cell, lo, hi)");
// A lane takes up to kStepsPerVote march steps between two wave votes (the flush / share / done ballots and their
// branches are about a seventh of an iteration's serial latency); it stops early when its ray ends or its FIFO could
// overflow (a step queues at most two entries).  Results do not depend on it (3.1 of DESIGN.md: verdicts do not depend
// on when leaves are evaluated).  Measured on the headline frame: 1 / 2 / 4 / 6 / 8 / 12 / 16 steps -> 6 989 / 7 270 /
// 7 422 / 7 529 / 7 514 / 7 368 / 7 095 Msamples/s (profiles/r03_variant_ab.log); 4 inside the ray-sharing rounds.
// Round 4, after the spills went: 5 / 6 / 8 / 10 -> 8 713 / 8 690 / 8 750 / lower (profiles/r04_variant_ab.log): 8.
#ifndef F3D_STEPS_PER_VOTE
#define F3D_STEPS_PER_VOTE 8
#endif
constexpr uint32_t kStepsPerVote = F3D_STEPS_PER_VOTE;
#ifndef F3D_STEPS_PER_VOTE_SHARED
#define F3D_STEPS_PER_VOTE_SHARED 4
#endif
constexpr uint32_t kStepsPerVoteShared = F3D_STEPS_PER_VOTE_SHARED;  // ... inside the ray-sharing rounds (march_shared)
// TIE entry (see "Corners" above): corner lattice point (X, Z) <= 8192 in 14 bits each, the ray's x / z
// direction, which side cell is next, and the flag; its `lo` word holds the corner's ray parameter T.
constexpr uint32_t kTieFlag = 0x80000000u, kTieSecond = 0x40000000u, kTieZFwd = 0x20000000u, kTieXFwd = 0x10000000u;
F3D_HD uint32_t tie_entry(uint32_t X, uint32_t Z, bool x_forward, bool z_forward) {
    return kTieFlag | (z_forward ? kTieZFwd : 0u) | (x_forward ? kTieXFwd : 0u) | (Z << 14) | X;
}

// Per-lane position of a march.
struct MarchState {
    float t_cur;
    uint32_t level, nx, nz;
    bool marching, unverified_start;
#if !defined(F3D_NO_BAND_PREFETCH)
    float band_mn, band_mx;  // the (min,max) band of the node the lane stands in, fetched when it moved there
    float mesh_mn, mesh_mx;  // FUSE: the node's mesh band (f3d_meshgrid.h)
#endif
};

// The band of a node is the one memory access of a step, and the step cannot decide anything before it has arrived.
// The lane therefore asks for it as soon as it knows where it goes next -- at the END of the previous step -- and the
// wave votes, the FIFO bookkeeping and the next step's plane arithmetic run while it is on its way
// (-DF3D_NO_BAND_PREFETCH: fetched where it is needed, the round-2 form; profiles/README.md).
// FUSE (the occlusion rays of the kernels compiled for scenes with a mesh, round 6): the scene's mesh is a second (min, max) band
// of every node (f3d_meshgrid.h) and the march looks for both at once -- it descends where either band passes, and a cell whose
// mesh band passes is queued with kMeshCell (with kMeshOnly when the terrain's own band rejects it): the drain puts the cell's
// triangles through the sweep's ray_triangle.  The terrain's verdicts are untouched (its leaves are queued and solved exactly
// when its own band passes); the mesh's are conservative by the grid's construction and exact in the triangle test.  Measured
// before it was built: a second band per step costs the configs[3] frame 0.4 ms, the occlusion rays' tree walks 8.9 ms.
constexpr uint32_t kMeshCell = 0x4000u, kMeshOnly = 0x8000u;  // (cell = cx | cz << 16 with cx, cz < 2^13: bits 13-15 are free)
template <bool FUSE = false, class Ctx>
F3D_HD void march_fetch(const TerrainDev &T, MarchState &m, Ctx &ctx) {
#if !defined(F3D_NO_BAND_PREFETCH)
    uint32_t band_offset, band_shift;
    ctx.band_entry(T, m.level, band_offset, band_shift);
    const NodeRec band = T.bands[band_offset + (m.nz << band_shift) + m.nx];
    m.band_mn = band.mn;
    m.band_mx = band.mx;
    if (FUSE) {
        const NodeRec mesh = T.mesh_bands[band_offset + (m.nz << band_shift) + m.nx];
        m.mesh_mn = mesh.mn;
        m.mesh_mx = mesh.mx;
    }
#else
    static_assert(!FUSE, "the fused mesh band rides with the band prefetch");
#endif
}

// Root slab interval of a ray (:288-297 for the root node): [lo, hi], empty when lo > hi.
F3D_HD void march_root_interval(const TerrainDev &T, const RayCtx &r, float &lo, float &hi) {
    const float ax = (plane_at(T.origin_x, 0u, T.spacing_x) - r.o.x) * r.inv_x;
    const float bx = (plane_at(T.origin_x, T.cell_w, T.spacing_x) - r.o.x) * r.inv_x;
    const float az = (plane_at(T.origin_z, 0u, T.spacing_z) - r.o.z) * r.inv_z;
    const float bz = (plane_at(T.origin_z, T.cell_h, T.spacing_z) - r.o.z) * r.inv_z;
    lo = f_max(f_max(f_min(ax, bx), f_min(az, bz)), r.tmin);
    hi = f_min(f_min(f_max(ax, bx), f_max(az, bz)), r.tmax);
}

// Put a lane at the start of its ray: at the root, or (secondary rays) in the cell the ray starts in,
// located from the position and validated by that cell's own slab interval on the first step.
F3D_HD MarchState march_begin(const TerrainDev &T, const RayCtx &r, bool start_in_cell) {
    MarchState m;
    float hi;
    march_root_interval(T, r, m.t_cur, hi);
    m.marching = !(m.t_cur > hi);
    m.level = T.mip_count - 1u;
    m.nx = 0u;
    m.nz = 0u;
    m.unverified_start = false;
    // (a ray that ENTERS the footprint, t_cur > tmin, may do so exactly through a lattice corner on the
    // boundary: only the walk down from the root sees the cell it touches there -- "Corners" above)
    if (start_in_cell && !(m.t_cur > r.tmin)) {
        const float fx = f_floor((f_fma(m.t_cur, r.d.x, r.o.x) - T.origin_x) * T.inv_spacing_x);
        const float fz = f_floor((f_fma(m.t_cur, r.d.z, r.o.z) - T.origin_z) * T.inv_spacing_z);
        m.nx = sat_u32(fx);
        m.nz = sat_u32(fz);
        m.nx = m.nx < T.cell_w - 1u ? m.nx : T.cell_w - 1u;
        m.nz = m.nz < T.cell_h - 1u ? m.nz : T.cell_h - 1u;
        m.level = 0u;
        m.unverified_start = true;
    }
    return m;
}

// One march step of a lane: test the current node and move DOWN, or ACROSS (+ UP).
// SLICED: the lane walks a slice of a ray (march_shared below): nodes entered at or beyond t_stop
// belong to the next slice.
// any_hit: the ray is an occlusion ray (corner ties matter, see the header); a constant at every call site.
// VERIFY: the lane may stand in a node that was located from a rounded position (m.unverified_start).  Only the FIRST
// step of a ray or slice can: the march loops take that step through the verifying instantiation (march_first_step)
// and every later one through VERIFY = false, which carries neither the flag nor its test.
template <bool CURVED, bool SLICED, bool VERIFY = true, bool FUSE = false, class Ctx>
F3D_HD void march_step(const TerrainDev &T, const RayCtx &r, MarchState &m, uint32_t &queued, Ctx &ctx, bool any_hit,
                       float t_stop = 3.0e38f) {
    ctx.note(0);
    const uint32_t top = T.mip_count - 1u;
    const uint32_t level = m.level, nx = m.nx, nz = m.nz;
    const bool x_forward = !(r.d.x < 0.0f), z_forward = !(r.d.z < 0.0f);
    // node extent in cells, clamped at ragged edges (:282-286), and its four plane parameters
    const uint32_t cx0 = nx << level, cz0 = nz << level;
    uint32_t cx1 = (nx + 1u) << level, cz1 = (nz + 1u) << level;
    cx1 = cx1 < T.cell_w ? cx1 : T.cell_w;
    cz1 = cz1 < T.cell_h ? cz1 : T.cell_h;
#if !defined(F3D_PACKED_F32)  // the shipped scalar form; packed f32 pairs measured 0.92x (5335 vs 5780 Msamples/s)
    const float tx0 = (plane_at(T.origin_x, cx0, T.spacing_x) - r.o.x) * r.inv_x;
    const float tx1 = (plane_at(T.origin_x, cx1, T.spacing_x) - r.o.x) * r.inv_x;
    const float tz0 = (plane_at(T.origin_z, cz0, T.spacing_z) - r.o.z) * r.inv_z;
    const float tz1 = (plane_at(T.origin_z, cz1, T.spacing_z) - r.o.z) * r.inv_z;
#else
    // (plane_at(origin, cell, spacing) - o) * inv for the x and the z plane of a corner at once (packed f32)
    const F2 grid_o = f2(T.origin_x, T.origin_z), grid_s = f2(T.spacing_x, T.spacing_z);
    const F2 ray_o = f2(r.o.x, r.o.z), ray_inv = f2(r.inv_x, r.inv_z);
    const F2 t_lo = (fma2(f2((float)cx0, (float)cz0), grid_s, grid_o) - ray_o) * ray_inv;
    const F2 t_hi = (fma2(f2((float)cx1, (float)cz1), grid_s, grid_o) - ray_o) * ray_inv;
    const float tx0 = t_lo.x, tz0 = t_lo.y, tx1 = t_hi.x, tz1 = t_hi.y;
#endif
    const float x_out = f_max(tx0, tx1), z_out = f_max(tz0, tz1);
    const float enter = f_max(f_min(tx0, tx1), f_min(tz0, tz1)), exit = f_min(x_out, z_out);
    if (VERIFY && m.unverified_start && !(enter <= m.t_cur && m.t_cur <= exit)) {
        // the position was rounded across a cell boundary: walk down from the root instead
        m.level = top;
        m.nx = 0u;
        m.nz = 0u;
    } else {
        const float lo = f_max(enter, r.tmin), hi = f_min(exit, r.tmax);
        // the node's (min,max) band: one 8-byte record of the row-major table of its level
#if !defined(F3D_NO_BAND_PREFETCH)
        const NodeRec band{m.band_mn, m.band_mx};
#else
        uint32_t band_offset, band_shift;
        ctx.band_entry(T, level, band_offset, band_shift);
        const NodeRec band = T.bands[band_offset + (nz << band_shift) + nx];
#endif
        const bool terrain_pass = !(lo > hi) & !march_band_rejects<CURVED>(r, lo, hi, band.mn, band.mx);  // :297-304 (bitwise: no branch)
        // (the mesh is met by the STRAIGHT ray: the curvature policy bends the terrain test only, hybrid_traversal.wgsl:204-259)
        const bool mesh_pass = FUSE ? (!(lo > hi) & !march_band_rejects<false>(r, lo, hi, m.mesh_mn, m.mesh_mx)) : false;
        const bool pass = terrain_pass | mesh_pass;
        if (pass) ctx.note(-1);  // statistics hook (host emulator only): the last step whose band test passed
#if defined(F3D_BRANCHLESS_STEP)
        // A/B (round 6, DESIGN.md 6 "plateau"): the step with NO data-dependent branch but the rare corner ties -- DOWN and ACROSS
        // are both formed and selected, the leaf goes into the FIFO's next free slot whether it is one or not (the slot counter
        // moves only for a leaf; the loops call a step only with two slots free).  Same values by the same expressions as the
        // branching form below.
        {
            static_assert(!FUSE, "the branch-free A/B form of the step has no fused mesh band");
            const bool down = pass & (level > 0u), leaf = pass & (level == 0u);
            const uint32_t cl = level > 0u ? level - 1u : 0u;
            const uint32_t xm = (2u * nx + 1u) << cl, zm = (2u * nz + 1u) << cl;
            const float txm = (plane_at(T.origin_x, xm, T.spacing_x) - r.o.x) * r.inv_x;
            const float tzm = (plane_at(T.origin_z, zm, T.spacing_z) - r.o.z) * r.inv_z;
            uint32_t ix = (x_forward != (txm <= m.t_cur)) ? 0u : 1u;
            uint32_t iz = (z_forward != (tzm <= m.t_cur)) ? 0u : 1u;
            ix = xm < T.cell_w ? ix : 0u;
            iz = zm < T.cell_h ? iz : 0u;
#if !defined(F3D_NO_CORNER_TIES)
            if (any_hit && down && (txm == m.t_cur || tzm == m.t_cur)) {  // (rare) on a child boundary: see the branching form
                uint32_t lv = level;
                F3D_OPAQUE(lv);
                const uint32_t ex1 = (nx + 1u) << lv, ez1 = (nz + 1u) << lv;
                const uint32_t xin = x_forward ? nx << lv : (ex1 < T.cell_w ? ex1 : T.cell_w);
                const uint32_t zin = z_forward ? nz << lv : (ez1 < T.cell_h ? ez1 : T.cell_h);
                const float txin = (plane_at(T.origin_x, xin, T.spacing_x) - r.o.x) * r.inv_x;
                const float tzin = (plane_at(T.origin_z, zin, T.spacing_z) - r.o.z) * r.inv_z;
                if (txm == m.t_cur && tzin == m.t_cur && xm < T.cell_w) {
                    ctx.fifo_put(queued, tie_entry(xm, zin, x_forward, z_forward), m.t_cur, m.t_cur);
                    queued++;
                } else if (tzm == m.t_cur && txin == m.t_cur && zm < T.cell_h) {
                    ctx.fifo_put(queued, tie_entry(xin, zm, x_forward, z_forward), m.t_cur, m.t_cur);
                    queued++;
                }
            }
#endif
            if (leaf) ctx.note(1);
            ctx.fifo_put(queued, nx | (nz << 16), lo, hi);  // (a free slot: harmless when this is no leaf)
            queued += leaf ? 1u : 0u;
            const bool cross_x = x_out <= z_out, cross_z = z_out <= x_out;
#if !defined(F3D_NO_CORNER_TIES)
            if (any_hit && !down && cross_x && cross_z && exit < r.tmax) {  // out through the node's own corner (rare)
                uint32_t lv = level;
                F3D_OPAQUE(lv);
                const uint32_t ex1 = (nx + 1u) << lv, ez1 = (nz + 1u) << lv;
                const uint32_t X = x_forward ? (ex1 < T.cell_w ? ex1 : T.cell_w) : nx << lv;
                const uint32_t Z = z_forward ? (ez1 < T.cell_h ? ez1 : T.cell_h) : nz << lv;
                ctx.fifo_put(queued, tie_entry(X, Z, x_forward, z_forward), exit, exit);
                queued++;
            }
#endif
            const uint32_t qx = nx + ((cross_x && x_forward) ? 1u : 0u) - ((cross_x && !x_forward) ? 1u : 0u);
            const uint32_t qz = nz + ((cross_z && z_forward) ? 1u : 0u) - ((cross_z && !z_forward) ? 1u : 0u);
            const bool left = !(exit < r.tmax) | (SLICED & !(exit < t_stop)) | ((qx << level) >= T.cell_w) |
                              ((qz << level) >= T.cell_h) | (march_height<CURVED && !FUSE>(r, exit) > r.y_exit);
            const bool up = level < top && (((qx ^ nx) | (qz ^ nz)) > 1u);
            m.nx = down ? 2u * nx + ix : (up ? qx >> 1 : qx);
            m.nz = down ? 2u * nz + iz : (up ? qz >> 1 : qz);
            m.level = down ? cl : (up ? level + 1u : level);
            m.t_cur = down ? m.t_cur : f_max(m.t_cur, exit);
            m.marching = down | !left;
        }
#else
        if (pass && level > 0u) {
            // DOWN into the child the ray is in at t_cur: it has passed the child boundary plane
            // iff that plane's parameter is <= t_cur
            const uint32_t cl = level - 1u;
            const uint32_t xm = (2u * nx + 1u) << cl, zm = (2u * nz + 1u) << cl;
#if !defined(F3D_PACKED_F32)
            const float txm = (plane_at(T.origin_x, xm, T.spacing_x) - r.o.x) * r.inv_x;
            const float tzm = (plane_at(T.origin_z, zm, T.spacing_z) - r.o.z) * r.inv_z;
#else
            const F2 t_mid = (fma2(f2((float)xm, (float)zm), grid_s, grid_o) - ray_o) * ray_inv;
            const float txm = t_mid.x, tzm = t_mid.y;
#endif
            uint32_t ix = (x_forward != (txm <= m.t_cur)) ? 0u : 1u;
            uint32_t iz = (z_forward != (tzm <= m.t_cur)) ? 0u : 1u;
            if (!(xm < T.cell_w)) ix = 0u;  // the far half lies outside the cell grid
            if (!(zm < T.cell_h)) iz = 0u;
#if !defined(F3D_NO_CORNER_TIES)  // A/B + test-of-the-tests switch: the round-1 behaviour
            if (any_hit && (txm == m.t_cur || tzm == m.t_cur)) {
                // the ray is ON a child boundary at its current parameter: if it is also on the plane it
                // entered this node through, it passes exactly through the lattice corner where both meet
                // (the half it does not enter is touched in that one point: "Corners" in the header).
                // Rare: the entry planes are recomputed here instead of being kept alive across the band fetch.
                uint32_t lv = level;
                F3D_OPAQUE(lv);
                const uint32_t ex1 = (nx + 1u) << lv, ez1 = (nz + 1u) << lv;
                const uint32_t xin = x_forward ? nx << lv : (ex1 < T.cell_w ? ex1 : T.cell_w);
                const uint32_t zin = z_forward ? nz << lv : (ez1 < T.cell_h ? ez1 : T.cell_h);
                const float txin = (plane_at(T.origin_x, xin, T.spacing_x) - r.o.x) * r.inv_x;
                const float tzin = (plane_at(T.origin_z, zin, T.spacing_z) - r.o.z) * r.inv_z;
                if (txm == m.t_cur && tzin == m.t_cur && xm < T.cell_w) {
                    ctx.fifo_put(queued, tie_entry(xm, zin, x_forward, z_forward), m.t_cur, m.t_cur);
                    queued++;
                } else if (tzm == m.t_cur && txin == m.t_cur && zm < T.cell_h) {
                    ctx.fifo_put(queued, tie_entry(xin, zm, x_forward, z_forward), m.t_cur, m.t_cur);
                    queued++;
                }
            }
#endif
            m.nx = 2u * nx + ix;
            m.nz = 2u * nz + iz;
            m.level = cl;
        } else {
            if (pass) {  // a leaf to solve: queue it and march on as if it had missed
                ctx.note(1);
                ctx.fifo_put(queued, nx | (nz << 16) | (FUSE ? (mesh_pass ? kMeshCell : 0u) | (terrain_pass ? 0u : kMeshOnly) : 0u), lo, hi);
                queued++;
            }
            // ---- across the exit boundary of this node (straight-line: no nested divergence) ----
            const bool cross_x = x_out <= z_out, cross_z = z_out <= x_out;
#if !defined(F3D_NO_CORNER_TIES)
            if (any_hit && cross_x && cross_z && exit < r.tmax) {  // out through the node's own corner (rare)
                uint32_t lv = level;
                F3D_OPAQUE(lv);
                const uint32_t ex1 = (nx + 1u) << lv, ez1 = (nz + 1u) << lv;
                const uint32_t X = x_forward ? (ex1 < T.cell_w ? ex1 : T.cell_w) : nx << lv;
                const uint32_t Z = z_forward ? (ez1 < T.cell_h ? ez1 : T.cell_h) : nz << lv;
                ctx.fifo_put(queued, tie_entry(X, Z, x_forward, z_forward), exit, exit);
                queued++;
            }
#endif
            // a backward step from column 0 wraps to 0xFFFFFFFF, whose shifted value is >= cell_w too
            // (cell_w <= 2^13, level <= 15), so one unsigned comparison covers both directions
            const uint32_t qx = nx + ((cross_x && x_forward) ? 1u : 0u) - ((cross_x && !x_forward) ? 1u : 0u);
            const uint32_t qz = nz + ((cross_z && z_forward) ? 1u : 0u) - ((cross_z && !z_forward) ? 1u : 0u);
            // above the whole terrain and climbing (RayCtx::y_exit): nothing ahead can pass its band test
            // (bitwise | on purpose: every term is a couple of vector compares, cheaper than the branches of a short-circuit)
            const bool left = !(exit < r.tmax) | (SLICED & !(exit < t_stop)) | ((qx << level) >= T.cell_w) |
                              ((qz << level) >= T.cell_h) | (march_height<CURVED && !FUSE>(r, exit) > r.y_exit);
            // leaving the parent as well: continue one level up (jumping h > 1 levels when the crossing
            // leaves h ancestors was modelled on the emulator's step logs: fewer IBL steps, but more
            // shadow steps and 7-17 % more wave iterations -- tools/march_model.py)
            const bool up = level < top && (((qx ^ nx) | (qz ^ nz)) > 1u);
            m.nx = up ? qx >> 1 : qx;
            m.nz = up ? qz >> 1 : qz;
            m.level = up ? level + 1u : level;
            m.t_cur = f_max(m.t_cur, exit);
            m.marching = !left;  // out of the footprint, or past tmax
        }
#endif
    }
    if (VERIFY) m.unverified_start = false;
    if (m.marching) march_fetch<FUSE>(T, m, ctx);
}
// The first step of the lanes whose start node still has to be validated (see VERIFY above).
template <bool CURVED, bool SLICED, bool FUSE = false, class Ctx>
F3D_HD void march_first_step(const TerrainDev &T, const RayCtx &r, MarchState &m, uint32_t &queued, Ctx &ctx, bool any_hit, float t_stop) {
#if !defined(F3D_VERIFY_EVERY_STEP)  // (A/B: the round-2 form tests the flag in every step)
    if (m.marching && m.unverified_start) march_step<CURVED, SLICED, true, FUSE>(T, r, m, queued, ctx, any_hit, t_stop);
#endif
}
#if !defined(F3D_VERIFY_EVERY_STEP)
constexpr bool kVerifyInLoop = false;
#else
constexpr bool kVerifyInLoop = true;
#endif

// Drain the lane's leaf FIFO: solve the queued leaves in ray order until one hits.  A TIE entry (any-hit
// rays only) stands for the two cells beside a lattice corner the ray passes exactly through: each is
// judged as the reference judges it -- cell range, the zero-length interval [T, T] clipped by the ray's
// (tmin, tmax), the cell's own (min,max) band (= min / max of its corner record; its ancestors' tests
// are implied, their intervals contain T and their bands contain the cell's), then the leaf solve.
// The slab interval of cell (cx, cz) clipped by the ray's (tmin, tmax), by the expressions march_step uses for a level-0 node.
F3D_HD void march_leaf_interval(const TerrainDev &T, const RayCtx &r, uint32_t cx, uint32_t cz, float &lo, float &hi) {
    const uint32_t cx1 = cx + 1u < T.cell_w ? cx + 1u : T.cell_w, cz1 = cz + 1u < T.cell_h ? cz + 1u : T.cell_h;
    const float tx0 = (plane_at(T.origin_x, cx, T.spacing_x) - r.o.x) * r.inv_x;
    const float tx1 = (plane_at(T.origin_x, cx1, T.spacing_x) - r.o.x) * r.inv_x;
    const float tz0 = (plane_at(T.origin_z, cz, T.spacing_z) - r.o.z) * r.inv_z;
    const float tz1 = (plane_at(T.origin_z, cz1, T.spacing_z) - r.o.z) * r.inv_z;
    lo = f_max(f_max(f_min(tx0, tx1), f_min(tz0, tz1)), r.tmin);
    hi = f_min(f_min(f_max(tx0, tx1), f_max(tz0, tz1)), r.tmax);
}
// hit_cell: the cell (cx | cz << 16) of the hit a closest-hit drain ends with (the sharing of closest-hit rays, march_shared_closest)
template <bool FUSE = false, class Ctx>
F3D_HD void march_drain(const TerrainDev &T, const RayCtx &r, bool any_hit, MarchState &m, uint32_t &queued,
                        TraceHit &res, Ctx &ctx, uint32_t &hit_cell) {
    uint32_t k = 0u;  // per lane: a tie entry is visited twice
    // (a plain divergent `while (k < queued && !res.hit)` measured 0.5 % slower than this vote per entry)
    while (ctx.any(k < queued && !res.hit)) {
        if (k < queued && !res.hit) {
            uint32_t cell;
            float lo, hi;
            ctx.fifo_get(k, cell, lo, hi);
            uint32_t cx = cell & (FUSE ? 0x1FFFu : 0xFFFFu), cz = cell >> 16;
            const bool tie = any_hit && (cell & kTieFlag) != 0u;
            bool solve = FUSE ? (tie || (cell & kMeshOnly) == 0u) : true;  // (a cell queued for its triangles only: no leaf to solve)
            const bool triangles = FUSE && !tie && (cell & kMeshCell) != 0u && T.mesh_cell_start != nullptr;
            if (kFifoWords == 1u) {  // the interval again, as march_step formed it (see kFifoWords)
                if (tie) {
                    lo = hi = (plane_at(T.origin_x, cell & 0x3FFFu, T.spacing_x) - r.o.x) * r.inv_x;
                } else {
                    march_leaf_interval(T, r, cx, cz, lo, hi);
                }
            }
            if (tie) {
                const uint32_t X = cell & 0x3FFFu, Z = (cell >> 14) & 0x3FFFu;
                const bool xf = (cell & kTieXFwd) != 0u, zf = (cell & kTieZFwd) != 0u, second = (cell & kTieSecond) != 0u;
                // first visit: the cell across the x plane in the row the ray comes from; second: the cell across
                // the z plane in the column it comes from (an index of -1 wraps and fails the range test)
                const uint32_t near_x = xf ? X - 1u : X, far_x = xf ? X : X - 1u;
                const uint32_t near_z = zf ? Z - 1u : Z, far_z = zf ? Z : Z - 1u;
                cx = second ? near_x : far_x;
                cz = second ? far_z : near_z;
                hi = f_min(lo, r.tmax);
                lo = f_max(lo, r.tmin);
                solve = cx < T.cell_w && cz < T.cell_h && !(lo > hi);
                if (!second) ctx.fifo_retag(k, cell | kTieSecond);
                else k++;
            } else {
                k++;
            }
            if (solve) {
                const LeafRec leaf = T.leaves[tiled_index(cx, cz, T.tiles_x[0])];
                float t;
                if (!(tie && band_rejects(r, lo, hi, min4(leaf), max4(leaf))) &&
                    leaf_solve(T, r, leaf, cx, cz, lo, hi, any_hit, t) && t < res.t) {
                    res.hit = true;  // first hit in ray order is final (see the header)
                    res.t = t;
                    res.n = leaf_normal(T, leaf, along(r.o, t, r.d), cx, cz);
                    hit_cell = cx | (cz << 16);
                }
            }
#if defined(F3D_NO_CELL_TRIANGLES)  // test of the tests: the fused march without its triangle tests must FAIL the mesh parity tests
            if (false) {
#else
            if (FUSE && triangles && !res.hit) {
#endif  // the cell's triangles through the sweep's own test (any hit: existence)
                const uint32_t c = cz * T.cell_w + cx;
                uint32_t e = T.mesh_cell_start[c];
                const uint32_t e_end = T.mesh_cell_start[c + 1u];
                for (; e < e_end; e++) {
                    const float4 a = T.mesh_cell_tris[3u * e], b = T.mesh_cell_tris[3u * e + 1u], c2 = T.mesh_cell_tris[3u * e + 2u];
                    float t;
                    V3 n;
                    if (ray_triangle(r.o, r.tmin, r.d, r.tmax, V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, V3{c2.x, c2.y, c2.z}, t, n)) {
                        res.hit = true;
                        res.t = t;
                        res.n = n;
                        break;
                    }
                }
            }
        }
    }
    queued = 0u;
    // (once, here: a second boolean carried through the loop above costs a lane-mask merge per level of nesting -- 1.4 %)
    m.marching = m.marching & !res.hit;
}
template <bool FUSE = false, class Ctx>
F3D_HD void march_drain(const TerrainDev &T, const RayCtx &r, bool any_hit, MarchState &m, uint32_t &queued,
                        TraceHit &res, Ctx &ctx) {
    uint32_t hit_cell;  // (nobody reads it: the stores go)
    march_drain<FUSE>(T, r, any_hit, m, queued, res, ctx, hit_cell);
}

// ---- the last few rays of a wave, shared by all its lanes -----------------------------------------
// The step counts of the IBL-occlusion rays are heavy-tailed: in the step logs of the headline frame
// 31 % of the IBL wave iterations run with ONE lane still marching, 46 % with at most two, 60 % with
// at most four (tools/march_model.py tail) -- a quarter of the whole frame.  An any-hit answer is the
// OR over the leaves along the ray, and every node / leaf verdict uses the node's own slab interval
// clipped by the RAY's (tmin, tmax) only, so a ray can be cut into SLICES walked by different lanes:
// a slice walks the nodes whose entry parameter is below `stop`, starting at the node of a given level
// that contains the ray at its `begin` (located from the position and validated by that node's own
// interval, like the in-cell start of secondary rays; the node that contains a cut is visited by both
// neighbours, none is skipped).  So when at most kShareBelow (16) lanes of the wave still march, the wave
// drains its leaf FIFOs and deals every surviving ray over its lanes as slices with geometric boundaries
// (the steps grow with the distance walked), tagged with the owner's lane; slices that hit mark the
// owner on a verdict board in LDS; when again only a few slices are left they are dealt again (up to
// kShareRounds times).  Same measure-zero caveat as the in-cell start: ancestors of a slice's start node
// are not consulted.  Offered to the rays WITHOUT the curvature policy only, i.e. the IBL rays: their
// directions differ per lane anyway, whereas the sun rays of a wave share one direction whose constants
// live in scalar registers as long as the ray is loop-invariant (and their tail is short: 17 % of the
// shadow iterations run with <= 4 lanes).
#ifndef F3D_SHARE_BELOW
#define F3D_SHARE_BELOW 16  // (with kShareAvail = 4 lanes per ray, a 64-lane wave never deals more than 16: larger values change nothing)
#endif
#ifndef F3D_SHARE_ROUNDS
#define F3D_SHARE_ROUNDS 10
#endif
#ifndef F3D_SHARE_AVAIL
#define F3D_SHARE_AVAIL 4
#endif
constexpr uint32_t kShareBelow = F3D_SHARE_BELOW, kShareRounds = F3D_SHARE_ROUNDS;
constexpr uint32_t kShareAvail = F3D_SHARE_AVAIL;  // ... and at least this many lanes of the call per marching ray

// Level-`level` node containing the ray at parameter t (clamped into the grid; the caller validates it).
F3D_HD void march_locate(const TerrainDev &T, const RayCtx &r, float t, uint32_t level, uint32_t &nx, uint32_t &nz) {
    const float fx = f_floor((f_fma(t, r.d.x, r.o.x) - T.origin_x) * T.inv_spacing_x);
    const float fz = f_floor((f_fma(t, r.d.z, r.o.z) - T.origin_z) * T.inv_spacing_z);
    nx = sat_u32(fx);
    nz = sat_u32(fz);
    nx = nx < T.cell_w - 1u ? nx : T.cell_w - 1u;
    nz = nz < T.cell_h - 1u ? nz : T.cell_h - 1u;
    nx >>= level;
    nz >>= level;
}

// A lane's share of the dealt rays: the ray, where the slice stops, and whose ray it is.
struct MarchSlice {
    RayCtx r;
    float t_stop, t_end;  // t_end: where the ray leaves the footprint
    uint32_t owner;       // lane that wants the verdict
};

// Deal the marching slices of a wave over the lanes of the call: lane i of the call works on ray (i / per), slice
// (i % per) of its remaining range; `per` = the largest power of two that fits.  Written against the wave
// primitives of Ctx (ballot / shfl / lane / fast_log2 / fast_exp2) so that the device (f3d_kernels.hip LdsPending)
// and the 64-lane host emulator (tests/emul) run the SAME dealing code.
#ifndef F3D_SHARE_LEVEL_GAIN
#define F3D_SHARE_LEVEL_GAIN 1.0f
#endif
F3D_HD uint32_t bits_set(unsigned long long v) { return (uint32_t)__builtin_popcountll(v); }
template <bool CURVED, class Ctx>
F3D_HD void march_deal(const TerrainDev &T, MarchSlice &s, MarchState &m, const Ctx &ctx) {
    const unsigned long long active = ctx.ballot(true), mask = ctx.ballot(m.marching);
    const unsigned long long below = (1ull << ctx.lane()) - 1ull;
    const uint32_t n = bits_set(mask);
    if (n == 0u) {
        m.marching = false;
        return;
    }
    const uint32_t avail = bits_set(active), rank = bits_set(active & below);
    // (dealing avail / n slices per ray instead of the power of two below measured slower: 5843 vs 6058)
    const uint32_t sh = 31u - (uint32_t)__builtin_clz(avail / n), per = 1u << sh;
    const uint32_t q = rank >> sh, k = rank & (per - 1u);
    const bool take = q < n;
    int src = (int)ctx.lane();  // lanes without a slice read their OWN registers below (always an active lane)
    {
        unsigned long long rest = mask;
        for (uint32_t i = 0u; i < n; i++) {  // wave-uniform, n <= the sharing threshold
            const int b = __builtin_ffsll((long long)rest) - 1;
            rest &= rest - 1ull;
            if (q == i) src = b;
        }
    }
    RayCtx r;
    r.o = V3{ctx.shfl(s.r.o.x, src), ctx.shfl(s.r.o.y, src), ctx.shfl(s.r.o.z, src)};
    r.d = V3{ctx.shfl(s.r.d.x, src), ctx.shfl(s.r.d.y, src), ctx.shfl(s.r.d.z, src)};
    r.tmin = ctx.shfl(s.r.tmin, src);
    r.tmax = ctx.shfl(s.r.tmax, src);
    r.inv_x = ctx.shfl(s.r.inv_x, src);
    r.inv_z = ctx.shfl(s.r.inv_z, src);
    if (CURVED) {
        r.c2 = ctx.shfl(s.r.c2, src);
        r.vertex = ctx.shfl(s.r.vertex, src);
        r.has_vertex = ctx.shfl((uint32_t)s.r.has_vertex, src) != 0u;
    } else {
        r.c2 = 0.0f;
        r.vertex = 0.0f;
        r.has_vertex = false;
    }
    r.y_exit = ctx.shfl(s.r.y_exit, src);
    const float t0 = ctx.shfl(m.t_cur, src), t_end = ctx.shfl(s.t_end, src);
    const float stop_src = ctx.shfl(s.t_stop, src);
    const float t1 = f_min(stop_src, t_end);  // the slice being cut again ends here
    const uint32_t owner = ctx.shfl(s.owner, src);
    const uint32_t level = ctx.shfl(m.level, src), nx = ctx.shfl(m.nx, src), nz = ctx.shfl(m.nz, src);
    // `per` slices of [t0, t1] with GEOMETRIC boundaries t0 (t1/t0)^(k/per): the march's steps grow with
    // the ray's clearance, i.e. roughly with the distance from its origin, so equal-t slices would leave
    // almost all the work in the first one.  Any boundaries are valid; slice k begins exactly where slice
    // k - 1 stops (same expression, same inputs); the last slice inherits the stop of the slice it cuts.
    const float base = f_max(t0, 1e-3f * f_max(t1, 1e-30f));  // t0 can be ~0 for a ray that has barely started
    const float lg = ctx.fast_log2(f_max(t1, base) / base) / (float)per;
    const float begin = k == 0u ? t0 : base * ctx.fast_exp2(lg * (float)k);
    const float stop = k + 1u == per ? stop_src : base * ctx.fast_exp2(lg * (float)(k + 1u));
    s.r = r;
    s.t_end = t_end;
    s.t_stop = stop;
    s.owner = take ? owner : ctx.lane();  // lanes without a slice own nothing but themselves
    m.t_cur = begin;
    if (k == 0u) {  // the first slice continues exactly where the source lane was
        m.level = level;
        m.nx = nx;
        m.nz = nz;
        m.unverified_start = false;
    } else {
        // nodes the march works on grow with the distance walked: about one level per doubling of t
        const uint32_t top = T.mip_count - 1u, lvl = level + (uint32_t)(F3D_SHARE_LEVEL_GAIN * lg * (float)k);
        m.level = lvl < top ? lvl : top;
        march_locate(T, r, begin, m.level, m.nx, m.nz);
        m.unverified_start = true;
    }
    m.marching = take && begin <= t1;
}

// Phase 2 of an any-hit march (see above).  `m` holds the lane's position on its own ray, own_hit its
// verdict so far; returns the final verdict of the lane's OWN ray.
template <bool CURVED, bool FUSE = false, class Ctx>
F3D_HD bool march_shared(const TerrainDev &T, const RayCtx &own_ray, MarchState m, bool own_hit, Ctx &ctx, float t_stop = 3.0e38f) {
    MarchSlice s;
    s.r = own_ray;
    s.t_stop = t_stop;  // (a certificate that nothing lies beyond: the last slice ends there)
    s.owner = ctx.lane();
    {
        float lo;
        march_root_interval(T, own_ray, lo, s.t_end);
    }
    ctx.verdict_post(own_hit);
    uint32_t queued = 0u;
    TraceHit res;
    res.n = V3{0.0f, 0.0f, 0.0f};
    for (uint32_t round = 0u;; round++) {
        ctx.template deal<CURVED>(T, s, m);  // m.marching now says whether this lane got a slice
        if (m.marching) march_fetch<FUSE>(T, m, ctx);
        march_first_step<CURVED, true, FUSE>(T, s.r, m, queued, ctx, true, s.t_stop);  // slices other than a ray's first start in a located node
        res.hit = false;
        res.t = s.r.tmax;
        bool again = false;
        for (;;) {
            if (m.marching) march_step<CURVED, true, kVerifyInLoop, FUSE>(T, s.r, m, queued, ctx, true, s.t_stop);
#if !defined(F3D_STEPS_UNROLLED)
#pragma unroll 1
            for (uint32_t extra = 1u; extra < kStepsPerVoteShared && m.marching && queued + 2u <= kLeafFifoRows; extra++)
                march_step<CURVED, true, kVerifyInLoop, FUSE>(T, s.r, m, queued, ctx, true, s.t_stop);
#else
#pragma unroll
            for (uint32_t extra = 1u; extra < kStepsPerVoteShared; extra++)
                if (m.marching && queued + 2u <= kLeafFifoRows) march_step<CURVED, true, kVerifyInLoop, FUSE>(T, s.r, m, queued, ctx, true, s.t_stop);
#endif
            again = round + 1u < kShareRounds && ctx.share_now(m.marching);
            if (again || ctx.flush_now(queued, m.marching)) {
                march_drain<FUSE>(T, s.r, true, m, queued, res, ctx);
                if (res.hit) ctx.verdict_set(s.owner);
                if (ctx.verdict_get(s.owner)) m.marching = false;  // another slice of this ray has hit
            }
            if (again || !ctx.any(m.marching || queued != 0u)) break;
        }
        if (!again) break;
    }
    return ctx.verdict_get(ctx.lane());
}

// ---- the same for CLOSEST-hit rays (round 6; the PBR path tracer's camera and bounce rays, Ctx::kShareClosest) -----------------
// A closest-hit ray wants the FIRST leaf in ray order that hits (see the header).  Cut into slices, each slice's drain ends with
// the first hit of its own stretch; the ray's answer is the hit that comes first in ray order among them.  "First in ray order"
// needs no parameter: consecutive cells along a ray differ by one step in x or in z (or both, through a corner) in the ray's
// direction, so the PROGRESS p = (x_forward ? cx : 8191 - cx) + (z_forward ? cz : 8191 - cz) grows strictly from one leaf of the
// march to the next.  A slice that hits posts key = p << 13 | cx on its owner's word of the board (an LDS atomic max of the
// complement: the smallest key wins; p and cx name the cell, cz follows from them); a slice standing in a node whose smallest
// progress lies beyond the posted one stops.  When all slices are done the owner solves the winning leaf ITSELF -- the interval by
// march_step's expressions on its own ray (what the one-word FIFO's drain does), the same leaf_solve / leaf_normal -- so t and the
// normal are the unshared march's bit for bit (the cut's node is visited by both neighbours: the same leaf gives the same key).
constexpr uint32_t kNoNearest = 0xFFFFFFFFu;
F3D_HD uint32_t march_progress_key(const RayCtx &r, uint32_t cx, uint32_t cz) {
    const uint32_t px = !(r.d.x < 0.0f) ? cx : 8191u - cx, pz = !(r.d.z < 0.0f) ? cz : 8191u - cz;
    return ((px + pz) << 13) | cx;
}
template <bool CURVED, class Ctx>
F3D_HD TraceHit march_shared_closest(const TerrainDev &T, const RayCtx &own_ray, MarchState m, TraceHit own, Ctx &ctx, float t_stop = 3.0e38f) {
    MarchSlice s;
    s.r = own_ray;
    s.t_stop = t_stop;
    s.owner = ctx.lane();
    {
        float lo;
        march_root_interval(T, own_ray, lo, s.t_end);
    }
    ctx.nearest_post();  // (a lane that has its hit already is not marching: nobody walks a slice for it)
    uint32_t queued = 0u;
    TraceHit res;
    res.n = V3{0.0f, 0.0f, 0.0f};
    for (uint32_t round = 0u;; round++) {
        ctx.template deal<CURVED>(T, s, m);
        if (m.marching) march_fetch(T, m, ctx);
        march_first_step<CURVED, true>(T, s.r, m, queued, ctx, false, s.t_stop);
        res.hit = false;
        res.t = s.r.tmax;
        uint32_t cell = 0u;  // of this round's hit (a slice stops at its first: later drains of the round post the same key again)
        bool again = false;
        for (;;) {
            if (m.marching) march_step<CURVED, true, kVerifyInLoop>(T, s.r, m, queued, ctx, false, s.t_stop);
#pragma unroll 1
            for (uint32_t extra = 1u; extra < kStepsPerVoteShared && m.marching && queued + 2u <= kLeafFifoRows; extra++)
                march_step<CURVED, true, kVerifyInLoop>(T, s.r, m, queued, ctx, false, s.t_stop);
            again = round + 1u < kShareRounds && ctx.share_now(m.marching);
            if (again || ctx.flush_now(queued, m.marching)) {
                march_drain(T, s.r, false, m, queued, res, ctx, cell);
                if (res.hit) ctx.nearest_set(s.owner, march_progress_key(s.r, cell & 0xFFFFu, cell >> 16));
                const uint32_t best = ctx.nearest_get(s.owner);
                if (m.marching && best != kNoNearest) {  // is everything this slice can still find behind the posted hit?
                    const uint32_t x0 = m.nx << m.level, z0 = m.nz << m.level;
                    uint32_t x1 = (m.nx + 1u) << m.level, z1 = (m.nz + 1u) << m.level;
                    x1 = (x1 < T.cell_w ? x1 : T.cell_w) - 1u;
                    z1 = (z1 < T.cell_h ? z1 : T.cell_h) - 1u;
                    const uint32_t p_min = (!(s.r.d.x < 0.0f) ? x0 : 8191u - x1) + (!(s.r.d.z < 0.0f) ? z0 : 8191u - z1);
                    if ((best >> 13) < p_min) m.marching = false;
                }
            }
            if (again || !ctx.any(m.marching || queued != 0u)) break;
        }
        if (!again) break;
    }
    const uint32_t best = ctx.nearest_get(ctx.lane());
    if (best != kNoNearest) {
        const uint32_t cx = best & 8191u, p = best >> 13;
        const uint32_t pz = p - (!(own_ray.d.x < 0.0f) ? cx : 8191u - cx), cz = !(own_ray.d.z < 0.0f) ? pz : 8191u - pz;
        float lo, hi, t;
        march_leaf_interval(T, own_ray, cx, cz, lo, hi);
        const LeafRec leaf = T.leaves[tiled_index(cx, cz, T.tiles_x[0])];
        if (leaf_solve(T, own_ray, leaf, cx, cz, lo, hi, false, t) && t < own.t) {
            own.hit = true;
            own.t = t;
            own.n = leaf_normal(T, leaf, along(own_ray.o, t, own_ray.d), cx, cz);
        }
    }
    return own;
}

// A camera ray whose pixel holds a certificate (f3d_cone.h primary_start): every ray of the pixel is above every cell it
// passes up to t_clear, so the nodes before it are exactly those the march would reject without solving a leaf.  The
// lane starts in the node of `level` that contains the ray at t_clear -- located from the position and validated by
// that node's own interval on the first step, like the in-cell start of the secondary rays.
F3D_HD MarchState march_begin_at(const TerrainDev &T, const RayCtx &r, float t_clear, uint32_t level) {
    MarchState m;
    float hi;
    march_root_interval(T, r, m.t_cur, hi);
    m.marching = !(m.t_cur > hi);
    m.level = T.mip_count - 1u;
    m.nx = 0u;
    m.nz = 0u;
    m.unverified_start = false;
    if (m.marching && t_clear > m.t_cur) {
        if (!(t_clear < hi)) {
            m.marching = false;  // clear until the ray leaves the footprint (or reaches tmax): no terrain hit
        } else {
            m.t_cur = t_clear;
            m.level = level < m.level ? level : m.level;
            march_locate(T, r, t_clear, m.level, m.nx, m.nz);
            m.unverified_start = true;
        }
    }
    return m;
}

// CURVED: the sun-ray curvature policy is active for this ray (compile-time so that the other
// two thirds of the rays do not carry the parabola arithmetic).  start_in_cell: begin in the cell
// the ray is in (secondary rays) instead of at the root (camera rays entering from outside).
// Ctx provides: note(), band_entry(), the FIFO storage fifo_put/fifo_get, the wave votes
// flush_now(queued, marching) / any(pred), and share_now / deal / verdict_* (ray sharing).
// hit_cell (closest-hit callers that want it, no ray sharing): the cell (cx | cz << 16) of the hit.
template <bool CURVED, bool STOP = CURVED, bool FUSE = false, class Ctx>
F3D_HD TraceHit march_terrain_from(const TerrainDev &T, const RayCtx &r, bool any_hit, MarchState m, Ctx &ctx,
                                   float t_stop = 3.0e38f, uint32_t *hit_cell = nullptr) {
    TraceHit res;
    res.hit = false;
    res.t = r.tmax;
    res.n = V3{0.0f, 0.0f, 0.0f};
    ctx.note(2 | (any_hit ? 1 : 0) | (CURVED ? 4 : 0));  // statistics hook: a new ray starts
    ctx.feature(r.d.y);
    uint32_t queued = 0u, cell = 0u;
    bool deal = false;
    if (m.marching) march_fetch<FUSE>(T, m, ctx);
    march_first_step<CURVED, STOP, FUSE>(T, r, m, queued, ctx, any_hit, t_stop);
    for (;;) {
        // t_stop (occlusion rays, f3d_cone.h sun_clear_from / ibl_stop): no terrain beyond it, so the lane stops after the node that
        // contains it -- the SLICED rule; node and leaf intervals are NOT clipped by it, every visited node is judged
        // exactly as the unbounded march judges it
        if (m.marching) march_step<CURVED, STOP, kVerifyInLoop, FUSE>(T, r, m, queued, ctx, any_hit, t_stop);
        // further steps before the wave votes again (kStepsPerVote above)
#if !defined(F3D_STEPS_UNROLLED)  // a real loop (A/B: unrolled copies of the step -- bigger code, 1-2 % slower)
#pragma unroll 1
        for (uint32_t extra = 1u; extra < kStepsPerVote && m.marching && queued + 2u <= kLeafFifoRows; extra++)
            march_step<CURVED, STOP, kVerifyInLoop, FUSE>(T, r, m, queued, ctx, any_hit, t_stop);
#else
#pragma unroll
        for (uint32_t extra = 1u; extra < kStepsPerVote; extra++)
            if (m.marching && queued + 2u <= kLeafFifoRows) march_step<CURVED, STOP, kVerifyInLoop, FUSE>(T, r, m, queued, ctx, any_hit, t_stop);
#endif
#if !defined(F3D_NO_SHARE)
#if defined(F3D_SHARE_CURVED)  // A/B: sun rays too, with their own threshold (profiles/README.md)
        if (any_hit) deal = CURVED ? ctx.share_now(m.marching, F3D_SHARE_CURVED) : ctx.share_now(m.marching);
#else
        if (!CURVED && any_hit) deal = ctx.share_now(m.marching);  // the last few IBL rays: share them (needs empty FIFOs)
#endif
        if (Ctx::kShareClosest && !CURVED && !any_hit) deal = ctx.share_now(m.marching);  // ... and of closest-hit rays (march_shared_closest)
#endif
        // (Balancing the queued leaf solves of a wave over its lanes -- ceil(sum / lanes) rounds instead of
        // max(queued), the owner's ray fetched by ds_bpermute -- was built and measured: bit-identical, 0.96x.)
        if (deal || ctx.flush_now(queued, m.marching)) march_drain<FUSE>(T, r, any_hit, m, queued, res, ctx, cell);
        if (deal || !ctx.any(m.marching || queued != 0u)) break;
    }
    if (hit_cell) *hit_cell = cell;
    if constexpr (Ctx::kShareClosest) {
        if (deal && !any_hit) return march_shared_closest<CURVED>(T, r, m, res, ctx, t_stop);
    }
    if (deal) {
        res.hit = march_shared<CURVED, FUSE>(T, r, m, res.hit, ctx, t_stop);
        res.t = r.tmin;  // any-hit callers read only `hit` (and t < tmax)
    }
    return res;
}

template <bool CURVED, bool FUSE = false, class Ctx>
F3D_HD TraceHit march_terrain(const TerrainDev &T, const RayCtx &r, bool any_hit, bool start_in_cell, Ctx &ctx,
                              float t_stop = 3.0e38f) {
    return march_terrain_from<CURVED, true, FUSE>(T, r, any_hit, march_begin(T, r, start_in_cell), ctx, t_stop);
}

// ---- a STREAM of occlusion rays through the lanes of a wave (wavefront kernels, f3d_kernels.hip k_wf_occl) -----------
// The fused frame kernel marches the rays of a wave in lockstep: one sun ray (then one IBL ray) per lane, and the wave
// iterates until its longest ray is done -- measured on the headline frame, a lane-step of the two occlusion phases costs
// 1.8x what a lane-step of the primary phase costs, because most iterations run with a few lanes (tools/march_model.py:
// lockstep utilisation 0.32 / 0.21 against 0.70).  Here the rays come from a queue: a lane whose ray is done takes the
// next one, so the wave stays full until the queue is empty.  Per-ray arithmetic is march_step / march_drain unchanged --
// the verdict of a ray does not depend on which lane walks it or on what its neighbours do (3.1 of DESIGN.md).
// Lanes finish one by one; retiring and refilling them one by one would run the (divergent) drain and ray set-up code
// at 1 / 64 utilisation, so a STALLED lane (no ray, or a ray that has stopped marching but may have leaves waiting in its
// FIFO) waits until `quorum` lanes are stalled; then the wave drains all FIFOs, retires the finished rays and deals new
// ones to every idle lane, together.
// Source:  bool refill(bool &have, RayCtx &r, float &t_stop, uint32_t &tag, Ctx &ctx)  -- give a ray to every lane with
//          !have that can get one; returns false once the queue is exhausted (wave-uniform);
//          void verdict(uint32_t tag, bool occluded)
#ifndef F3D_STREAM_QUORUM
#define F3D_STREAM_QUORUM 16
#endif
template <bool CURVED, class Ctx, class Source>
F3D_HD void march_stream(const TerrainDev &T, Source &src, Ctx &ctx, uint32_t quorum = F3D_STREAM_QUORUM) {
    bool have = false;
    MarchState m;
    m.marching = false;
    m.unverified_start = false;
    m.t_cur = 0.0f;
    m.level = m.nx = m.nz = 0u;
    RayCtx r = make_ray(T, V3{0.0f, 0.0f, 0.0f}, 0.0f, V3{0.0f, 1.0f, 0.0f}, 0.0f, false);
    TraceHit res;
    res.hit = false;
    res.t = 0.0f;
    res.n = V3{0.0f, 0.0f, 0.0f};
    float t_stop = 3.0e38f;
    uint32_t tag = 0u, queued = 0u;
    bool more = true;  // wave-uniform: the source may still have rays
    const uint32_t lanes = bits_set(ctx.ballot(true));
    for (;;) {
        const uint32_t stalled = bits_set(ctx.ballot(!have || !m.marching));
        if (stalled >= quorum || stalled == lanes) {
            if (ctx.any(queued != 0u)) march_drain(T, r, true, m, queued, res, ctx);
            if (have && !m.marching) {  // (a drain may have stopped further lanes: they retire now too)
                src.verdict(tag, res.hit);
                have = false;
            }
            if (more) {
                const bool was = have;
                more = src.refill(have, r, t_stop, tag, ctx);
                if (have && !was) {
                    m = march_begin(T, r, true);  // occlusion rays start on the surface: in the origin's cell
                    if (m.marching) march_fetch(T, m, ctx);
                    res.hit = false;
                    res.t = r.tmax;
                    queued = 0u;
                }
            }
            if (!ctx.any(have)) {
                if (!more) break;
                continue;  // (a refill that handed out nothing although rays remain: ask again)
            }
        }
        if (have && m.marching) march_step<CURVED, true>(T, r, m, queued, ctx, true, t_stop);
        if (ctx.flush_now(queued, true)) march_drain(T, r, true, m, queued, res, ctx);  // a FIFO is full
    }
}

// Curvature is a per-ray policy AND a per-render switch (wave-uniform): pick the instantiation.
template <class Pending>
F3D_HD TraceHit march_ray(const TerrainDev &T, const RayCtx &r, bool any_hit, bool start_in_cell, Pending &pend) {
    if (r.c2 != 0.0f || r.has_vertex) return march_terrain<true>(T, r, any_hit, start_in_cell, pend);
    return march_terrain<false>(T, r, any_hit, start_in_cell, pend);
}

}  // namespace f3d
