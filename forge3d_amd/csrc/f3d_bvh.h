// forge3d_amd/csrc/f3d_bvh.h -- mesh acceleration structure (host build, HIP-free).
//
// Reference: the driver builds a SAH BVH on the CPU for the optional mesh (`accel::build_bvh`,
// src/accel/sah_cpu.rs:23-96, called from render_terrain.rs:597-627) and uploads it, but its
// kernel never reads it: `intersect_mesh` sweeps every triangle (hybrid_traversal.wgsl:137-172).
// The RESULT of that sweep is "the triangle with the smallest Moller-Trumbore t, the lowest index
// among equal t" (closest hit) or "is there any triangle hit" (occlusion rays).  Here the BVH is
// actually used: same per-triangle arithmetic, same answer, O(log T) instead of O(T) per ray.
//
// Layout for the GPU: a THREADED binary BVH in depth-first preorder -- a ray that enters node i
// continues at i + 1 (its first child, or the node after a leaf's triangles are tested), a ray
// that misses it continues at skip[i] (the node after i's subtree).  No stack, no per-lane LDS, one
// 32-byte record (two dwordx4 loads) per visited node; triangles are copied into leaf order as three
// float4 (xyz + the ORIGINAL triangle index in v0.w) so a leaf is a contiguous 48-byte-stride run.
// Visiting order is fixed (not front-to-back); closest-hit rays prune with the best t so far.
//
// Node boxes are padded (kPadRel of the scene diagonal): Moller-Trumbore accepts points a few ulps
// outside the exact triangle, and the slab test rounds too; the padding dominates both, so a triangle
// the sweep would accept is never culled (tests compare with the brute-force oracle bit for bit).
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "f3d_scene.h"

namespace f3d {

constexpr uint32_t kBvhLeafMax = 4u;    // triangles per leaf (the reference's builder: leaf <= 4)
constexpr uint32_t kBvhBins = 16u;      // SAH bins per axis
constexpr float kBvhPadRel = 1e-5f;     // box padding as a fraction of the scene diagonal
constexpr uint32_t kBvhParallelMin = 20000u;  // triangles from which the build uses worker threads
constexpr uint32_t kBvhTopDepth = 6u;         // levels split on the calling thread (<= 64 subtree tasks)

struct MeshBvh {
    std::vector<BvhNode> nodes;
    std::vector<float> tris;  // 12 floats per triangle: v0.xyz, index bits, v1.xyz, 0, v2.xyz, 0
    size_t bytes() const { return nodes.size() * sizeof(BvhNode) + tris.size() * sizeof(float); }
};

namespace bvh_detail {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) {
            lo[a] = INFINITY;
            hi[a] = -INFINITY;
        }
    }
    void grow(const float *p) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box &b) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    Box box;
    float centroid[3];
    uint32_t tri;
};

struct Builder {
    const float *verts;
    const uint32_t *idx;
    std::vector<Prim> prims;
    MeshBvh out;
    float pad;

    // Binned SAH split of prims[first, first + count), count > kBvhLeafMax (cost model of
    // sah_cpu.rs:235-306: SA-weighted primitive counts); returns the size of the left part.
    uint32_t split(uint32_t first, uint32_t count) {
        Box cb;
        cb.reset();
        for (uint32_t i = first; i < first + count; i++) cb.grow(prims[i].centroid);
        float best_cost = INFINITY;
        int best_axis = -1;
        uint32_t best_bin = 0;
        for (int a = 0; a < 3; a++) {
            const float extent = cb.hi[a] - cb.lo[a];
            if (!(extent > 0.0f)) continue;
            Box bin_box[kBvhBins];
            uint32_t bin_n[kBvhBins] = {};
            for (auto &b : bin_box) b.reset();
            const float scale = (float)kBvhBins / extent;
            for (uint32_t i = first; i < first + count; i++) {
                uint32_t b = (uint32_t)((prims[i].centroid[a] - cb.lo[a]) * scale);
                b = b < kBvhBins ? b : kBvhBins - 1u;
                bin_n[b]++;
                bin_box[b].grow(prims[i].box);
            }
            float right_area[kBvhBins];
            uint32_t right_n[kBvhBins];
            Box acc;
            acc.reset();
            uint32_t n = 0;
            for (int b = (int)kBvhBins - 1; b > 0; b--) {
                acc.grow(bin_box[b]);
                n += bin_n[b];
                right_area[b] = acc.half_area();
                right_n[b] = n;
            }
            acc.reset();
            n = 0;
            for (uint32_t b = 0; b + 1 < kBvhBins; b++) {
                acc.grow(bin_box[b]);
                n += bin_n[b];
                if (n == 0 || right_n[b + 1] == 0) continue;
                const float cost = acc.half_area() * (float)n + right_area[b + 1] * (float)right_n[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = a;
                    best_bin = b;
                }
            }
        }
        if (best_axis < 0) return count / 2u;  // coincident centroids: split by order
        const float extent = cb.hi[best_axis] - cb.lo[best_axis];
        const float scale = (float)kBvhBins / extent;
        auto mid = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim &p) {
            uint32_t b = (uint32_t)((p.centroid[best_axis] - cb.lo[best_axis]) * scale);
            b = b < kBvhBins ? b : kBvhBins - 1u;
            return b <= best_bin;
        });
        const uint32_t left = (uint32_t)(mid - (prims.begin() + first));
        if (left == 0u || left == count) return count / 2u;
        return left;
    }

    // Build the subtree over prims[first, first + count) into `out` (indices local to that arena).
    void emit(MeshBvh &out, uint32_t first, uint32_t count, uint32_t depth) {
        Box bounds;
        bounds.reset();
        for (uint32_t i = first; i < first + count; i++) bounds.grow(prims[i].box);
        const uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.push_back(BvhNode{});
        // leaves of <= kBvhLeafMax triangles; a runaway one-sided SAH recursion falls back to halving
        const uint32_t left = count <= kBvhLeafMax ? 0u : (depth < 48u ? split(first, count) : count / 2u);
        uint32_t leaf = 0u;
        if (left == 0u) {
            // a leaf: copy its triangles (ascending original index, like the sweep's order) into leaf order
            std::sort(prims.begin() + first, prims.begin() + first + count,
                      [](const Prim &a, const Prim &b) { return a.tri < b.tri; });
            const uint32_t tri_first = (uint32_t)(out.tris.size() / 12u);
            for (uint32_t i = first; i < first + count; i++) {
                const uint32_t t = prims[i].tri;
                for (int v = 0; v < 3; v++) {
                    const float *p = verts + 3u * (size_t)idx[3u * (size_t)t + v];
                    out.tris.push_back(p[0]);
                    out.tris.push_back(p[1]);
                    out.tris.push_back(p[2]);
                    float w = 0.0f;
                    if (v == 0) std::memcpy(&w, &t, sizeof(float));
                    out.tris.push_back(w);
                }
            }
            leaf = (tri_first << 3) | count;
        } else {
            emit(out, first, left, depth + 1u);
            emit(out, first + left, count - left, depth + 1u);
        }
        BvhNode &n = out.nodes[me];
        for (int a = 0; a < 3; a++) {
            n.bmin[a] = bounds.lo[a] - pad;
            n.bmax[a] = bounds.hi[a] + pad;
        }
        n.skip = (uint32_t)out.nodes.size();
        n.leaf = leaf;
    }
};

}  // namespace bvh_detail

// verts: xyz per vertex; idx: 3 per triangle, every index < vertex_count (validate_scene).
inline MeshBvh build_mesh_bvh(const float *verts, uint32_t vertex_count, const uint32_t *idx, uint32_t index_count,
                              uint32_t parallel_min = kBvhParallelMin) {
    using namespace bvh_detail;
    Builder b;
    b.verts = verts;
    b.idx = idx;
    const uint32_t ntri = index_count / 3u;
    b.prims.reserve(ntri);
    Box scene;
    scene.reset();
    for (uint32_t t = 0; t < ntri; t++) {
        Prim p;
        p.box.reset();
        p.tri = t;
        bool ok = true;
        for (int v = 0; v < 3; v++) ok = ok && idx[3u * (size_t)t + v] < vertex_count;
        if (!ok) continue;  // the sweep skips such triangles (hybrid_traversal.wgsl:150-153)
        for (int v = 0; v < 3; v++) p.box.grow(verts + 3u * (size_t)idx[3u * (size_t)t + v]);
        for (int a = 0; a < 3; a++) p.centroid[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
        scene.grow(p.box);
        b.prims.push_back(p);
    }
    if (b.prims.empty()) return b.out;
    const float dx = scene.hi[0] - scene.lo[0], dy = scene.hi[1] - scene.lo[1], dz = scene.hi[2] - scene.lo[2];
    float mag = 0.0f;
    for (int a = 0; a < 3; a++) mag = std::max(mag, std::max(std::fabs(scene.lo[a]), std::fabs(scene.hi[a])));
    b.pad = kBvhPadRel * std::sqrt(dx * dx + dy * dy + dz * dz) + 4e-6f * mag + 1e-30f;
    // Large meshes: the top kBvhTopDepth levels are split on this thread (the splits partition disjoint
    // ranges of `prims`), the subtrees below are built by a few worker threads into their own arenas
    // and spliced into preorder afterwards (`skip` links and triangle offsets shifted).  The result does
    // not depend on the number of threads.
    const uint32_t total = (uint32_t)b.prims.size();
#ifdef F3D_BVH_TIMING
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    auto lap = [&](const char *what) { double t = now(); fprintf(stderr, "  bvh %-10s %.3f s\n", what, t - t_mark); t_mark = t; };
    lap("prims");
#else
    auto lap = [](const char *) {};
#endif
    if (total < parallel_min) {
        b.out.nodes.reserve(2u * (size_t)total);
        b.out.tris.reserve(12u * (size_t)total);
        b.emit(b.out, 0u, total, 0u);
        return b.out;
    }
    struct Top {
        uint32_t first, count, left;  // left == 0: a subtree task (index in `task`)
        uint32_t task;
    };
    std::vector<Top> tops;       // preorder of the top part
    std::vector<MeshBvh> arenas;  // one per subtree task
    {
        struct Item {
            uint32_t first, count, depth;
        };
        std::vector<Item> stack{{0u, total, 0u}};
        while (!stack.empty()) {  // depth-first, left child first == preorder
            const Item it = stack.back();
            stack.pop_back();
            if (it.depth >= kBvhTopDepth || it.count <= 4u * kBvhLeafMax) {
                tops.push_back(Top{it.first, it.count, 0u, (uint32_t)arenas.size()});
                arenas.emplace_back();
                continue;
            }
            const uint32_t left = b.split(it.first, it.count);
            tops.push_back(Top{it.first, it.count, left, 0u});
            stack.push_back(Item{it.first + left, it.count - left, it.depth + 1u});
            stack.push_back(Item{it.first, left, it.depth + 1u});
        }
    }
    lap("top");
    {
        std::vector<uint32_t> jobs;
        for (uint32_t i = 0; i < (uint32_t)tops.size(); i++)
            if (tops[i].left == 0u) jobs.push_back(i);
        std::atomic<uint32_t> next{0u};
        auto work = [&]() {
            for (;;) {
                const uint32_t j = next.fetch_add(1u);
                if (j >= (uint32_t)jobs.size()) break;
                const Top &t = tops[jobs[j]];
                MeshBvh &arena = arenas[t.task];
                arena.nodes.reserve(2u * (size_t)t.count);
                arena.tris.reserve(12u * (size_t)t.count);
                b.emit(arena, t.first, t.count, kBvhTopDepth);
            }
        };
        const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
        const uint32_t nthreads = std::min<uint32_t>(std::min<uint32_t>(hw, 32u), (uint32_t)jobs.size());
        std::vector<std::thread> pool;
        for (uint32_t i = 1; i < nthreads; i++) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
    }
    lap("subtrees");
    // splice: walk the top part in preorder again
    b.out.nodes.reserve(2u * (size_t)total);
    b.out.tris.reserve(12u * (size_t)total);
    struct Frame {
        uint32_t top, node, pending;  // index in tops, node index in out, children still to finish
    };
    std::vector<Frame> open;
    auto close_finished = [&]() {
        while (!open.empty() && open.back().pending == 0u) {
            BvhNode &n = b.out.nodes[open.back().node];
            n.skip = (uint32_t)b.out.nodes.size();
            // box of an inner top node = union of its children's (already padded) boxes
            const BvhNode &l = b.out.nodes[open.back().node + 1u];
            const BvhNode &r = b.out.nodes[l.skip];
            for (int a = 0; a < 3; a++) {
                n.bmin[a] = std::min(l.bmin[a], r.bmin[a]);
                n.bmax[a] = std::max(l.bmax[a], r.bmax[a]);
            }
            n.leaf = 0u;
            open.pop_back();
            if (!open.empty()) open.back().pending--;
        }
    };
    for (uint32_t i = 0; i < (uint32_t)tops.size(); i++) {
        const Top &t = tops[i];
        if (t.left != 0u) {
            open.push_back(Frame{i, (uint32_t)b.out.nodes.size(), 2u});
            b.out.nodes.push_back(BvhNode{});
            continue;
        }
        const MeshBvh &arena = arenas[t.task];
        const uint32_t node0 = (uint32_t)b.out.nodes.size(), tri0 = (uint32_t)(b.out.tris.size() / 12u);
        for (BvhNode n : arena.nodes) {
            n.skip += node0;
            if (n.leaf != 0u) n.leaf = (((n.leaf >> 3) + tri0) << 3) | (n.leaf & 7u);
            b.out.nodes.push_back(n);
        }
        b.out.tris.insert(b.out.tris.end(), arena.tris.begin(), arena.tris.end());
        if (!open.empty()) open.back().pending--;
        close_finished();
    }
    lap("splice");
    return b.out;
}

// ---- the same tree, four children wide ------------------------------------------------------------------------------
// The threaded binary walk visits a node per box test and every visit is a dependent, scattered 32-byte load: measured on
// the 600 000-triangle stand-in of BASELINE.json configs[3], 45 loop iterations per walk at ~1 300 cycles each, ~150 of
// them the wave's own instructions.  Collapsing the binary tree -- a node adopts its grandchildren,
// largest box first, until it has four children -- puts four box tests behind ONE 128-byte load and cuts the chain of
// dependent loads to the number of ENTERED nodes.  Leaves keep their triangles (the binary tree's leaf order).
// Returns an empty vector when the tree is deeper than the walk's per-level stack (kBvh4MaxLevels).
inline std::vector<Bvh4Node> collapse_bvh4(const MeshBvh &bvh) {
    std::vector<Bvh4Node> out;
    if (bvh.nodes.empty()) return out;
    const auto area = [&](uint32_t n) {
        const BvhNode &b = bvh.nodes[n];
        const float dx = b.bmax[0] - b.bmin[0], dy = b.bmax[1] - b.bmin[1], dz = b.bmax[2] - b.bmin[2];
        return dx * dy + dy * dz + dz * dx;
    };
    struct Item {
        uint32_t binary, wide, level;  // binary inner node -> record `wide` at `level`
    };
    std::vector<Item> todo;
    out.emplace_back();
    bool too_deep = false;
    const auto fill = [&](Bvh4Node &n, uint32_t slot, uint32_t b) {
        const BvhNode &s = bvh.nodes[b];
        n.lo_x[slot] = s.bmin[0];
        n.hi_x[slot] = s.bmax[0];
        n.lo_y[slot] = s.bmin[1];
        n.hi_y[slot] = s.bmax[1];
        n.lo_z[slot] = s.bmin[2];
        n.hi_z[slot] = s.bmax[2];
        n.leaf[slot] = s.leaf;
    };
    const auto clear = [&](Bvh4Node &n) {
        for (uint32_t s = 0; s < 4u; s++) {
            // an EMPTY slot: both planes at +inf on every axis.  The walk's slab test orders each axis's two plane
            // parameters with min / max, so the "inverted" box (+inf, -inf) of round 4 passed for every ray (enter = tmin,
            // exit = t_best) and only its leaf word of 0 triangles kept the answer right (round-4 advice).  With both planes
            // at +inf an axis gives (+inf, +inf) for a positive direction component -- enter = +inf -- and (-inf, -inf) for
            // a negative one -- exit = -inf; a ray's t_best is finite (1e30 at most), so enter <= exit fails either way.
            n.lo_x[s] = n.lo_y[s] = n.lo_z[s] = INFINITY;
            n.hi_x[s] = n.hi_y[s] = n.hi_z[s] = INFINITY;
            n.leaf[s] = 0u;
        }
        n.first_child = n.inner = n.pad0 = n.pad1 = 0u;
    };
    clear(out[0]);
    if (bvh.nodes[0].leaf != 0u) {  // a mesh of one leaf: a root whose only child is that leaf
        fill(out[0], 0u, 0u);
        return out;
    }
    todo.push_back(Item{0u, 0u, 0u});
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        // the children of binary node i: i + 1 and the node after i + 1's subtree
        uint32_t kids[4] = {it.binary + 1u, bvh.nodes[it.binary + 1u].skip, 0u, 0u};
        uint32_t n = 2u;
        while (n < 4u) {  // adopt the grandchildren of the inner child with the largest box
            int pick = -1;
            float best = -1.0f;
            for (uint32_t k = 0; k < n; k++)
                if (bvh.nodes[kids[k]].leaf == 0u && area(kids[k]) > best) {
                    best = area(kids[k]);
                    pick = (int)k;
                }
            if (pick < 0) break;
            const uint32_t b = kids[pick];
            kids[pick] = b + 1u;
            kids[n++] = bvh.nodes[b + 1u].skip;
        }
        // inner children first: they become consecutive records
        uint32_t order[4], inner = 0u, m = 0u;
        for (uint32_t k = 0; k < n; k++)
            if (bvh.nodes[kids[k]].leaf == 0u) order[m++] = kids[k];
        inner = m;
        for (uint32_t k = 0; k < n; k++)
            if (bvh.nodes[kids[k]].leaf != 0u) order[m++] = kids[k];
        const uint32_t first_child = (uint32_t)out.size();
        if (inner != 0u && it.level > kBvh4MaxLevels - 1u) too_deep = true;  // (no row left to record the siblings it descends past)
        for (uint32_t k = 0; k < inner; k++) {
            out.emplace_back();
            clear(out.back());
        }
        Bvh4Node &rec = out[it.wide];
        for (uint32_t k = 0; k < n; k++) fill(rec, k, order[k]);
        for (uint32_t k = 0; k < inner; k++) rec.leaf[k] = 0u;
        rec.first_child = first_child;
        rec.inner = inner;
        for (uint32_t k = 0; k < inner; k++) todo.push_back(Item{order[k], first_child + k, it.level + 1u});
    }
    if (too_deep || out.size() >= (1u << 23)) out.clear();  // (the stack word holds first_child in 24 bits: f3d_shade.h mesh_bvh4)
    return out;
}

}  // namespace f3d
