// forge3d_amd/csrc/f3d_meshgrid.h -- the mesh as a second band of the terrain's min-max pyramid (round 6; host build).
//
// Reference: intersect_mesh (hybrid_traversal.wgsl:137-172) sweeps every triangle for every ray; its answer for an occlusion
// ray is "is any triangle accepted by ray_triangle_intersect in (tmin, tmax)".  The BVH walks of f3d_shade.h give that answer
// with a tree of their own -- a second traversal per ray, as long as the terrain march the ray has just finished (measured on
// the BASELINE configs[3] stand-in: 8.9 ms of a 32.6 ms frame are the occlusion rays' walks).  But a city stands ON its DEM: the
// terrain's quadtree is already a spatial index of it.  So the triangles are binned into the terrain's CELLS (by their x-z
// boxes, padded), every node of the pyramid gets a second (min, max) band -- the heights of the triangles binned under it --
// and the occlusion rays' march (f3d_march.h, FUSE) tests both bands of a node from one visit: it descends where either
// passes, and at a cell whose mesh band passes it puts the cell's triangles through the sweep's own ray_triangle.
//
// Conservative, not exact, by construction -- exactness is the sweep's arithmetic on the triangles that are tested: a triangle
// the ray meets is binned (with a margin for what rounding can move) in a cell the ray's march visits, under nodes whose mesh
// bands contain it.  A mesh with a triangle outside the DEM's footprint (the march never goes there), with non-finite vertices,
// or with triangles so large that the lists explode is left to the tree walk (ok = false).
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "f3d_setup.h"

namespace f3d {

struct MeshGrid {
    bool ok = false;
    std::vector<NodeRec> bands;        // the terrain band table's layout (TableLayout band_offset / band_shift), (+inf, -inf) where empty
    std::vector<uint32_t> cell_start;  // cell_w * cell_h + 1 prefix sums: the triangles of cell (cx, cz) are [start[cz * cell_w + cx], start[.. + 1])
    std::vector<float> tris;           // 3 float4 (v0, v1, v2; w = 0) per listed triangle, cell after cell
    float top = 0.0f;                  // the root's mesh band maximum
};

inline MeshGrid build_mesh_grid(const TableLayout &L, float origin_x, float origin_z, float spacing_x, float spacing_z, const float *vertices,
                                uint32_t vertex_count, const uint32_t *indices, uint32_t index_count) {
    MeshGrid g;
    const uint32_t tri_count = index_count / 3u, cw = L.cell_w, ch = L.cell_h;
    if (tri_count == 0u || cw == 0u || ch == 0u || !(spacing_x > 0.0f) || !(spacing_z > 0.0f)) return g;
    // margins: what float rounding can move a hit point, a plane parameter or a height (a few ulps of the largest coordinate),
    // generously
    double max_abs = std::fabs((double)origin_x) + (double)cw * spacing_x + std::fabs((double)origin_z) + (double)ch * spacing_z, max_y = 0.0;
    for (uint32_t v = 0; v < vertex_count; v++) {
        const float *p = vertices + 3u * (size_t)v;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) return g;
        max_y = std::fmax(max_y, std::fabs((double)p[1]));
    }
    const double pad_x = 0.01 * spacing_x + 1e-5 * max_abs, pad_z = 0.01 * spacing_z + 1e-5 * max_abs, pad_y = 1e-3 * (1.0 + max_y);
    const double x_lo = origin_x, x_hi = (double)origin_x + (double)cw * spacing_x, z_lo = origin_z, z_hi = (double)origin_z + (double)ch * spacing_z;
    struct Span {
        uint32_t cx0, cx1, cz0, cz1;
        float y0, y1;
    };
    std::vector<Span> spans(tri_count);
    std::vector<uint32_t> count((size_t)cw * ch + 1u, 0u);
    uint64_t entries = 0;
    const uint64_t cap = std::max<uint64_t>(16ull * tri_count, 1ull << 16);
    for (uint32_t t = 0; t < tri_count; t++) {
        double bx0 = 1e300, bx1 = -1e300, bz0 = 1e300, bz1 = -1e300, by0 = 1e300, by1 = -1e300;
        for (uint32_t k = 0; k < 3u; k++) {
            const uint32_t vi = indices[3u * (size_t)t + k];
            if (vi >= vertex_count) return g;
            const float *p = vertices + 3u * (size_t)vi;
            bx0 = std::fmin(bx0, p[0]), bx1 = std::fmax(bx1, p[0]);
            by0 = std::fmin(by0, p[1]), by1 = std::fmax(by1, p[1]);
            bz0 = std::fmin(bz0, p[2]), bz1 = std::fmax(bz1, p[2]);
        }
        bx0 -= pad_x, bx1 += pad_x, bz0 -= pad_z, bz1 += pad_z;
        if (bx0 < x_lo || bx1 > x_hi || bz0 < z_lo || bz1 > z_hi) return g;  // beyond the DEM: the march never goes there
        Span s;
        s.cx0 = (uint32_t)std::floor((bx0 - x_lo) / spacing_x), s.cx1 = (uint32_t)std::floor((bx1 - x_lo) / spacing_x);
        s.cz0 = (uint32_t)std::floor((bz0 - z_lo) / spacing_z), s.cz1 = (uint32_t)std::floor((bz1 - z_lo) / spacing_z);
        s.cx1 = s.cx1 < cw - 1u ? s.cx1 : cw - 1u;
        s.cz1 = s.cz1 < ch - 1u ? s.cz1 : ch - 1u;
        s.y0 = (float)(by0 - pad_y), s.y1 = (float)(by1 + pad_y);
        spans[t] = s;
        entries += (uint64_t)(s.cx1 - s.cx0 + 1u) * (s.cz1 - s.cz0 + 1u);
        if (entries > cap) return g;
        for (uint32_t z = s.cz0; z <= s.cz1; z++)
            for (uint32_t x = s.cx0; x <= s.cx1; x++) count[(size_t)z * cw + x]++;
    }
    g.cell_start.assign((size_t)cw * ch + 1u, 0u);
    {
        uint32_t run = 0u;
        for (size_t c = 0; c < (size_t)cw * ch; c++) {
            g.cell_start[c] = run;
            run += count[c];
        }
        g.cell_start[(size_t)cw * ch] = run;
    }
    g.tris.assign((size_t)entries * 12u, 0.0f);
    g.bands.assign(L.band_count, NodeRec{INFINITY, -INFINITY});
    std::vector<uint32_t> fill(g.cell_start.begin(), g.cell_start.end() - 1);
    for (uint32_t t = 0; t < tri_count; t++) {
        const Span &s = spans[t];
        for (uint32_t z = s.cz0; z <= s.cz1; z++)
            for (uint32_t x = s.cx0; x <= s.cx1; x++) {
                float *dst = g.tris.data() + 12u * (size_t)fill[(size_t)z * cw + x]++;
                for (uint32_t k = 0; k < 3u; k++) {
                    const float *p = vertices + 3u * (size_t)indices[3u * (size_t)t + k];
                    dst[4u * k] = p[0], dst[4u * k + 1u] = p[1], dst[4u * k + 2u] = p[2];
                }
                NodeRec &b = g.bands[L.band_offset[0] + ((size_t)z << L.band_shift[0]) + x];
                b.mn = std::fmin(b.mn, s.y0);
                b.mx = std::fmax(b.mx, s.y1);
            }
    }
    for (uint32_t l = 1; l < L.levels; l++) {  // a node's band: its (up to four) children's
        const uint32_t w = L.level_w[l], rows = L.band_rows[l], cw_prev = L.level_w[l - 1u], rows_prev = L.band_rows[l - 1u];
        for (uint32_t z = 0; z < rows; z++)
            for (uint32_t x = 0; x < w; x++) {
                NodeRec out{INFINITY, -INFINITY};
                for (uint32_t dz = 0; dz < 2u; dz++)
                    for (uint32_t dx = 0; dx < 2u; dx++) {
                        const uint32_t sx = 2u * x + dx, sz = 2u * z + dz;
                        if (sx >= cw_prev || sz >= rows_prev) continue;
                        const NodeRec &c = g.bands[L.band_offset[l - 1u] + ((size_t)sz << L.band_shift[l - 1u]) + sx];
                        out.mn = std::fmin(out.mn, c.mn);
                        out.mx = std::fmax(out.mx, c.mx);
                    }
                g.bands[L.band_offset[l] + ((size_t)z << L.band_shift[l]) + x] = out;
            }
    }
    g.top = g.bands[L.band_offset[L.levels - 1u]].mx;
    g.ok = true;
    return g;
}

}  // namespace f3d
