// forge3d_amd/csrc/f3d_build.h -- per-record bodies of the acceleration-table builders.
// Reference: build_minmax_mips (terrain_heightfield.rs:132-202) runs single-threaded on the
// CPU once per call; here every record of every level is one GPU thread (k_leaf_build /
// k_level_build in f3d_kernels.hip) writing straight into the tiled HBM layout.
#pragma once

#include "f3d_scene.h"

namespace f3d {

struct PyramidBuildParams {
    const float *heights;  // row-major DEM (w x h)
    uint32_t w, h;
    float exaggeration;
    LeafRec *leaves;
    uint32_t leaf_tiles_x, leaf_dim_x, leaf_dim_y;  // padded leaf-table dims (multiples of 8)
};

struct LevelBuildParams {
    // builds level `level` (>= 1) from level - 1 (the leaf table when level == 1)
    const LeafRec *leaves;
    const NodeRec *src;  // level - 1 records (level >= 2)
    NodeRec *dst;
    uint32_t level;
    uint32_t src_w, src_h;          // logical pow2 dims of level - 1 (reference dims)
    uint32_t dst_w, dst_h;          // logical dims of this level
    uint32_t src_tiles_x, dst_tiles_x;
    uint32_t dst_dim_x, dst_dim_y;  // padded dims of this level
    uint32_t cell_w, cell_h;
};

struct BandBuildParams {
    // row-major (min,max) copy of one level for the march: level 0 from the corner records,
    // level >= 1 from the tiled node table
    const LeafRec *leaves;
    const NodeRec *src;  // tiled records of this level (level >= 1)
    NodeRec *dst;        // bands + band_offset[level]
    uint32_t level;
    uint32_t width, height;  // logical pow2 dims of the level (= row pitch and row count)
    uint32_t shift;          // log2(width)
    uint32_t src_tiles_x;
    uint32_t cell_w, cell_h;
};

F3D_HD float min4(const LeafRec &h) { return f_min(f_min(f_min(h.h00, h.h10), h.h01), h.h11); }
F3D_HD float max4(const LeafRec &h) { return f_max(f_max(f_max(h.h00, h.h10), h.h01), h.h11); }

// Corner record of cell (x, y): terrain_cell_heights (hybrid_terrain_traversal.wgsl:149-156)
// evaluated once at build time; zeros outside the cell grid (never read by the traversal).
F3D_HD void leaf_build_at(const PyramidBuildParams &B, uint32_t x, uint32_t y) {
    LeafRec rec{0.0f, 0.0f, 0.0f, 0.0f};
    if (x + 1u < B.w && y + 1u < B.h) {
        const size_t i = (size_t)y * B.w + x;
        rec.h00 = B.heights[i] * B.exaggeration;
        rec.h10 = B.heights[i + 1] * B.exaggeration;
        rec.h01 = B.heights[i + B.w] * B.exaggeration;
        rec.h11 = B.heights[i + B.w + 1] * B.exaggeration;
    }
    B.leaves[tiled_index(x, y, B.leaf_tiles_x)] = rec;
}

// Node (x, y) of level >= 1: 2x2 reduce with clamp-to-edge once an axis has collapsed
// (terrain_heightfield.rs:174-190); (+inf, -inf) outside the logical level and for level-0
// padding cells (:154-155).
F3D_HD void level_build_at(const LevelBuildParams &B, uint32_t x, uint32_t y) {
    float mn = __builtin_inff(), mx = -__builtin_inff();
    if (x < B.dst_w && y < B.dst_h) {
        for (uint32_t dy = 0u; dy < 2u; dy++) {
            for (uint32_t dx = 0u; dx < 2u; dx++) {
                uint32_t sx = 2u * x + dx, sy = 2u * y + dy;
                sx = sx < B.src_w - 1u ? sx : B.src_w - 1u;
                sy = sy < B.src_h - 1u ? sy : B.src_h - 1u;
                if (B.level == 1u) {
                    if (sx < B.cell_w && sy < B.cell_h) {
                        const LeafRec h = B.leaves[tiled_index(sx, sy, B.src_tiles_x)];
                        mn = f_min(mn, min4(h));
                        mx = f_max(mx, max4(h));
                    }
                } else {
                    const NodeRec s = B.src[tiled_index(sx, sy, B.src_tiles_x)];
                    mn = f_min(mn, s.mn);
                    mx = f_max(mx, s.mx);
                }
            }
        }
    }
    B.dst[tiled_index(x, y, B.dst_tiles_x)] = NodeRec{mn, mx};
}

F3D_HD void band_build_at(const BandBuildParams &B, uint32_t x, uint32_t z) {
    NodeRec out{__builtin_inff(), -__builtin_inff()};
    if (B.level == 0u) {
        if (x < B.cell_w && z < B.cell_h) {
            const LeafRec h = B.leaves[tiled_index(x, z, B.src_tiles_x)];
            out = NodeRec{min4(h), max4(h)};
        }
    } else {
        out = B.src[tiled_index(x, z, B.src_tiles_x)];
    }
    B.dst[((size_t)z << B.shift) + x] = out;
}

}  // namespace f3d
