// forge3d_amd/csrc/f3d_scene.h
// Kernel-side scene description and the HBM layout of the terrain acceleration data.
//
// Reference data (terrain_heightfield.rs:132-202, :204-336): an R32F DEM texture plus an
// RG32F min-max mip chain, fetched one texel per visited node and 4 + 4 texels per leaf.
// MI355X layout (no texture units; everything is plain 16-byte-vector loads):
//
//  * leaf table  : one 16-byte record per DEM cell = its four corner heights already
//                  multiplied by `exaggeration` (h00, h10, h01, h11).  Level 0 of the
//                  min-max chain is NOT stored: min/max of a cell are the min/max of
//                  those four values, exactly what build_minmax_mips stores for level 0.
//  * node table  : levels >= 1 of the chain, (min, max) * exaggeration, 8 bytes a node.
//  * both tables are tiled 8x8 with a Z-order inside the tile, so the 2x2 children of a
//    node are 4 consecutive records (64 B of leaves = one cache line, 32 B of nodes) and
//    the 4x4 grandchildren share a 256 B / 128 B run.  A visited node therefore costs ONE
//    dependent round trip that returns all four children, instead of one per child.
//  * padded records (outside the cell grid, or added to round a level up to 8x8) hold
//    (+inf, -inf) / zeros; the traversal never uses them because the cell-range test
//    (reference hybrid_terrain_traversal.wgsl:284, :332) comes first.
#pragma once

#include "f3d_math.h"

namespace f3d {

constexpr uint32_t kMaxLevels = 16;       // 8192-cell DEMs need 14
constexpr uint32_t kRestirMCap = 512;     // TERRAIN_RESTIR_M_CAP, hybrid_terrain_traversal.wgsl:77
constexpr uint32_t kWelfordWindow = 32;   // WELFORD_WINDOW, render_terrain.rs:236
// Rows of neighbour state a strip keeps above and below its own.  The spatial pass offsets a neighbour by
// floor(u * 7) - 3 (pt_restir_spatial.wgsl:171-176) and u = f32(x) / 2^32 IS 1.0 for the top 128 values of x, so the
// reach is [-3, +4], not the nominal radius 3: 2^-25 of the draws look four rows down.
constexpr uint32_t kHaloRows = 4;
#if !defined(F3D_SPATIAL_REACH_HI)  // (3 = round 4's window: test-of-the-tests build for tests/test_gpu_head_reach.py)
#define F3D_SPATIAL_REACH_HI 4
#endif
constexpr uint32_t kSpatialReachLo = 3, kSpatialReachHi = F3D_SPATIAL_REACH_HI;  // the same reach per axis, for code that bounds the pass's reads
static_assert(kHaloRows >= kSpatialReachHi && kHaloRows >= kSpatialReachLo, "a strip's halo covers the spatial pass");
constexpr uint32_t kIblSectors = 8;      // azimuth sectors of the IBL rays' far-horizon certificate (f3d_cone.h)
constexpr uint32_t kDefaultLeafQuorum = 64;  // lanes with a queued leaf that trigger a wave drain
                                             // (64 = only when a FIFO is full or nobody marches; measured best)

struct alignas(16) LeafRec {
    float h00, h10, h01, h11;
};
struct alignas(8) NodeRec {
    float mn, mx;
};

// Index of the first of the four child records of parent (px, py); tiles_x = number of
// 8x8 tiles per row of the CHILD level.  Children follow in (cx, cy) = (0,0),(1,0),(0,1),(1,1)
// order = the reference's scan order (cy outer, cx inner).
F3D_HD uint32_t child_group_index(uint32_t px, uint32_t py, uint32_t tiles_x) {
    uint32_t tile = mul24((py >> 2), tiles_x) + (px >> 2);  // (both factors < 2^13: the full-rate 24-bit multiply)
    uint32_t g = (px & 1u) | ((py & 1u) << 1) | ((px & 2u) << 1) | ((py & 2u) << 2);
    return (tile << 6) | (g << 2);
}
// Index of record (x, y) in a tiled level (used by the builders).
F3D_HD uint32_t tiled_index(uint32_t x, uint32_t y, uint32_t tiles_x) {
    return child_group_index(x >> 1, y >> 1, tiles_x) | (x & 1u) | ((y & 1u) << 1);
}

// Everything the traversal needs; passed to kernels by value (kernarg -> SGPRs).
struct TerrainDev {
    const LeafRec *leaves;  // tiled, dims padded to multiples of 8
    const NodeRec *nodes;   // levels 1.. back to back, each tiled
    uint32_t node_offset[kMaxLevels];  // record offset of level l (l >= 1) inside `nodes`
    uint32_t tiles_x[kMaxLevels];      // tiles per row of level l (l = 0 -> leaf table)
    // (min,max)*exaggeration of EVERY level (0 = per cell), row-major with a power-of-two pitch:
    // record (x, z) of level l is bands[band_offset[l] + (z << band_shift[l]) + x].  This is the
    // table the stackless march reads (one 8-byte record per step, 2-instruction address).
    const NodeRec *bands;
    uint32_t band_offset[kMaxLevels];
    uint32_t band_shift[kMaxLevels];
    uint32_t mip_count;                // levels of the reference chain (incl. level 0)
    uint32_t cell_w, cell_h;
    float origin_x, origin_z, spacing_x, spacing_z, inv_spacing_x, inv_spacing_z;
    float inv_two_r_prime;  // EarthCurvatureUniforms, terrain_heightfield.rs:42-84
    uint32_t curvature_enabled;
    // far-horizon table of the IBL rays (f3d_cone.h): per block of 2^horizon_level cells 8 sector slopes; null = off
    const float *horizon;
    uint32_t horizon_level, horizon_bx;
    uint32_t leaf_quorum;   // lanes of a wave that must hold a fat leaf before the leaf body runs
    uint32_t share_below;   // ray sharing (f3d_march.h): deal when at most this many lanes still march; 0 = default
    // The scene's mesh as a second band of the same pyramid (f3d_meshgrid.h; the occlusion rays' march of the kernels compiled
    // for scenes with a mesh, f3d_march.h FUSE): mesh_bands has the layout of `bands`; the triangles binned in cell (cx, cz) are
    // mesh_cell_tris[3 e .. 3 e + 2] for e in [mesh_cell_start[cz * cell_w + cx], mesh_cell_start[.. + 1]).  Without a grid
    // mesh_cell_start is null and mesh_bands = bands (the march then tests the terrain's band twice and the tree is walked).
    const NodeRec *mesh_bands;
    const uint32_t *mesh_cell_start;
    const float4 *mesh_cell_tris;
    float mesh_top;  // the root's mesh band maximum
    uint32_t mesh_reserved;  // (explicit: the record has no padding, its bytes are hashed by f3d_session_fingerprint)
};

// Threaded BVH node (f3d_bvh.h): preorder layout, enter -> node + 1, miss / subtree done -> skip.
struct BvhNode {
    float bmin[3];
    uint32_t skip;
    float bmax[3];
    uint32_t leaf;  // 0 = inner node; else (first triangle << 3) | triangle count (1..4)
};
static_assert(sizeof(BvhNode) == 32, "BvhNode is two dwordx4 loads");

// 4-wide BVH node (f3d_bvh.h collapse_bvh4, f3d_shade.h mesh_bvh4): the CHILDREN's boxes live in the parent, one 128-byte
// record = one cache line per visited node.  Slots [0, inner) are inner children -- the records first_child + slot --,
// slots [inner, 4) leaves ((first triangle << 3) | count in `leaf[slot]`) or empty (box (+inf, -inf), leaf 0).
struct alignas(128) Bvh4Node {
    float lo_x[4], hi_x[4], lo_y[4], hi_y[4], lo_z[4], hi_z[4];
    uint32_t leaf[4];
    uint32_t first_child, inner, pad0, pad1;
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node is eight dwordx4 loads, one cache line");
// Levels of the 4-wide tree whose nodes may have unvisited siblings waiting: the walk keeps one word per level in the
// lane's LDS column, in the rows of the terrain march's leaf FIFO (empty while a mesh is walked).  Deeper trees are walked
// in the threaded binary form.
constexpr uint32_t kBvh4MaxLevels = 15;

struct MeshDev {  // HybridUniforms mesh part, hybrid_traversal.wgsl:9-17
    const float4 *vertices;  // xyz + pad (reference MeshVertex)
    const uint32_t *indices;
    uint32_t vertex_count, index_count;
    uint32_t traversal_mode;  // 0 hybrid (mesh + terrain), 3 terrain only
    // acceleration structure over the triangles (null -> the reference's sweep over all of them)
    const BvhNode *bvh_nodes;
    const float4 *bvh_tris;  // 3 float4 per triangle in leaf order; v0.w = original triangle index (bits)
    uint32_t bvh_node_count;
    const Bvh4Node *bvh4_nodes;  // the same tree four children wide (null: walk the binary form); shares bvh_tris
    uint32_t bvh4_node_count;
};

struct EnvDev {  // equirect environment, hybrid_terrain_traversal.wgsl:392-405
    const float4 *texels;  // rgb + pad, row-major
    uint32_t width, height;  // 0 -> constant white
    float intensity;
};

struct CameraDev {  // Uniforms, hybrid_kernel.wgsl:8-23 (+ derived constants)
    V3 origin, right, up, forward;
    float half_w, half_h;  // aspect * tan(fov/2), tan(fov/2)
    float exposure;
    uint32_t width, height;  // FULL image size
    uint32_t seed_hi, seed_lo;
    float cone_delta;  // f3d_cone.h pixel_cone_delta(): half-width of a pixel's ray cone per unit depth (< 0: no certificates)
};

// Half-width of a pixel's ray cone per unit depth, with margin (the certificates of f3d_cone.h are built for it);
// < 0: pixels too wide for certificates.
F3D_HD float pixel_cone_delta_of(float half_w, float half_h, uint32_t width, uint32_t height) {
    const float px = half_w / (float)width, py = half_h / (float)height;
    const float plane = f_sqrt(px * px + py * py);
    return plane < 0.25f ? 1.01f * plane * (1.0f + plane * plane) : -1.0f;
}

struct LightDev {  // LightingUniforms, hybrid_kernel.wgsl:27-38 (+ derived constants)
    V3 wi;         // normalize(light_dir): candidate sample direction
    V3 wi_reuse;   // normalize(wi): what the kernels get from a stored reservoir direction
    V3 color;      // light_color = intensity * colour
    V3 albedo;     // terrain albedo
    uint32_t shadows_enabled;
};

// Packed reservoir (16 B) replacing the reference's 80-byte Reservoir
// (src/path_tracing/restir/types.rs:6-37).  The sun is the only light, so
// sample.direction / intensity / light_index are render constants and
// sample.position never influences a result; what the three reference passes read is
// w_sum, m, weight, target_pdf and whether sample.light_type == 1, kept in bit 31 of m
// (m <= 9 * (512 + 64) by construction).
struct alignas(16) PackedReservoir {
    float w_sum;
    uint32_t m_lt;  // bit 31: sample.light_type == 1; bits 0..30: m
    float weight;
    float target_pdf;
};
constexpr uint32_t kLightTypeBit = 0x80000000u;

// Ray queues of the wavefront form of a trace batch (f3d_kernels.hip k_wf_primary -> k_wf_occl): k_wf_primary leaves
// every sample's record with both occlusion verdicts assumed "visible" and files the occlusion rays that have to be
// decided; k_wf_occl streams them through persistent waves and zeroes the term of a record whose ray is blocked.
// The queues are REGIONS of 64 slots, one per (frame of the batch, tile, round of the tile's wave): the wave compacts
// its rays into its own region (ballot + prefix count, no atomic -- a counter shared by the 130 000 waves of a frame was
// measured at 3 ns per atomic, 2.5 ms a frame) and leaves the three counts in one word.  Regions are numbered by TILE,
// so consecutive regions hold rays that start next to one another.  `tag` of a ray = index of its sample's record pair
// in FrameParams::trace.
struct WfQueues {
    float4 *sun_o;     // [regions][64] {origin, tag bits}: rays along light.wi fill a region from its FRONT, rays along
    float *sun_stop;   // [regions][64] light.wi_reuse from its BACK (the two differ in the last bit for most suns); t_stop
    float4 *ibl_o;     // [regions][64] {origin, tag bits}
    float4 *ibl_d;     // [regions][64] {direction, t_stop}
    uint32_t *counts;  // [regions] front sun rays | back sun rays << 8 | IBL rays << 16
    uint32_t *cursors; // [3] chunk cursors of the three consumer launches (cleared per batch)
    uint32_t regions_per_frame;  // tiles * rounds
};
constexpr uint32_t kWfRegion = 64u;

// Per-frame kernel parameters.
struct FrameParams {
    TerrainDev terrain;
    MeshDev mesh;
    EnvDev env;
    CameraDev cam;
    LightDev light;
    uint32_t spp;
    uint32_t frame_index;
    uint32_t row_begin, row_end;  // owned image rows
    // state (strip-local): pixel (gx, gy) lives at (gy - row_begin) * width + gx, the
    // reservoir buffers have kHaloRows extra rows above and below.
    const PackedReservoir *res_in;  // temporal output of frame-1 (incl. halos)
    PackedReservoir *res_out;       // temporal output of this frame
    float4 *accum_mean;             // rgb = sum of per-frame means, w = Welford mean
    float *welford_m2;
    const float4 *gbuffer_n;        // xyz = centre-ray normal ((0,0,1) on sky), w = hit code
    uint32_t *stats;                // [0] max m2 bits, [1] nonfinite, [2] any_valid, [3] bad
    uint32_t collect_stats;         // this frame closes a convergence window
    uint32_t tile_map;              // workgroup -> tile mapping (f3d_kernels.hip tile_pixel)
    uint32_t sample_lanes;          // lanes per pixel in the frame kernel: 1 (frame_pixel), 2, 4, 8 (frame_lanes)
    uint2 *head;                    // sample-lane form only: per-pixel record of k_head {reuse_w bits, flags}
    float2 *sun_clear;              // per pixel {parameter after which no sun ray of the pixel meets terrain, depth of the centre hit}; null = off
    uint2 *primary_start;           // per pixel {t_clear bits, level}: where its camera rays may start (f3d_cone.h); null = at the root
    // longest-first dispatch (f3d_kernels.hip k_tile_order): frame-kernel workgroup b renders tile tile_order[b]
    // (null: the tile_map formula); every wave leaves its duration in tile_cost[tile] for the next ordering
    const uint32_t *tile_order;
    uint32_t *tile_cost;
    unsigned long long *wave_times;  // diagnostics (builds with -DF3D_WAVE_TIMES): {start, end} clock per workgroup
    uint32_t band_begin, band_end;  // image rows THIS launch covers (a band of the strip; the host pipelines bands
                                    // of consecutive frames over several streams, f3d_host.hip)
    // frames in flight (f3d_kernels.hip k_trace / k_merge): two float4 per (frame, sample, pixel) -- record r of
    // sample s of frame f of strip pixel lp is trace[2 * (((f - trace_first) * spp + s) * pixels + lp) + {0, 1}]
    float4 *trace;
    uint32_t trace_first;
    uint32_t same_sun;  // the frame head's two sun directions (wi, normalize(wi)) are the same bits
    // pixel-frames whose sun direction was mispredicted: k_merge traces them again itself
    uint32_t *fix_list;   // (unused since round 4: round 3 listed the pixels for a second kernel)
    uint32_t *fix_count;  // [2]: running total (diagnostics)
    WfQueues wf;          // wavefront form of the trace batch (sun_o == null: k_trace traces the rays itself)
};

}  // namespace f3d
