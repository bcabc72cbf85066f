// forge3d_amd/csrc/f3d_launch.h -- host-callable launchers of the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "f3d_aether.h"
#include "f3d_build.h"
#include "f3d_scene.h"

namespace f3d {

struct RayBatchParams {
    TerrainDev terrain;
    const float4 *rays;  // 2 float4 per ray: (origin, tmin), (direction, tmax)
    uint32_t n;
    uint32_t any_hit, apply_curvature;  // any_hit: 0 closest / 1 any-hit (sorted descent), 2 any / 3 closest (march)
    uint32_t start_in_cell;             // march only: start in the origin cell instead of at the root
    uint32_t *out_hit;
    float *out_t;
    float *out_normal;  // 3 per ray
};

struct ResolveParams {
    FrameParams frame;  // res_in = final temporal output, frame_index unused
    uint32_t frames;
    uint8_t *rgba;
    float *albedo, *normal;
    AetherDev aether;    // enabled = 0: the plain Reinhard resolve
    const float *depth;  // frame-0 depth AOV (the aerial-perspective post's segment length)
};

hipError_t launch_head(const FrameParams &p, hipStream_t stream);  // sample-lane form: before launch_frame
hipError_t launch_frame(const FrameParams &p, int variant, hipStream_t stream);
hipError_t launch_trace(const FrameParams &p, uint32_t frames, hipStream_t stream);  // frames in flight: a batch of frames
hipError_t launch_trace_wavefront(const FrameParams &p, uint32_t frames, uint32_t quorum, hipStream_t stream);  // same records, rays through queues
hipError_t launch_merge(const FrameParams &p, hipStream_t stream);
hipError_t launch_trace_init(const FrameParams &p, hipStream_t stream);               // first sun-direction predictions                   // ... and the ordered half of one
// longest-first dispatch of the frame kernel: tile count (and grid) of the current band, and the ordering kernel
uint32_t frame_tile_count(const FrameParams &p, uint32_t *grid);
hipError_t launch_tile_order(const FrameParams &p, const uint32_t *cost, uint32_t *order, hipStream_t stream);
hipError_t launch_gbuffer(const FrameParams &p, float4 *gbuffer_n, float *depth, hipStream_t stream);
hipError_t launch_resolve(const ResolveParams &p, hipStream_t stream);
hipError_t launch_ray_batch(const RayBatchParams &p, hipStream_t stream);
hipError_t launch_leaf_build(const PyramidBuildParams &p, hipStream_t stream);
hipError_t launch_level_build(const LevelBuildParams &p, hipStream_t stream);
hipError_t launch_band_build(const BandBuildParams &p, hipStream_t stream);
// far-horizon table of the IBL rays (f3d_cone.h): block level and block counts for a cell grid; the build (terrain
// must hold layout, bands, origin and spacing); table = bx * bz * 8 floats
void horizon_table_dims(uint32_t cell_w, uint32_t cell_h, uint32_t *level, uint32_t *bx, uint32_t *bz);
hipError_t launch_horizon_build(const TerrainDev &terrain, float *table, hipStream_t stream);

}  // namespace f3d
