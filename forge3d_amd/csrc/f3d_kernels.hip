// forge3d_amd/csrc/f3d_kernels.hip -- gfx950 kernels of the terrain path tracer.
// Host-callable launchers; the kernels themselves are in f3d_frame.h.
#include "f3d_frame.h"

namespace f3d {

// ---- launchers ---------------------------------------------------------------------
void horizon_table_dims(uint32_t cell_w, uint32_t cell_h, uint32_t *level, uint32_t *bx, uint32_t *bz) {
    *level = horizon_block_level(cell_w, cell_h);
    *bx = (cell_w + (1u << *level) - 1u) >> *level;
    *bz = (cell_h + (1u << *level) - 1u) >> *level;
}
hipError_t launch_horizon_build(const TerrainDev &terrain, float *table, hipStream_t stream) {
    HorizonBuildParams B{};
    B.terrain = terrain;
    B.terrain.horizon = nullptr;
    horizon_table_dims(terrain.cell_w, terrain.cell_h, &B.level, &B.bx, &B.bz);
    B.table = table;
    hipLaunchKernelGGL(k_horizon_build, dim3(((B.bx + 7u) >> 3) * ((B.bz + 7u) >> 3)), dim3(kWave), 0, stream, B);
    return hipGetLastError();
}
static inline uint32_t frame_grid(const FrameParams &p, uint32_t lanes = 1u) {
    const uint32_t log_s = lanes == 1u ? 0u : (lanes == 2u ? 1u : (lanes == 4u ? 2u : 3u));
#if defined(F3D_TILE_LOGW_S4)
    const uint32_t log_w = lanes <= 2u ? 3u : (lanes == 4u ? F3D_TILE_LOGW_S4 : 2u), log_h = 6u - log_s - log_w;
#else
    const uint32_t log_w = lanes <= 2u ? 3u : 2u, log_h = 6u - log_s - log_w;  // TileShape<S>
#endif
    const uint32_t rows = p.band_end - p.band_begin;
    const uint32_t tiles_x = (p.cam.width + (1u << log_w) - 1u) >> log_w, tiles_y = (rows + (1u << log_h) - 1u) >> log_h;
    if (tiles_y == 0u) return 0u;
    if (p.tile_map == 2u) return ((tiles_y + kNumXcd - 1u) / kNumXcd) * tiles_x * kNumXcd;
    return ((tiles_x * tiles_y + kNumXcd - 1u) / kNumXcd) * kNumXcd;
}

hipError_t launch_head(const FrameParams &p, hipStream_t stream) {  // the band's pixels, one lane each (8x8 tiles)
    if (frame_grid(p) == 0u) return hipSuccess;
    hipLaunchKernelGGL(k_head, dim3(frame_grid(p)), dim3(kWave), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_frame(const FrameParams &p, int variant, hipStream_t stream) {
    const uint32_t lanes = p.sample_lanes ? p.sample_lanes : 1u;
    if (frame_grid(p, lanes) == 0u) return hipSuccess;  // an empty band
    const dim3 grid(frame_grid(p, lanes)), block(kWave);
    // (F3D_FORCE_MESH_KERNEL=1: A/B switch, the mesh-capable instantiation for a terrain-only scene -- same results)
    static const bool force_mesh = getenv("F3D_FORCE_MESH_KERNEL") != nullptr;
    const bool mesh = p.mesh.traversal_mode == 0u || force_mesh;
    if (lanes != 1u) {  // one wave per workgroup (register-budget A/B variants for 4 and 8 lanes only)
        switch (lanes * 1000 + (uint32_t)(variant % 1000)) {
            case 2000:
                if (mesh) hipLaunchKernelGGL((k_frame<0, 6, 2, true>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((k_frame<0, 6, 2>), grid, block, 0, stream, p);
                break;
            case 4000:
                if (mesh) hipLaunchKernelGGL((k_frame<0, 6, 4, true>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((k_frame<0, 6, 4>), grid, block, 0, stream, p);
                break;
            case 4105:
                if (mesh) hipLaunchKernelGGL((k_frame<0, 5, 4, true>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((k_frame<0, 5, 4>), grid, block, 0, stream, p);
                break;
            case 8000:
                if (mesh) hipLaunchKernelGGL((k_frame<0, 6, 8, true>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((k_frame<0, 6, 8>), grid, block, 0, stream, p);
                break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (variant % 1000) {
        case 0:
            if (mesh) hipLaunchKernelGGL((k_frame<0, 6, 1u, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_frame<0, 6>), grid, block, 0, stream, p);
            break;  // default: 80 VGPRs, 6 waves/SIMD
        case 104:
            if (mesh) hipLaunchKernelGGL((k_frame<0, 4, 1u, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_frame<0, 4>), grid, block, 0, stream, p);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// frames [p.frame_index, p.frame_index + frames) of the strip into p.trace (grid.y = frame)
hipError_t launch_trace(const FrameParams &p, uint32_t frames, hipStream_t stream) {
    const uint32_t lanes = p.sample_lanes ? p.sample_lanes : 1u;
    if (frame_grid(p, lanes) == 0u || frames == 0u) return hipSuccess;
    const dim3 grid(frame_grid(p, lanes), frames), block(kWave);
    const bool mesh = p.mesh.traversal_mode == 0u;
    switch (lanes) {
        case 1:
            if (mesh) hipLaunchKernelGGL((k_trace<6, 1, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_trace<6, 1>), grid, block, 0, stream, p);
            break;
        case 2:
            if (mesh) hipLaunchKernelGGL((k_trace<6, 2, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_trace<6, 2>), grid, block, 0, stream, p);
            break;
        case 4:
            if (mesh) hipLaunchKernelGGL((k_trace<6, 4, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_trace<6, 4>), grid, block, 0, stream, p);
            break;
        case 8:
            if (mesh) hipLaunchKernelGGL((k_trace<6, 8, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((k_trace<6, 8>), grid, block, 0, stream, p);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// the wavefront form of launch_trace: primaries + queues, then the three queues through persistent waves
template <class K>
static uint32_t persistent_grid(K kernel) {
    int device = 0, cus = 256, per_cu = 0;
    (void)hipGetDevice(&device);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kWave, 0) != hipSuccess || per_cu <= 0) per_cu = 16;
    return (uint32_t)cus * (uint32_t)per_cu;
}
hipError_t launch_trace_wavefront(const FrameParams &p, uint32_t frames, uint32_t quorum, hipStream_t stream) {
    const uint32_t lanes = p.sample_lanes ? p.sample_lanes : 1u;
    if (frame_grid(p, lanes) == 0u || frames == 0u) return hipSuccess;
    hipError_t err = hipMemsetAsync(p.wf.cursors, 0, 4u * sizeof(uint32_t), stream);
    if (err != hipSuccess) return err;
    const dim3 grid(frame_grid(p, lanes), frames), block(kWave);
    switch (lanes) {
        case 1: hipLaunchKernelGGL((k_wf_primary<6, 1>), grid, block, 0, stream, p); break;
        case 2: hipLaunchKernelGGL((k_wf_primary<6, 2>), grid, block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL((k_wf_primary<6, 4>), grid, block, 0, stream, p); break;
        case 8: hipLaunchKernelGGL((k_wf_primary<6, 8>), grid, block, 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    static const uint32_t sun_grid = persistent_grid(k_wf_occl<true, 7>), ibl_grid = persistent_grid(k_wf_occl<false, 8>);
    WfOcclParams W{};
    W.terrain = p.terrain;
    W.trace = p.trace;
    W.counts = p.wf.counts;
    W.regions = frames * p.wf.regions_per_frame;
    W.quorum = quorum;
    for (uint32_t q = 0u; q < 2u; q++) {  // sun rays along light.wi (front of the regions), along light.wi_reuse (back)
        if (q == 1u && p.same_sun != 0u) break;  // (the same bits: k_wf_primary files every sun ray at the front then)
        W.ray_o = p.wf.sun_o;
        W.ray_d = nullptr;
        W.ray_stop = p.wf.sun_stop;
        W.cursor = p.wf.cursors + q;
        W.kind = q;
        W.dir = q ? p.light.wi_reuse : p.light.wi;
        hipLaunchKernelGGL((k_wf_occl<true, 7>), dim3(sun_grid), block, 0, stream, W);
    }
    W.ray_o = p.wf.ibl_o;
    W.ray_d = p.wf.ibl_d;
    W.ray_stop = nullptr;
    W.cursor = p.wf.cursors + 2u;
    W.kind = 2u;
    hipLaunchKernelGGL((k_wf_occl<false, 8>), dim3(ibl_grid), block, 0, stream, W);
    return hipGetLastError();
}
hipError_t launch_trace_init(const FrameParams &p, hipStream_t stream) {
    if (frame_grid(p) == 0u) return hipSuccess;
    hipLaunchKernelGGL(k_trace_init, dim3(frame_grid(p)), dim3(kWave), 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_merge(const FrameParams &p, hipStream_t stream) {
    if (frame_grid(p) == 0u) return hipSuccess;
    hipLaunchKernelGGL(k_merge, dim3(frame_grid(p)), dim3(kWave), 0, stream, p);  // (re-traces its mispredicted pixel-frames itself)
    return hipGetLastError();
}
hipError_t launch_tile_order(const FrameParams &p, const uint32_t *cost, uint32_t *order, hipStream_t stream) {
    const uint32_t lanes = p.sample_lanes ? p.sample_lanes : 1u;
    const uint32_t log_s = lanes == 1u ? 0u : (lanes == 2u ? 1u : (lanes == 4u ? 2u : 3u));
    const uint32_t log_w = lanes <= 2u ? 3u : 2u, log_h = 6u - log_s - log_w;  // TileShape<S>
    const uint32_t rows = p.band_end - p.band_begin;
    TileOrderParams B{cost, order, (p.cam.width + (1u << log_w) - 1u) >> log_w, (rows + (1u << log_h) - 1u) >> log_h};
    hipLaunchKernelGGL(k_tile_order, dim3(kNumXcd), dim3(1024), 0, stream, B);
    return hipGetLastError();
}
uint32_t frame_tile_count(const FrameParams &p, uint32_t *grid) {
    const uint32_t lanes = p.sample_lanes ? p.sample_lanes : 1u;
    const uint32_t log_s = lanes == 1u ? 0u : (lanes == 2u ? 1u : (lanes == 4u ? 2u : 3u));
    const uint32_t log_w = lanes <= 2u ? 3u : 2u, log_h = 6u - log_s - log_w;
    const uint32_t rows = p.band_end - p.band_begin;
    if (grid) *grid = frame_grid(p, lanes);
    return ((p.cam.width + (1u << log_w) - 1u) >> log_w) * ((rows + (1u << log_h) - 1u) >> log_h);
}
hipError_t launch_gbuffer(const FrameParams &p, float4 *gbuffer_n, float *depth, hipStream_t stream) {
    hipLaunchKernelGGL(k_gbuffer, dim3(frame_grid(p)), dim3(kWave), 0, stream, p, gbuffer_n, depth);
    return hipGetLastError();
}
hipError_t launch_resolve(const ResolveParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(k_resolve, dim3(frame_grid(p.frame)), dim3(kWave), 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_ray_batch(const RayBatchParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(k_ray_batch, dim3((p.n + kWave - 1) / kWave), dim3(kWave), 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_leaf_build(const PyramidBuildParams &p, hipStream_t stream) {
    dim3 block(16, 16), grid((p.leaf_dim_x + 15) / 16, (p.leaf_dim_y + 15) / 16);
    hipLaunchKernelGGL(k_leaf_build, grid, block, 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_band_build(const BandBuildParams &p, hipStream_t stream) {
    dim3 block(16, 16), grid((p.width + 15) / 16, (p.height + 15) / 16);
    hipLaunchKernelGGL(k_band_build, grid, block, 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_level_build(const LevelBuildParams &p, hipStream_t stream) {
    dim3 block(16, 16), grid((p.dst_dim_x + 15) / 16, (p.dst_dim_y + 15) / 16);
    hipLaunchKernelGGL(k_level_build, grid, block, 0, stream, p);
    return hipGetLastError();
}

}  // namespace f3d

#if defined(F3D_MESH_STATS)  // diagnostics build only (tools/experiments/c4_window.py): not part of the ABI
extern "C" int f3d_debug_mesh_stats(unsigned long long *out, int reset) {
    if (reset) {
        unsigned long long zero[8] = {};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(f3d::g_mesh_stats), zero, sizeof(zero));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(f3d::g_mesh_stats), 8 * sizeof(unsigned long long));
}
#endif
