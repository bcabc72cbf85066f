// forge3d_amd/csrc/f3d_lbvh.h -- GPU linear-BVH builder (f3d_lbvh.hip) for the optional mesh.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

#include "f3d_scene.h"

namespace f3d {

struct LbvhResult {
    BvhNode *nodes = nullptr;  // device, threaded preorder (f3d_bvh.h layout); owned by the caller (hipFree)
    float4 *tris = nullptr;    // device, 3 float4 per triangle in leaf order, original index in v0.w
    uint32_t node_count = 0, tri_count = 0;
    size_t node_bytes = 0, tri_bytes = 0;
};

// d_vertices: xyz + pad per vertex; d_indices: 3 per triangle (triangles with an index out of range are left out,
// like the reference's sweep skips them).  Synchronises `stream`.
hipError_t build_mesh_lbvh(const float4 *d_vertices, uint32_t vertex_count, const uint32_t *d_indices, uint32_t index_count,
                           hipStream_t stream, LbvhResult *result);

}  // namespace f3d
