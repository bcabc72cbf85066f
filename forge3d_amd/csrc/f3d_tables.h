// forge3d_amd/csrc/f3d_tables.h -- the terrain acceleration tables of a DEM for other translation units of the library
// (the PBR tracer's heightfield primitive, f3d_wavefront.hip).  Implemented in f3d_host.hip on top of the scene cache:
// a DEM the terrain tracer has rendered is not uploaded or built again.
#pragma once

#include <memory>

#include "f3d_scene.h"

namespace f3d {

struct SharedTerrain {
    std::shared_ptr<void> keep;  // keeps the cached tables alive
    TerrainDev dev;              // layout + table pointers (placement scalars are the caller's)
    uint64_t bytes = 0;
};
// Tables for `heights` (w x h, times exaggeration) on the current device; throws Failure like the terrain tracer.
SharedTerrain acquire_shared_terrain(const float *heights, uint32_t w, uint32_t h, float exaggeration, hipStream_t stream);

}  // namespace f3d
