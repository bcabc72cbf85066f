// forge3d_amd/csrc/f3d_aether_ref.hip -- the AETHER acceptance reference on gfx950 (C ABI f3d_aether_reference_render,
// include/f3d_terrain_pt.h).  Three launches: the per-sample stream positions of every pixel (sequential in the
// sample index, a lane per pixel), one lane per (pixel, sample, wavelength) path, and the ordered fold of each pixel's
// values (f3d_aether_ref.h).  The DEM's acceleration tables are the terrain tracer's (scene cache, f3d_tables.h).
#include <hip/hip_runtime.h>

#include <cmath>
#include <exception>
#include <vector>

#include "../../include/f3d_terrain_pt.h"
#include "f3d_shade.h"

#include "f3d_aether_ref.h"
#include "f3d_aether_ref_host.h"
#include "f3d_devmem.h"
#include "f3d_lds.h"
#include "f3d_tables.h"

using namespace f3d;
using namespace f3d::aref;

namespace {

struct RefKernelParams {
    RefScene S;
    uint32_t *states;  // [pixel][sample]
    float *values;     // [pixel][sample][wavelength]
    uint32_t *hits;    // [pixel][sample]
    float4 *accum;     // [pixel] sum_xyz, primary hits
    float2 *welford;   // [pixel] mean_y, m2_y
    uint32_t pixels;
};

__global__ __launch_bounds__(64) void k_ref_states(const RefKernelParams P) {
    const uint32_t pixel = blockIdx.x * blockDim.x + threadIdx.x;
    if (pixel >= P.pixels) return;
    uint32_t state = pixel_seed(P.S, pixel % P.S.cam.width, pixel / P.S.cam.width);
    for (uint32_t s = 0u; s < P.S.spp; s++) {
        P.states[(size_t)pixel * P.S.spp + s] = state;
        rng_skip(state, kDrawsPerSample);
    }
}

__global__ __launch_bounds__(kWave) void k_ref_paths(const RefKernelParams P) {
    __shared__ uint32_t lds[kLdsWords];
    LdsPending pend = make_pending(lds, P.S.terrain);
    const size_t path = (size_t)blockIdx.x * kWave + threadIdx.x;
    const size_t total = (size_t)P.pixels * P.S.spp * kWavelengths;
    if (path >= total) return;
    const uint32_t w = (uint32_t)(path % kWavelengths);
    const size_t ps = path / kWavelengths;  // pixel * spp + sample
    const uint32_t pixel = (uint32_t)(ps / P.S.spp);
    bool primary_hit;
    const float value = sample_path(P.S, pixel % P.S.cam.width, pixel / P.S.cam.width, P.states[ps], w, w == 0u, primary_hit, pend);
    P.values[path] = value;
    if (w == 0u) P.hits[ps] = primary_hit ? 1u : 0u;
}

__global__ __launch_bounds__(64) void k_ref_fold(const RefKernelParams P) {
    const uint32_t pixel = blockIdx.x * blockDim.x + threadIdx.x;
    if (pixel >= P.pixels) return;
    float out4[4], wf[2];
    fold_pixel(P.values + (size_t)pixel * P.S.spp * kWavelengths, P.hits + (size_t)pixel * P.S.spp, P.S.spp, out4, wf);
    P.accum[pixel] = float4{out4[0], out4[1], out4[2], out4[3]};
    P.welford[pixel] = float2{wf[0], wf[1]};
}

void ok(hipError_t e, const char *what) {
    if (e != hipSuccess) fail(F3D_STATUS_DEVICE, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace

extern "C" int f3d_aether_reference_render(const f3d_aether_ref_desc *desc, f3d_aether_ref_out *out, char *err, size_t errlen) {
    if (err && errlen) err[0] = 0;
    std::vector<void *> owned;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = F3D_STATUS_OK;
    try {
        if (!desc || !out || !out->mean_xyz || !out->linear_rgb) fail(F3D_STATUS_VALUE, "null argument");
        if (desc->struct_size != sizeof(f3d_aether_ref_desc))
            fail(F3D_STATUS_VALUE, "f3d_aether_ref_desc.struct_size is %u, this library (ABI %u) expects %zu", desc->struct_size, F3D_ABI_VERSION,
                 sizeof(f3d_aether_ref_desc));
        const f3d_aether_ref_desc &d = *desc;
        validate_ref_desc(d);
        check_ref_terrain(d);
        const size_t pixels = (size_t)d.width * d.height;
        out->variance = 0.0f;
        out->terrain_primary_hits = 0;
        out->gpu_resource_bytes = 0;
        out->kernel_seconds = 0.0;
        if (!d.enabled) {  // explicit black, :430-434
            std::fill(out->mean_xyz, out->mean_xyz + 3 * pixels, 0.0f);
            std::fill(out->linear_rgb, out->linear_rgb + 3 * pixels, 0.0f);
            out->converged = 1;
            return F3D_STATUS_OK;
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) fail(F3D_STATUS_DEVICE, "no HIP device available: libf3dhip has no CPU fallback");
        auto alloc = [&](size_t bytes, const char *what) {
            void *p = nullptr;
            ok(device_alloc(&p, bytes), what);
            owned.push_back(p);
            return p;
        };

        RefKernelParams P{};
        SharedTerrain terrain = acquire_shared_terrain(d.heights, d.dem_width, d.dem_height, d.exaggeration, nullptr);
        P.S.terrain = terrain.dev;
        fill_ref_scene(d, P.S);
        P.pixels = (uint32_t)pixels;

        const size_t samples = pixels * d.spp, paths = samples * kWavelengths;
        P.states = (uint32_t *)alloc(samples * sizeof(uint32_t), "stream positions");
        P.values = (float *)alloc(paths * sizeof(float), "path values");
        P.hits = (uint32_t *)alloc(samples * sizeof(uint32_t), "primary hits");
        P.accum = (float4 *)alloc(pixels * sizeof(float4), "accumulation");
        P.welford = (float2 *)alloc(pixels * sizeof(float2), "welford");
        out->gpu_resource_bytes = terrain.bytes + samples * 8u + paths * 4u + pixels * 24u;

        ok(hipEventCreate(&e0), "event");
        ok(hipEventCreate(&e1), "event");
        ok(hipEventRecord(e0, nullptr), "event");
        hipLaunchKernelGGL(k_ref_states, dim3((unsigned)((pixels + 63u) / 64u)), dim3(64), 0, nullptr, P);
        hipLaunchKernelGGL(k_ref_paths, dim3((unsigned)((paths + kWave - 1u) / kWave)), dim3(kWave), 0, nullptr, P);
        hipLaunchKernelGGL(k_ref_fold, dim3((unsigned)((pixels + 63u) / 64u)), dim3(64), 0, nullptr, P);
        ok(hipEventRecord(e1, nullptr), "event");
        ok(hipEventSynchronize(e1), "AETHER spectral reference");
        ok(hipGetLastError(), "AETHER spectral reference kernels");
        float ms = 0.0f;
        ok(hipEventElapsedTime(&ms, e0, e1), "event");
        out->kernel_seconds = ms * 1e-3;

        std::vector<float> accum(4 * pixels), welford(2 * pixels);
        ok(hipMemcpy(accum.data(), P.accum, pixels * sizeof(float4), hipMemcpyDeviceToHost), "accumulation read-back");
        ok(hipMemcpy(welford.data(), P.welford, pixels * sizeof(float2), hipMemcpyDeviceToHost), "welford read-back");
        finalize_ref(d, accum.data(), welford.data(), *out);
    } catch (const Failure &f) {
        rc = f.status;
        if (err && errlen) snprintf(err, errlen, "%s", f.message.c_str());
    } catch (const std::exception &e) {
        rc = F3D_STATUS_DEVICE;
        if (err && errlen) snprintf(err, errlen, "host failure: %s", e.what());
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (void *p : owned) (void)device_free(p);
    return rc;
}
