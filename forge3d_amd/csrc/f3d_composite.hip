// forge3d_amd/csrc/f3d_composite.hip -- smoke-over-terrain composites on gfx950 (C ABI f3d_smoke_composite, include/
// f3d_terrain_pt.h): one pass over RGBA8 images, a lane per four pixels of a row so that every access of base, layer
// and output is a 16-byte one where the layer lines up (12 bytes of HBM traffic per pixel; the pass is traffic-bound
// and two orders of magnitude cheaper than the frames it combines).  Per-pixel arithmetic in f3d_composite.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <exception>
#include <vector>

#include "../../include/f3d_terrain_pt.h"
#include "f3d_composite.h"
#include "f3d_devmem.h"
#include "f3d_setup.h"

using namespace f3d;
using namespace f3d::composite;

namespace {

constexpr uint32_t kPixelsPerLane = 4u;

__global__ __launch_bounds__(256) void k_composite(const Params P, const uint32_t *__restrict__ base, const uint32_t *__restrict__ layer,
                                                   uint32_t *__restrict__ out) {
    const uint32_t x0 = (blockIdx.x * blockDim.x + threadIdx.x) * kPixelsPerLane, y = blockIdx.y;
    if (x0 >= P.width) return;
    const uint32_t n = P.width - x0 < kPixelsPerLane ? P.width - x0 : kPixelsPerLane;
    uint32_t px[kPixelsPerLane];
    for (uint32_t k = 0; k < kPixelsPerLane; k++)
        if (k < n) px[k] = pixel(P, base, layer, x0 + k, y);
    uint32_t *dst = out + (size_t)y * P.width + x0;
    if (n == kPixelsPerLane && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0u) {
        *reinterpret_cast<uint4 *>(dst) = uint4{px[0], px[1], px[2], px[3]};
    } else {
        for (uint32_t k = 0; k < n; k++) dst[k] = px[k];
    }
}

void ok(hipError_t e, const char *what) {
    if (e != hipSuccess) fail(F3D_STATUS_DEVICE, "%s: %s", what, hipGetErrorString(e));
}

struct Buffers {
    std::vector<void *> owned;
    ~Buffers() {
        for (void *p : owned) (void)device_free(p);
    }
    // a device view of `src`: the pointer itself when it already is device memory, else an uploaded copy
    const uint32_t *in(const uint8_t *src, size_t bytes) {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeDevice) return reinterpret_cast<const uint32_t *>(src);
        (void)hipGetLastError();
        void *d = nullptr;
        ok(device_alloc(&d, bytes), "composite input");
        owned.push_back(d);
        ok(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice), "composite upload");  // (a fresh allocation: nothing in flight reads it)
        return static_cast<const uint32_t *>(d);
    }
};

}  // namespace

extern "C" int f3d_smoke_composite(const f3d_composite_desc *desc, uint8_t *out_rgba, double *kernel_seconds, char *err, size_t errlen) {
    if (err && errlen) err[0] = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = F3D_STATUS_OK;
    try {
        if (!desc || !out_rgba) fail(F3D_STATUS_VALUE, "null argument");
        if (desc->struct_size != sizeof(f3d_composite_desc))
            fail(F3D_STATUS_VALUE, "f3d_composite_desc.struct_size is %u, this library (ABI %u) expects %zu", desc->struct_size, F3D_ABI_VERSION,
                 sizeof(f3d_composite_desc));
        const f3d_composite_desc &d = *desc;
        if (d.mode > F3D_COMPOSITE_OVER) fail(F3D_STATUS_VALUE, "unknown composite mode %u", d.mode);
        if (d.width == 0u || d.height == 0u) fail(F3D_STATUS_VALUE, "width and height must be >= 1");
        if (!d.base) fail(F3D_STATUS_VALUE, "base image is null");
        if (!d.layer && d.mode != F3D_COMPOSITE_SMOKE_MAPS) fail(F3D_STATUS_VALUE, "layer image is null");
        if (d.mode != F3D_COMPOSITE_OVER && d.layer) {
            if (d.layer_width != d.width || d.layer_height != d.height)  // numpy would refuse to broadcast
                fail(F3D_STATUS_VALUE, "layer is %ux%u, base is %ux%u: images must match", d.layer_width, d.layer_height, d.width, d.height);
            if (d.offset_x != 0 || d.offset_y != 0) fail(F3D_STATUS_VALUE, "an offset is only meaningful in F3D_COMPOSITE_OVER");
        }
        if (d.mode == F3D_COMPOSITE_OVER && (d.layer_width == 0u || d.layer_height == 0u)) fail(F3D_STATUS_VALUE, "layer width and height must be >= 1");
        if (d.mode == F3D_COMPOSITE_SMOKE_MAPS) {
            if (!std::isfinite(d.base_alpha) || !std::isfinite(d.layer_alpha)) fail(F3D_STATUS_VALUE, "alpha scales must be finite");
            if (d.max_alpha > 255u) fail(F3D_STATUS_VALUE, "max_alpha must be <= 255");
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) fail(F3D_STATUS_DEVICE, "no HIP device available: libf3dhip has no CPU fallback");

        Params P{};
        P.mode = d.mode;
        P.width = d.width;
        P.height = d.height;
        P.has_layer = d.layer ? 1u : 0u;
        P.layer_width = d.layer ? d.layer_width : 0u;
        P.layer_height = d.layer ? d.layer_height : 0u;
        P.offset_x = d.offset_x;
        P.offset_y = d.offset_y;
        P.base_alpha = d.base_alpha;
        P.layer_alpha = d.layer_alpha;
        P.max_alpha = d.max_alpha;
        P.max_alpha_fraction = (float)((double)d.max_alpha / 255.0);  // HYBRID_SMOKE_MAX_ALPHA / 255.0 in double, then float32

        Buffers buf;
        const size_t bytes = (size_t)d.width * d.height * 4u;
        const uint32_t *base = buf.in(d.base, bytes);
        const uint32_t *layer = d.layer ? buf.in(d.layer, (size_t)P.layer_width * P.layer_height * 4u) : nullptr;
        hipPointerAttribute_t attr{};
        const bool out_on_device = hipPointerGetAttributes(&attr, out_rgba) == hipSuccess && attr.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        uint32_t *out = reinterpret_cast<uint32_t *>(out_rgba);
        if (!out_on_device) {
            void *p = nullptr;
            ok(device_alloc(&p, bytes), "composite output");
            buf.owned.push_back(p);
            out = static_cast<uint32_t *>(p);
        }
        // device images on both sides and nobody asking for the kernel time: the call returns with its launch enqueued (the
        // inputs it read where they are must not change until the stream gets there -- a resident sequence's next call is
        // behind this one in the same stream)
        const bool timed = kernel_seconds != nullptr || !out_on_device || !buf.owned.empty() || !current_smoke_context()->async_ok;
        const dim3 block(256), grid((d.width + 256u * kPixelsPerLane - 1u) / (256u * kPixelsPerLane), d.height);
        if (timed) {
            ok(hipEventCreate(&e0), "event");
            ok(hipEventCreate(&e1), "event");
            ok(hipEventRecord(e0, call_stream()), "event");
        }
        hipLaunchKernelGGL(k_composite, grid, block, 0, call_stream(), P, base, layer, out);
        ok(hipGetLastError(), "composite kernel");
        if (timed) {
            ok(hipEventRecord(e1, call_stream()), "event");
            ok(hipEventSynchronize(e1), "composite");
            float ms = 0.0f;
            ok(hipEventElapsedTime(&ms, e0, e1), "event");
            if (kernel_seconds) *kernel_seconds = ms * 1e-3;
        }
        if (!out_on_device) {
            ok(hipStreamSynchronize(call_stream()), "composite");
            ok(hipMemcpy(out_rgba, out, bytes, hipMemcpyDeviceToHost), "composite read-back");
        }
    } catch (const Failure &f) {
        rc = f.status;
        if (err && errlen) snprintf(err, errlen, "%s", f.message.c_str());
    } catch (const std::exception &e) {
        rc = F3D_STATUS_DEVICE;
        if (err && errlen) snprintf(err, errlen, "host failure: %s", e.what());
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}
