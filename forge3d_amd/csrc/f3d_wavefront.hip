// forge3d_amd/csrc/f3d_wavefront.hip -- multi-bounce PBR path tracer on gfx950: kernels + host driver + C ABI
// (include/f3d_wavefront.h).  The per-pixel path logic lives in f3d_wf_path.h (host + device, so
// the test emulator runs the same code); this file owns the launch shape and the scene upload.
//
// Reference driver: render_pt_reference (src/path_tracing/adjudication.rs:76-364) submits, for each of the spp frames,
// raygen + up to 16 x {intersect, shade, shadow, scatter} dispatches of ceil(4 W H / 256) workgroups each and maps the
// queue header back to the host between bounces.  Here a round of (up to thousands of) frames is two launches: a
// producer that traces every (pixel, frame) path with all the parallelism W x H x frames offers and leaves one
// 16-byte total per pixel-frame in HBM, and a streaming fold that adds them per pixel in frame order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <vector>

// This translation unit's march keeps one-word leaf-FIFO entries (f3d_march.h: the drain forms a leaf's interval again, ~20
// instructions a drained leaf): 5 rows of the wave's LDS block instead of 15, which is what lets the path's whole loop-carried
// state live in LDS rows instead of scratch (f3d_wf_path.h LaneState) at the same 6 400 bytes a wave.
#ifndef F3D_FIFO_WORDS
#define F3D_FIFO_WORDS 1
#endif
#ifndef F3D_WF_PARK
#define F3D_WF_PARK 18
#endif
#include "../../include/f3d_wavefront.h"
#include "f3d_devmem.h"
#include "f3d_lds.h"
#include "f3d_tables.h"
#include "f3d_wf_host.h"

using namespace f3d;

namespace {

struct WfParams {
    wf::SceneDev S;
    float4 *totals;  // [frame - first][pixel]: a frame's contributions, summed in stage order (w unused)
    uint32_t first, count, frames_per_lane;
    unsigned long long *vertices;
};

// Producer.  One wave = one 8x8 pixel tile x one group of `frames_per_lane` consecutive frames; a lane follows its
// pixel's paths of those frames in the flat loop of f3d_wf_path.h and writes one 16-byte total per frame.  Frame totals
// do not depend on the pixel's running sum, so there are W x H x frames / frames_per_lane independent lanes -- at the
// adjudication gate's 512 x 512 one lane per pixel would be 4 096 waves for 1 024 SIMDs, the sky ones done at once.
// Workgroup index = group * tiles + tile: consecutive workgroups are neighbouring tiles, dealt round robin over the XCDs.
#ifndef F3D_WF_WAVES
#define F3D_WF_WAVES 4
#endif
// The instantiation with the heightfield primitive waits for scattered table records more than it computes: six waves a SIMD
// at 80 registers and 284 bytes of spills beat four at 128 and 92 (C3 GI, 1080p x 32 paths: 4 / 5 / 6 / 7 / 8 waves ->
// 23.6 / 22.9 / 22.1 / 21.8-22.7 / 22.2-22.4 ms; its LDS block is the compact one of f3d_lds.h so that the waves fit).  Without
// the primitive the BLAS walk's spills decide: 4 / 5 / 6 -> 137.6 / 178.6 / 203.8 ms at the adjudication gate.
#ifndef F3D_WF_FRAME_LANES
#define F3D_WF_FRAME_LANES 4  // lanes a pixel in the kernels with the heightfield primitive (f3d_wf_path.h HipWave)
#endif
#ifndef F3D_WF_FRAME_LANES_PLAIN
#define F3D_WF_FRAME_LANES_PLAIN 4  // ... and in the kernel without it (adjudication gate, 512 x 512 x 4096: 1 / 2 / 4 / 8 lanes -> 139.8 / 139.2 / 135.7 / 134.8 ms)
#endif
#ifndef F3D_WF_WAVES_TERRAIN
#define F3D_WF_WAVES_TERRAIN 6
#endif
// LITE (round 5): a scene of spheres + heightfield + environment + directional lights only -- what render_terrain_gi builds
// (BASELINE.json configs[2]) -- runs an instantiation without the BLAS walk, the hair segments, the area lights and the fog:
// their live ranges were part of why the per-vertex state of the full kernel lives in scratch (284 B a lane).
template <bool TERRAIN, bool LITE = false>  // scenes with the heightfield primitive (f3d_wf_path.h) run their own instantiation
__global__ __launch_bounds__(64, TERRAIN ? F3D_WF_WAVES_TERRAIN : F3D_WF_WAVES) void k_wf_paths(const WfParams P) {
    // traversal context of the terrain march (f3d_lds.h)
    __shared__ __attribute__((aligned(16))) uint32_t lds[TERRAIN ? kCompactLdsWords : 1];
    LdsPendingCompact pend{};
    if (TERRAIN) pend = make_pending<LdsPendingCompact>(lds, P.S.terrain, kCompactRows);
    using Wave = wf::HipWave<LdsPendingCompact, TERRAIN, LITE, TERRAIN ? (uint32_t)F3D_WF_FRAME_LANES : (uint32_t)F3D_WF_FRAME_LANES_PLAIN>;
    constexpr uint32_t kFL = Wave::kFrameStride, kTW = Wave::kTileW, kTH = Wave::kTileH;
    const uint32_t tiles_x = (P.S.width + kTW - 1u) / kTW, tiles = tiles_x * ((P.S.height + kTH - 1u) / kTH);
    // (all frame groups of a tile together and image rows from the bottom up -- heavy tiles first -- measured 21.3-21.4 against
    // 21.4-21.6 ms on C3 GI: the tail is not what the round waits for; not taken)
    const uint32_t tile = blockIdx.x % tiles, group = blockIdx.x / tiles;
    const uint32_t x0 = (tile % tiles_x) * kTW, y0 = (tile / tiles_x) * kTH;
    const uint32_t turn = threadIdx.x % kFL, x = x0 + (threadIdx.x / kFL) % kTW, y = y0 + threadIdx.x / (kFL * kTW);
    // the wave's frames: kFL * frames_per_lane consecutive ones, the lanes of a pixel taking them in turns
    const uint32_t begin = group * (kFL * P.frames_per_lane);
    uint32_t vertices = 0u;
    if (x < P.S.width && y < P.S.height && begin + turn < P.count) {
        const uint32_t all = P.count - begin < kFL * P.frames_per_lane ? P.count - begin : kFL * P.frames_per_lane;
        const uint32_t n = (all - turn + kFL - 1u) / kFL;
        const Wave wave{&pend, x0, y0, P.S.width};
        // (the output address is formed from the lane's pixel where a frame ends, like everything else that depends on it:
        // nothing per-lane but the march's own state is alive across a march)
        vertices = wf::trace_frames(P.S, P.first + begin + turn, n, wave, [&](uint32_t frame, V3 total) {
            const size_t pixels = (size_t)P.S.width * P.S.height;
            P.totals[(size_t)(frame - P.first) * pixels + wave.pixel()] = float4{total.x, total.y, total.z, 0.0f};
        });
    }
    unsigned long long total = vertices;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if (threadIdx.x == 0u && P.vertices) atomicAdd(P.vertices, total);
}

struct FoldParams {
    const float4 *totals;
    float4 *accum;
    uint32_t pixels, count;
};

// Consumer: the frame totals of a pixel are added to its running sum in frame order (the one order-sensitive step).
// Lanes = pixels, loads are coalesced and independent of the adds; reads count x 16 B per pixel at HBM speed.
__global__ __launch_bounds__(256) void k_wf_fold(const FoldParams F) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= F.pixels) return;
    float4 a = F.accum[p];
    const float4 *t = F.totals + p;
    uint32_t f = 0u;
    for (; f + 8u <= F.count; f += 8u) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = t[(size_t)(f + k) * F.pixels];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a.x = a.x + v[k].x;
            a.y = a.y + v[k].y;
            a.z = a.z + v[k].z;
        }
    }
    for (; f < F.count; f++) {
        const float4 v = t[(size_t)f * F.pixels];
        a.x = a.x + v.x;
        a.y = a.y + v.y;
        a.z = a.z + v.z;
    }
    F.accum[p] = a;
}

struct ResolveParamsWf {
    const float4 *accum;
    float4 *hdr;
    uchar4 *rgba;
    uint32_t pixels, frames;
    float exposure;
};

// mean over frames, alpha 1 (adjudication.rs:296-306); Reinhard + piecewise sRGB + u8 (core/tonemap.rs:11-30)
__global__ void k_wf_resolve(const ResolveParamsWf R) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R.pixels) return;
    const float inv = 1.0f / (float)R.frames;
    const float4 a = R.accum[i];
    const float m[3] = {a.x * inv, a.y * inv, a.z * inv};
    if (R.hdr) R.hdr[i] = float4{m[0], m[1], m[2], 1.0f};
    if (R.rgba) {
        uint8_t c[3];
        for (int k = 0; k < 3; k++) {
            const float x = f_max(m[k], 0.0f) * R.exposure;
            const float t = x / (1.0f + x);
            const float s = t <= 0.0031308f ? 12.92f * t : 1.055f * pow_det(t, 1.0f / 2.4f) - 0.055f;
            c[k] = (uint8_t)(f_clamp(s, 0.0f, 1.0f) * 255.0f + 0.5f);
        }
        R.rgba[i] = uchar4{c[0], c[1], c[2], 255};
    }
}

void ok(hipError_t e, const char *what) {
    if (e != hipSuccess) fail(F3D_STATUS_DEVICE, "HIP failure in %s: %s", what, hipGetErrorString(e));
}

struct DeviceScope {  // the caller's current device is restored on every exit path
    int before = -1;
    bool switched = false;
    explicit DeviceScope(int want) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
            fail(F3D_STATUS_DEVICE, "no HIP device available: libf3dhip has no CPU fallback");
        if (want >= 0) {
            if (want >= n) fail(F3D_STATUS_VALUE, "device %d out of range (%d devices)", want, n);
            ok(hipGetDevice(&before), "hipGetDevice");
            if (before != want) {
                ok(hipSetDevice(want), "hipSetDevice");
                switched = true;
            }
        }
    }
    ~DeviceScope() {
        if (switched) (void)hipSetDevice(before);
    }
};

}  // namespace

extern "C" int f3d_wavefront_render(const f3d_wf_scene *scene_in, uint32_t width, uint32_t height, uint32_t first_frame,
                                    uint32_t frame_count, uint32_t frames_per_launch, int32_t device, f3d_wf_out *out, char *err,
                                    size_t errlen) {
    if (err && errlen) err[0] = 0;
    std::vector<void *> owned;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = F3D_STATUS_OK;
    try {
        if (!scene_in || !out) fail(F3D_STATUS_VALUE, "null argument");
        const f3d_wf_scene scene_now = wf::scene_of_caller(scene_in);
        const f3d_wf_scene *scene = &scene_now;
        wf::validate_scene(*scene, width, height, frame_count);
        if ((uint64_t)first_frame + frame_count > 0xFFFFFFFFull) fail(F3D_STATUS_VALUE, "frame range overflows u32");
        DeviceScope scope(device);
        auto alloc = [&](size_t bytes, const char *what) {
            void *p = nullptr;
            ok(device_alloc(&p, bytes), what);
            owned.push_back(p);
            return p;
        };
        auto upload = [&](const void *src, size_t bytes, const char *what) {
            void *p = alloc(bytes, what);
            if (bytes) ok(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice), what);
            return p;
        };
        wf::PreparedScene prep = wf::prepare_scene(*scene, width, height);
        WfParams P{};
        P.S = prep.S;
        wf::SceneDev &S = P.S;
        S.spheres = (const wf::SphereDev *)upload(prep.spheres.data(), prep.spheres.size() * sizeof(wf::SphereDev), "spheres");
        S.mats = (const wf::MaterialDev *)upload(prep.mats.data(), prep.mats.size() * sizeof(wf::MaterialDev), "materials");
        std::vector<wf::BlasDev> blas(prep.bvh.size());
        for (size_t m = 0; m < prep.bvh.size(); m++) {  // one threaded BVH per BLAS
            const MeshBvh &bvh = prep.bvh[m];
            blas[m].nodes = (const BvhNode *)upload(bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode), "bvh nodes");
            blas[m].tris = (const float4 *)upload(bvh.tris.data(), bvh.tris.size() * sizeof(float), "bvh triangles");
            blas[m].node_count = (uint32_t)bvh.nodes.size();
            blas[m].pad = 0u;
        }
        S.blas = (const wf::BlasDev *)upload(blas.data(), blas.size() * sizeof(wf::BlasDev), "blas table");
        S.inst = (const wf::InstanceDev *)upload(prep.inst.data(), prep.inst.size() * sizeof(wf::InstanceDev), "instances");
        S.dir = (const wf::DirLightDev *)upload(prep.dir.data(), prep.dir.size() * sizeof(wf::DirLightDev), "directional lights");
        S.area = (const wf::AreaLightDev *)upload(prep.area.data(), prep.area.size() * sizeof(wf::AreaLightDev), "area lights");
        S.hair = (const wf::HairDev *)upload(prep.hair.data(), prep.hair.size() * sizeof(wf::HairDev), "hair segments");
        SharedTerrain terrain;  // the heightfield primitive: the terrain tracer's tables (scene cache), its placement from prepare_scene
        if (scene->terrain) {
            terrain = acquire_shared_terrain(scene->terrain->heights, scene->terrain->dem_width, scene->terrain->dem_height,
                                             scene->terrain->exaggeration, nullptr);
            const TerrainDev placed = S.terrain;
            S.terrain = terrain.dev;
            S.terrain.origin_x = placed.origin_x;
            S.terrain.origin_z = placed.origin_z;
            S.terrain.spacing_x = placed.spacing_x;
            S.terrain.spacing_z = placed.spacing_z;
            S.terrain.inv_spacing_x = placed.inv_spacing_x;
            S.terrain.inv_spacing_z = placed.inv_spacing_z;
        }

        const size_t pixels = (size_t)width * height;
        float4 *d_accum = (float4 *)alloc(pixels * sizeof(float4), "accumulation");
        if (out->accum)
            ok(hipMemcpy(d_accum, out->accum, pixels * sizeof(float4), hipMemcpyHostToDevice), "accumulation upload");
        else
            ok(hipMemset(d_accum, 0, pixels * sizeof(float4)), "accumulation clear");
        P.vertices = (unsigned long long *)alloc(sizeof(unsigned long long), "counter");
        ok(hipMemset(P.vertices, 0, sizeof(unsigned long long)), "counter clear");

        // Rounds: a round traces `round_frames` frames of every pixel into the totals buffer (16 B per pixel-frame, 4 GB
        // by default -- a sliver of the 288 GB) and folds them; lanes take `frames_per_lane` frames each: few enough
        // that a round has tens of thousands of waves, many enough that the lanes of a wave end their batches together.
        // (the kernels with the heightfield primitive: F3D_WF_FRAME_LANES lanes a pixel, tiles of 64 / that many pixels -- f3d_wf_path.h HipWave)
        const uint32_t frame_lanes = S.has_terrain ? (uint32_t)F3D_WF_FRAME_LANES : (uint32_t)F3D_WF_FRAME_LANES_PLAIN;
        const uint32_t tile_w = wf::wf_tile_w(frame_lanes), tile_h = wf::wf_tile_h(frame_lanes);
        const uint32_t tiles = ((width + tile_w - 1u) / tile_w) * ((height + tile_h - 1u) / tile_h);
        const uint64_t budget = 4ull << 30;
        uint32_t round_frames = frames_per_launch ? frames_per_launch : (uint32_t)std::min<uint64_t>(frame_count, std::max<uint64_t>(1, budget / (pixels * sizeof(float4))));
        round_frames = std::min(round_frames, frame_count);
        uint32_t fpl = 32u;  // measured at the gate (512 x 512 x 4096): 8 / 16 / 32 / 64 / 128 / 256 -> 150 / 147 / 145 / 148 / 157 / 176 ms
        // (with the heightfield primitive the waves are long and differ more -- sky tiles end at once -- so the tail of a round
        // asks for more, shorter ones: C3 GI at 1080p x 32 frames, 32 / 16 / 8 / 4 / 2 -> 24.4 / 22.3 / 21.4 / 21.9 / 24.1 ms)
        const uint64_t want_waves = S.has_terrain ? 100000ull : 32768ull;
        while (fpl > 8u && (uint64_t)tiles * ((round_frames + frame_lanes * fpl - 1u) / (frame_lanes * fpl)) < want_waves) fpl >>= 1;
        if (const char *e = getenv("F3D_WF_FRAMES_PER_LANE")) fpl = (uint32_t)std::max(1, atoi(e));
        fpl = std::min(std::min(fpl, round_frames), wf::kMaxFramesPerCall);  // (a lane's frame counter has 11 bits: f3d_wf_path.h LaneState)
        P.frames_per_lane = fpl;
        P.totals = (float4 *)alloc((size_t)round_frames * pixels * sizeof(float4), "frame totals");
        ok(hipEventCreate(&e0), "event");
        ok(hipEventCreate(&e1), "event");
        ok(hipEventRecord(e0, nullptr), "event");
        for (uint32_t done = 0u; done < frame_count; done += round_frames) {
            P.first = first_frame + done;
            P.count = std::min(round_frames, frame_count - done);
            const uint32_t groups = (P.count + frame_lanes * fpl - 1u) / (frame_lanes * fpl);
            const bool lite = S.has_terrain && S.blas_count == 0u && S.inst_count == 0u && S.hair_count == 0u && S.area_count == 0u && S.medium_on == 0u &&
                              getenv("F3D_WF_FULL_KERNEL") == nullptr;  // (A/B + test switch: the full instantiation for a lite scene -- same results)
            if (lite) hipLaunchKernelGGL((k_wf_paths<true, true>), dim3(tiles * groups), dim3(64), 0, nullptr, P);
            else if (S.has_terrain) hipLaunchKernelGGL(k_wf_paths<true>, dim3(tiles * groups), dim3(64), 0, nullptr, P);
            else hipLaunchKernelGGL(k_wf_paths<false>, dim3(tiles * groups), dim3(64), 0, nullptr, P);
            ok(hipGetLastError(), "path tracing kernel");
            const FoldParams F{P.totals, d_accum, (uint32_t)pixels, P.count};
            hipLaunchKernelGGL(k_wf_fold, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, nullptr, F);
            ok(hipGetLastError(), "fold kernel");
        }
        ok(hipEventRecord(e1, nullptr), "event");
        const uint32_t total_frames = first_frame + frame_count;
        float4 *d_hdr = out->hdr ? (float4 *)alloc(pixels * sizeof(float4), "hdr") : nullptr;
        uchar4 *d_rgba = out->rgba ? (uchar4 *)alloc(pixels * 4, "rgba") : nullptr;
        if (d_hdr || d_rgba) {
            const ResolveParamsWf R{d_accum, d_hdr, d_rgba, (uint32_t)pixels, total_frames, scene->cam_exposure};
            hipLaunchKernelGGL(k_wf_resolve, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, nullptr, R);
            ok(hipGetLastError(), "resolve kernel");
        }
        ok(hipDeviceSynchronize(), "path tracing");
        if (d_hdr) ok(hipMemcpy(out->hdr, d_hdr, pixels * sizeof(float4), hipMemcpyDeviceToHost), "hdr readback");
        if (d_rgba) ok(hipMemcpy(out->rgba, d_rgba, pixels * 4, hipMemcpyDeviceToHost), "rgba readback");
        if (out->accum) ok(hipMemcpy(out->accum, d_accum, pixels * sizeof(float4), hipMemcpyDeviceToHost), "accumulation readback");
        unsigned long long vertices = 0;
        ok(hipMemcpy(&vertices, P.vertices, sizeof(vertices), hipMemcpyDeviceToHost), "counter readback");
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        out->loop_seconds = ms * 1e-3;
        out->paths = (uint64_t)pixels * frame_count;
        out->path_vertices = vertices;
    } catch (const Failure &f) {
        rc = report(f, err, errlen);
    } catch (const std::exception &e) {
        if (err && errlen) snprintf(err, errlen, "host failure: %s", e.what());
        rc = F3D_STATUS_DEVICE;
    } catch (...) {
        rc = F3D_STATUS_DEVICE;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (void *p : owned) (void)device_free(p);
    return rc;
}
