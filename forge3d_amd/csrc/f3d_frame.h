// forge3d_amd/csrc/f3d_frame.h -- device code of the terrain path tracer's gfx950 kernels (k_frame, k_trace, k_merge,
// k_head, k_gbuffer, k_resolve, the table builders).  The launchers are in f3d_kernels.hip; development builds
// (tools/kf_resources.sh) instantiate single kernels from this header.
//
//
// Launch shape: one wave (64 lanes) per workgroup = one pixel tile (8x8 pixels like the
// reference's @workgroup_size(8,8,1), hybrid_terrain_traversal.wgsl:445, or fewer pixels with
// several sample lanes each -- frame_lanes below), so primary rays of a wave stay coherent.
// Workgroup b runs on XCD b % 8 (observed dispatch order, MI355X_MICROARCH.md); tile ROWS are
// dealt round-robin to the XCDs (tile_pixel).  No MFMA anywhere: there is no dense contraction
// on this path.
#pragma once
#include "f3d_launch.h"
#include "f3d_shade.h"
#include "f3d_lds.h"

namespace f3d {

constexpr int kNumXcd = 8;

// Pixel tile of a wave with S sample lanes per pixel: 64 / S pixels, TW x TH.
template <uint32_t S>
struct TileShape {
    static constexpr uint32_t kLogS = S == 1u ? 0u : (S == 2u ? 1u : (S == 4u ? 2u : 3u));
#if defined(F3D_TILE_LOGW_S4)  // A/B of the tile shape (profiles/README.md)
    static constexpr uint32_t kLogW = S <= 2u ? 3u : (S == 4u ? F3D_TILE_LOGW_S4 : 2u);
#else
    static constexpr uint32_t kLogW = S <= 2u ? 3u : 2u;      // 8, 8, 4, 4 pixels wide
#endif
    static constexpr uint32_t kLogH = 6u - kLogS - kLogW;     // 8, 4, 4, 2 pixels high
};

// Pixel of this lane: the launch covers the image rows [band_begin, band_end) of the strip, tiled from band_begin.
// `tile` returns the tile id of the wave (0xFFFFFFFF: the workgroup is padding).
// wg: the workgroup's position in the dispatch order of ONE frame (blockIdx.x, except in k_trace's batches: batch_slot).
template <uint32_t S = 1u>
__device__ __forceinline__ bool tile_pixel(const FrameParams &P, uint32_t &gx, uint32_t &gy, uint32_t &tile,
                                           const uint32_t *order = nullptr, uint32_t wg = 0xFFFFFFFFu) {
    if (wg == 0xFFFFFFFFu) wg = blockIdx.x;
    using Shape = TileShape<S>;
    constexpr uint32_t TW = 1u << Shape::kLogW, TH = 1u << Shape::kLogH;
    const uint32_t rows = P.band_end - P.band_begin;
    const uint32_t tiles_x = (P.cam.width + TW - 1u) >> Shape::kLogW, tiles_y = (rows + TH - 1u) >> Shape::kLogH;
    const uint32_t ntiles = tiles_x * tiles_y;
    tile = 0xFFFFFFFFu;
    // Workgroup b is observed to run on XCD b % 8.  tile_map picks how tiles are dealt to XCDs:
    //   1  tile id = workgroup id (consecutive tiles on different XCDs)
    //   2  tile ROWS dealt round-robin to XCDs (row r -> XCD r % 8) -- the default: the load
    //      balance of 1 with each XCD's L2 still seeing whole rows of coherent rays
    //   3  contiguous image bands per XCD (best L2 locality, but a sky band idles its XCD:
    //      measured 1.77x slower on the headline scene)
    // `order` (map 2 only): the same row -> XCD dealing, but each XCD starts its most expensive tiles first.
    uint32_t t;
    if (order) {
        t = order[wg];
    } else if (P.tile_map == 1u) {
        t = wg;
    } else if (P.tile_map == 2u) {
        const uint32_t xcd = wg % kNumXcd, i = wg / kNumXcd;
        const uint32_t rows_per_xcd = (tiles_y + kNumXcd - 1u) / kNumXcd;
        const uint32_t ty = (i / tiles_x) * kNumXcd + xcd;
        if (i >= rows_per_xcd * tiles_x || ty >= tiles_y) return false;
        t = ty * tiles_x + (i % tiles_x);
    } else {  // 3 (and anything else): contiguous bands
        const uint32_t per_xcd = (ntiles + kNumXcd - 1u) / kNumXcd;
        t = (wg % kNumXcd) * per_xcd + wg / kNumXcd;
    }
    if (t >= ntiles) return false;
    tile = t;
    const uint32_t pixel = threadIdx.x >> Shape::kLogS;  // the S sample lanes of a pixel are neighbours
    gx = (t % tiles_x) * TW + (pixel & (TW - 1u));
    gy = P.band_begin + (t / tiles_x) * TH + (pixel >> Shape::kLogW);
    return gx < P.cam.width && gy < P.band_end;
}
__device__ __forceinline__ bool tile_pixel(const FrameParams &P, uint32_t &gx, uint32_t &gy) {
    uint32_t tile;
    return tile_pixel<1u>(P, gx, gy, tile);
}
// The pixel of this lane again, from the wave's tile (wave-uniform: scalar registers) and the hardware lane id: the
// sample-lane kernels form (gx, gy) where they need them instead of keeping them -- and every address built from them --
// alive across the marches (lanes outside the image get coordinates outside it, as tile_pixel gives them).
template <uint32_t S>
__device__ __forceinline__ void lane_pixel(const FrameParams &P, uint32_t tile, uint32_t &gx, uint32_t &gy) {
    using Shape = TileShape<S>;
    constexpr uint32_t TW = 1u << Shape::kLogW, TH = 1u << Shape::kLogH;
    const uint32_t tiles_x = (P.cam.width + TW - 1u) >> Shape::kLogW;
    const uint32_t pixel = lane_now() >> Shape::kLogS;
    gx = (tile % tiles_x) * TW + (pixel & (TW - 1u));
    gy = P.band_begin + (tile / tiles_x) * TH + (pixel >> Shape::kLogW);
}

// ---- longest-first dispatch ------------------------------------------------------------------------
// Wave durations of the frame kernel are heavy-tailed (headline frame: median 20 us, 99th percentile 560 us,
// maximum 1.4 ms -- tools/wave_times.py): dispatched in image order, the long waves that happen to start late
// keep the kernel alive while the chip is empty (2.96 ms against 2.69 ms of perfectly packed wave time; a thin
// multi-GPU strip: 0.63 against 0.31 ms).  A tile costs about the same in consecutive frames, so every wave
// leaves its duration in tile_cost and k_tile_order sorts the tiles of each XCD (rows stay dealt round-robin to
// the XCDs) by descending cost class for the next frames: list scheduling, longest first.  One workgroup per
// XCD: LDS histogram over 64 logarithmic classes, prefix, scatter; the order inside a class is whatever the
// atomics give -- dispatch order never changes a result.
struct TileOrderParams {
    const uint32_t *cost;  // per tile, 100 MHz ticks
    uint32_t *order;       // per frame-kernel workgroup: tile id, 0xFFFFFFFF = padding
    uint32_t tiles_x, tiles_y;
};
__device__ __forceinline__ uint32_t cost_class(uint32_t ticks) {  // 0 = most expensive ... 63 = cheapest
    const uint32_t v = ticks | 1u, lg = 31u - (uint32_t)__clz((int)v);
    const uint32_t frac = lg >= 2u ? (v >> (lg - 2u)) & 3u : 0u;
    const uint32_t q = lg * 4u + frac;  // log2 with two fractional bits; 8 us ... 2.6 ms -> 38 ... 71
    const uint32_t c = q > 71u ? 71u : q;
    return c < 9u ? 63u : (71u - c > 63u ? 63u : 71u - c);
}
__global__ __launch_bounds__(1024) void k_tile_order(const TileOrderParams B) {
    __shared__ uint32_t hist[65];
    const uint32_t xcd = blockIdx.x, rows_per_xcd = (B.tiles_y + kNumXcd - 1u) / kNumXcd, slots = rows_per_xcd * B.tiles_x;
    if (threadIdx.x < 65u) hist[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) {
        const uint32_t ty = (i / B.tiles_x) * kNumXcd + xcd;
        const uint32_t cls = ty < B.tiles_y ? cost_class(B.cost[ty * B.tiles_x + i % B.tiles_x]) : 64u;
        atomicAdd(&hist[cls], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t run = 0u;
        for (uint32_t c = 0u; c < 65u; c++) {
            const uint32_t n = hist[c];
            hist[c] = run;
            run += n;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) {
        const uint32_t ty = (i / B.tiles_x) * kNumXcd + xcd;
        const bool real = ty < B.tiles_y;
        const uint32_t tile = real ? ty * B.tiles_x + i % B.tiles_x : 0xFFFFFFFFu;
        const uint32_t cls = real ? cost_class(B.cost[tile]) : 64u;
        B.order[atomicAdd(&hist[cls], 1u) * kNumXcd + xcd] = tile;
    }
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, kWave);
        v = v > o ? v : o;
    }
    return v;
}

__device__ __forceinline__ void publish_window_stats(const FrameParams &P, bool active, float m2) {
    // max over pixels of the Welford m2 (render_terrain.rs:1211-1226); m2 >= 0 so the bit
    // pattern orders like the value; non-finite values are flagged separately.
    const bool bad = active && !f_finite(m2);
    uint32_t bits = (active && !bad) ? f_bits(f_max(m2, 0.0f)) : 0u;
    bits = wave_max_u32(bits);
    const unsigned long long any_bad = __ballot(bad);
    if (lane_now() == 0u) {
        // most waves lose the race for the maximum: look before paying for the atomic
        if (bits > __hip_atomic_load(&P.stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&P.stats[0], bits);
        if (any_bad) atomicOr(&P.stats[1], 1u);
    }
}

// ---- sample-lane form of the frame ---------------------------------------------------------
// A lane of frame_pixel walks its pixel's spp samples one after the other, so a wave lasts
// spp x 3 traversals (~3 ms at 8 spp on the headline scene) however few waves there are: a strip of
// one eighth of a 1080p frame (4 080 waves for 6 144 wave slots) takes as long as half the frame,
// and a 512 x 512 image cannot fill the chip.  Here S neighbouring lanes trace S samples of the
// SAME pixel at once (tile = 64 / S pixels), which needs the two couplings between samples
// (f3d_shade.h) resolved:
//   (a) RNG stream: the state at the start of sample s depends on how many earlier samples hit.
//       The hit flags are PREDICTED (G-buffer centre ray: right for every pixel that is not on a
//       silhouette), every lane traces its primary ray from the predicted state, the group compares
//       flags (one ballot) and lanes whose start state was wrong trace again; sample 0 is always
//       right, so this settles in at most S rounds and in one for almost every pixel.
//   (b) radiance sum and candidate reservoir: every lane of the group replays all S contributions in
//       sample order (7 ds_bpermute per sample), so each holds the exact running values.
// Results are bit-identical to frame_pixel.  The frame head (spatial reuse of the previous frame,
// ~1 200 instructions a pixel) would run redundantly on all S lanes, so it runs in its own
// pixel-parallel launch (k_head) and leaves an 8-byte record per pixel; the short tail runs on
// sample lane 0.
// (`valid` masks the lanes outside the image: every lane of the wave runs the function to its end.)
template <uint32_t S, class Pending>
__device__ __forceinline__ float frame_lanes(const FrameParams &P, uint32_t tile, bool valid, Pending &pend) {
    constexpr uint32_t kGroup = (1u << S) - 1u;
    // Register budget (round 4).  At 80 VGPRs the marches leave room for ~25 values of the frame; what does not fit lives
    // in scratch, and 160 bytes of scratch per lane x 6 144 waves a CU did not fit the L2 (1.4 GB written per launch).
    // So nothing that can be formed again is kept: the pixel (lane_pixel), the lane's position in its group (lane_now),
    // every address; and what is the lane's own but only read between the marches is parked in its LDS column: the
    // accumulators of the samples so far (6 words) and the frame head's record (2 words).
    uint32_t *park = pend.col + kParkRow * kWave;
    uint32_t stream = 0u;  // state at the start of the current round
    {
        uint32_t gx, gy;
        lane_pixel<S>(P, tile, gx, gy);
        uint2 rec = uint2{f_bits(1.0f), 0u};  // no reuse, centre ray missed, no usable history
        if (valid) {
            rec = P.head[(size_t)(gy - P.row_begin) * P.cam.width + gx];  // k_head
            stream = P.cam.seed_hi ^ (gx * 1664525u) ^ (gy * 1013904223u) ^ (P.frame_index * 92837111u) ^ P.cam.seed_lo;
        }
        park[kParkHead * kWave] = rec.x;
        park[(kParkHead + 1) * kWave] = rec.y;
    }
#pragma unroll
    for (int w = 0; w < kParkHead; w++) park[w * kWave] = 0u;  // radiance = 0, empty candidate reservoir
    for (uint32_t s0 = 0u; s0 < P.spp; s0 += S) {  // wave-uniform
        const uint32_t n_act = P.spp - s0 < S ? P.spp - s0 : S;
        const bool act = valid && (lane_now() & (S - 1u)) < n_act;
        uint32_t pred = (park[(kParkHead + 1) * kWave] & kHeadCentreHit) != 0u ? kGroup : 0u;  // predicted hit flags of this round's samples
        uint32_t traced = 0xFFFFFFFFu;               // draws in front of my sample when I last traced it
        PrimaryHit ph;
        ph.hit.kind = 0u;
        ph.rng = 0u;
        for (;;) {
            const uint32_t j = lane_now() & (S - 1u);
            const uint32_t draws = 2u * j + 2u * (uint32_t)__popc(pred & ((1u << j) - 1u));
            const bool need = act && draws != traced;
            if (__ballot(need) == 0ull) break;
            if (need) {
                uint32_t st = stream, gx, gy;
                rng_skip(st, draws);
                lane_pixel<S>(P, tile, gx, gy);
                ph = sample_primary(P, gx, gy, st, pend);
                traced = draws;
            }
            pred = (uint32_t)(__ballot(act && ph.hit.kind != 0u) >> (lane_now() & ~(S - 1u))) & kGroup;
        }
        SampleOut o;
        o.a = V3{0.0f, 0.0f, 0.0f};
        o.b = V3{0.0f, 0.0f, 0.0f};
        o.target_pdf = 0.0f;
        IblRay q;
        q.valid = false;
        q.o = q.d = q.b0 = V3{0.0f, 0.0f, 0.0f};
        q.key = 2.0f;
        q.t_stop = 3.0e38f;
        if (act) {
            uint32_t rng = ph.rng;
            const bool prev_valid = (park[(kParkHead + 1) * kWave] & kHeadPrevValid) != 0u;
            q = sample_shade_sun(P, prev_valid, [park]() { return f_from_bits(park[kParkHead * kWave]); }, ph, rng, o, pend);
        }
        // (Sorting the IBL rays of a 4-wave workgroup by cos(normal, ray) -- a good predictor of the march
        // length, 1.6x fewer IBL wave iterations in the step-log model -- was built and measured: bit-identical,
        // but 0.81x: the waves that finish early wait at the workgroup barrier and the occupancy the kernel
        // lives on is gone.  profiles/README.md)
        if (q.valid) o.b = q.b0 * (ibl_occluded(P, q.o, q.d, pend, q.t_stop) ? 0.0f : 1.0f);
        V3 radiance = V3{f_from_bits(park[0]), f_from_bits(park[kWave]), f_from_bits(park[2 * kWave])};
        Reservoir cand;
        cand.w_sum = f_from_bits(park[3 * kWave]);
        cand.m = park[4 * kWave] & ~kLightTypeBit;
        cand.target_pdf = f_from_bits(park[5 * kWave]);
        cand.directional = (park[4 * kWave] & kLightTypeBit) != 0u;
        cand.weight = 0.0f;
        const int base = (int)(lane_now() & ~(S - 1u));
#pragma unroll
        for (uint32_t k = 0u; k < S; k++) {
            const int src = base + (int)k;
            const V3 a = V3{__shfl(o.a.x, src, kWave), __shfl(o.a.y, src, kWave), __shfl(o.a.z, src, kWave)};
            const V3 b = V3{__shfl(o.b.x, src, kWave), __shfl(o.b.y, src, kWave), __shfl(o.b.z, src, kWave)};
            const float tp = __shfl(o.target_pdf, src, kWave);
            if (k < n_act) accumulate_sample(cand, radiance, a, b, tp);
        }
        park[0] = f_bits(radiance.x);
        park[kWave] = f_bits(radiance.y);
        park[2 * kWave] = f_bits(radiance.z);
        park[3 * kWave] = f_bits(cand.w_sum);
        park[4 * kWave] = cand.m | (cand.directional ? kLightTypeBit : 0u);
        park[5 * kWave] = f_bits(cand.target_pdf);
        rng_skip(stream, 2u * n_act + 2u * (uint32_t)__popc(pred));
    }
    if (!valid || (lane_now() & (S - 1u)) != 0u) return 0.0f;
    Reservoir cand;
    cand.w_sum = f_from_bits(park[3 * kWave]);
    cand.m = park[4 * kWave] & ~kLightTypeBit;
    cand.target_pdf = f_from_bits(park[5 * kWave]);
    cand.directional = (park[4 * kWave] & kLightTypeBit) != 0u;
    cand.weight = 0.0f;
    uint32_t gx, gy;
    lane_pixel<S>(P, tile, gx, gy);
    return frame_tail(P, gx, gy, cand, V3{f_from_bits(park[0]), f_from_bits(park[kWave]), f_from_bits(park[2 * kWave])});
}

// ---- frames in flight ------------------------------------------------------------------------------
// Everything a frame TRACES is independent of the frames before it: the RNG stream is keyed by (pixel, frame), the
// rays by the stream and the scene.  What depends on frame f - 1 is cheap and ordered: the frame head (spatial reuse
// of the previous reservoirs -> the weight `reuse_w` that multiplies the sun term, and which of the two equal-valued
// sun directions is read), the order-sensitive sums over the samples, the temporal merge and the accumulation.
// k_frame runs both per frame, so a render is a chain of launches each as long as its longest wave (a multi-GPU
// strip: 0.53 ms against 0.32 ms of packed work).  Here the two are separate kernels:
//   k_trace   one launch for a BATCH of frames (grid.y = frame): per (frame, pixel, sample) the primary, sun and IBL
//             rays exactly as frame_lanes traces them -- with reuse_w = 1 (x * 1.0f == x) -- and one 32-byte record
//             {sun term or miss radiance, target pdf; IBL term, hit flag}: W x H x frames independent lanes' worth
//             of work, no chain, no tail per frame;
//   k_merge   per frame, in order, one lane per pixel: frame_head, the samples' records through accumulate_sample
//             with the sun term multiplied by the real reuse_w (the operation k_frame does at that point), frame_tail.
// The records wait in HBM: 32 B x spp x pixels per frame in flight (0.53 GB at 1080p, 8 spp) -- room the 288 GB have.
// One thing a frame traces DOES look at the frame before: the head picks the sun direction `wi` or normalize(wi)
// by whether the merged reservoir is valid, and the two may differ in the last bit.  Validity is persistent (after
// the first frame it changes for next to no pixel), so k_trace PREDICTS it -- frame 0: invalid (exact); later
// frames: what k_merge last saw for the pixel (before that: the centre ray faces the sun) -- and notes the prediction
// in the record; k_merge compares with the real head and, for the rare pixel-frame that was mispredicted, traces
// the pixel's primary and sun rays again with the right direction in k_fix (the IBL terms do not depend on it).  With equal
// bits (same_sun) nothing is predicted.  Results are those of k_frame bit for bit either way.
template <uint32_t S, class Pending>
__device__ __forceinline__ void trace_lanes(const FrameParams &P, uint32_t frame, uint32_t tile, bool valid, Pending &pend) {
    constexpr uint32_t kGroup = (1u << S) - 1u;
    const size_t pixels = (size_t)(P.row_end - P.row_begin) * P.cam.width;
    // (registers: as frame_lanes -- the pixel and every address are formed where they are used, the head's flags are parked)
    uint32_t *park = pend.col + kParkRow * kWave;
    uint32_t stream = 0u;
    {
        uint32_t gx, gy, flags = 0u;
        lane_pixel<S>(P, tile, gx, gy);
        if (valid) {
            const size_t lp = (size_t)(gy - P.row_begin) * P.cam.width + gx;
            if (P.gbuffer_n[lp].w != 0.0f) flags |= kHeadCentreHit;  // prediction of the hit flags only
            // which sun direction the head will read: predicted (see above); immaterial when they are the same bits
            if (frame > 0u && (P.head[lp].y & kHeadPrevValid) != 0u) flags |= kHeadPrevValid;
            stream = P.cam.seed_hi ^ (gx * 1664525u) ^ (gy * 1013904223u) ^ (frame * 92837111u) ^ P.cam.seed_lo;
        }
        park[(kParkHead + 1) * kWave] = flags;
    }
    for (uint32_t s0 = 0u; s0 < P.spp; s0 += S) {  // wave-uniform
        const uint32_t n_act = P.spp - s0 < S ? P.spp - s0 : S;
        const bool act = valid && (lane_now() & (S - 1u)) < n_act;
        uint32_t pred = (park[(kParkHead + 1) * kWave] & kHeadCentreHit) != 0u ? kGroup : 0u;
        uint32_t traced = 0xFFFFFFFFu;
        PrimaryHit ph;
        ph.hit.kind = 0u;
        ph.rng = 0u;
        for (;;) {
            const uint32_t j = lane_now() & (S - 1u);
            const uint32_t draws = 2u * j + 2u * (uint32_t)__popc(pred & ((1u << j) - 1u));
            const bool need = act && draws != traced;
            if (__ballot(need) == 0ull) break;
            if (need) {
                uint32_t st = stream, gx, gy;
                rng_skip(st, draws);
                lane_pixel<S>(P, tile, gx, gy);
                ph = sample_primary(P, gx, gy, st, pend);
                traced = draws;
            }
            pred = (uint32_t)(__ballot(act && ph.hit.kind != 0u) >> (lane_now() & ~(S - 1u))) & kGroup;
        }
        SampleOut o;
        o.a = V3{0.0f, 0.0f, 0.0f};
        o.b = V3{0.0f, 0.0f, 0.0f};
        o.target_pdf = 0.0f;
        IblRay q;
        q.valid = false;
        q.o = q.d = q.b0 = V3{0.0f, 0.0f, 0.0f};
        q.key = 2.0f;
        q.t_stop = 3.0e38f;
        if (act) {
            uint32_t rng = ph.rng;
            const bool prev_valid = (park[(kParkHead + 1) * kWave] & kHeadPrevValid) != 0u;
            q = sample_shade_sun(P, prev_valid, []() { return 1.0f; }, ph, rng, o, pend);  // the reuse weight is applied by k_merge
        }
        if (q.valid) o.b = q.b0 * (ibl_occluded(P, q.o, q.d, pend, q.t_stop) ? 0.0f : 1.0f);
        if (act) {
            uint32_t gx, gy;
            lane_pixel<S>(P, tile, gx, gy);
            const size_t lp = (size_t)(gy - P.row_begin) * P.cam.width + gx;
            const bool prev_valid = (park[(kParkHead + 1) * kWave] & kHeadPrevValid) != 0u;
            float4 *rec = P.trace + 2u * (((size_t)(frame - P.trace_first) * P.spp + s0 + (lane_now() & (S - 1u))) * pixels + lp);
            rec[0] = float4{o.a.x, o.a.y, o.a.z, o.target_pdf};
            rec[1] = float4{o.b.x, o.b.y, o.b.z, trace_code(ph.hit.kind != 0u, prev_valid)};
        }
        rng_skip(stream, 2u * n_act + 2u * (uint32_t)__popc(pred));
    }
}

// ---- wavefront form of a trace batch ---------------------------------------------------------------------------------
// k_trace spends 70 % of its time in the two occlusion phases, at a third of the primary phase's lane utilisation: every
// wave waits for the longest sun ray and then for the longest IBL ray of its 64 samples (timing builds: 0.72 ms for
// primaries + shading, 0.76 ms sun rays, 0.97 ms IBL rays of a 2.49 ms frame).  The wavefront form takes the occlusion
// rays out of the pixel's wave: k_wf_primary traces the primaries, evaluates the shading with both rays assumed
// unblocked, writes the records and APPENDS the rays to queues in HBM (ballot + one atomic per wave); k_wf_occl runs
// persistent waves that stream a queue through march_stream (f3d_march.h) -- a lane takes the next ray when its own is
// done -- and zero the term of a record whose ray is blocked.  Records and merges are those of the frames-in-flight
// pipeline, bit for bit: y * 0.0f where the fused kernel multiplies by vis = 0, untouched where it multiplies by 1.
template <uint32_t S>
__device__ __forceinline__ void wf_primary_lanes(const FrameParams &P, uint32_t frame, uint32_t tile, uint32_t gx, uint32_t gy, bool valid,
                                                 LdsPending &pend) {
    const uint32_t lane = threadIdx.x & (kWave - 1u), j = lane & (S - 1u), base = lane & ~(S - 1u);
    constexpr uint32_t kGroup = (1u << S) - 1u;
    const size_t pixels = (size_t)(P.row_end - P.row_begin) * P.cam.width;
    const size_t lp = valid ? (size_t)(gy - P.row_begin) * P.cam.width + gx : 0u;
    FrameHead h;  // as trace_lanes: predicted hit flags and sun direction, reuse weight applied by k_merge
    h.centre_hit = valid && P.gbuffer_n[lp].w != 0.0f;
    h.prev_valid = valid && frame > 0u && (P.head[lp].y & kHeadPrevValid) != 0u;
    h.reuse_w = 1.0f;
    h.rng = P.cam.seed_hi ^ (gx * 1664525u) ^ (gy * 1013904223u) ^ (frame * 92837111u) ^ P.cam.seed_lo;
    uint32_t stream = h.rng;
    const size_t frame_base = (size_t)(frame - P.trace_first) * P.spp * pixels;
    const uint32_t rounds = (P.spp + S - 1u) / S;
    for (uint32_t s0 = 0u; s0 < P.spp; s0 += S) {  // wave-uniform
        const uint32_t n_act = P.spp - s0 < S ? P.spp - s0 : S;
        const bool act = valid && j < n_act;
        uint32_t pred = h.centre_hit ? kGroup : 0u;
        uint32_t traced = 0xFFFFFFFFu;
        PrimaryHit ph;
        ph.hit.kind = 0u;
        ph.rng = 0u;
        ph.sun_tmax = 1e30f;
        for (;;) {
            const uint32_t draws = 2u * j + 2u * (uint32_t)__popc(pred & ((1u << j) - 1u));
            const bool need = act && draws != traced;
            if (__ballot(need) == 0ull) break;
            if (need) {
                uint32_t st = stream;
                rng_skip(st, draws);
                ph = sample_primary(P, gx, gy, st, pend);
                traced = draws;
            }
            pred = (uint32_t)(__ballot(act && ph.hit.kind != 0u) >> base) & kGroup;
        }
        SampleOut o;
        o.a = o.b = V3{0.0f, 0.0f, 0.0f};
        o.target_pdf = 0.0f;
        bool sun_ray = false, sun_back = false, ibl_ray = false;
        ShadeSetup su;
        su.q.o = su.q.d = V3{0.0f, 0.0f, 0.0f};
        su.q.t_stop = 3.0e38f;
        const uint32_t tag = (uint32_t)(frame_base + (size_t)(s0 + j) * pixels + lp);
        if (act) {
            uint32_t rng = ph.rng;
            su = sample_shade_setup(P, h.prev_valid, ph, rng, o);
            if (su.need_sun) o.a = su.y;  // (y * 1.0f) * 1.0f: k_wf_occl makes it (y * 0.0f) * 1.0f if the ray is blocked
            sun_ray = su.need_sun && P.light.shadows_enabled != 0u;
            sun_back = h.prev_valid && P.same_sun == 0u;  // the ray runs along light.wi_reuse (when that differs from light.wi at all)
            ibl_ray = su.q.valid;
            if (ibl_ray) o.b = su.q.b0;
            float4 *rec = P.trace + 2u * (size_t)tag;
            rec[0] = float4{o.a.x, o.a.y, o.a.z, o.target_pdf};
            rec[1] = float4{o.b.x, o.b.y, o.b.z, trace_code(ph.hit.kind != 0u, h.prev_valid)};
        }
        // this wave-round's region of the queues: compacted by ballots, counts in one word
        const size_t region = ((size_t)(frame - P.trace_first) * (P.wf.regions_per_frame / rounds) + tile) * rounds + s0 / S;
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long m_front = __ballot(sun_ray && !sun_back), m_back = __ballot(sun_ray && sun_back), m_ibl = __ballot(ibl_ray);
        if (sun_ray) {
            const size_t slot = region * kWfRegion + (sun_back ? kWfRegion - 1u - (uint32_t)__popcll(m_back & below) : (uint32_t)__popcll(m_front & below));
            P.wf.sun_o[slot] = float4{su.q.o.x, su.q.o.y, su.q.o.z, f_from_bits(tag)};
            P.wf.sun_stop[slot] = ph.sun_tmax;
        }
        if (ibl_ray) {
            const size_t slot = region * kWfRegion + (uint32_t)__popcll(m_ibl & below);
            P.wf.ibl_o[slot] = float4{su.q.o.x, su.q.o.y, su.q.o.z, f_from_bits(tag)};
            P.wf.ibl_d[slot] = float4{su.q.d.x, su.q.d.y, su.q.d.z, su.q.t_stop};
        }
        if (lane == 0u) P.wf.counts[region] = (uint32_t)__popcll(m_front) | ((uint32_t)__popcll(m_back) << 8) | ((uint32_t)__popcll(m_ibl) << 16);
        rng_skip(stream, 2u * n_act + 2u * (uint32_t)__popc(pred));
    }
}

template <int MIN_WAVES, uint32_t S>
__global__ __launch_bounds__(kWave, MIN_WAVES) void k_wf_primary(const FrameParams P) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLdsWords];
    const unsigned long long t_start = wall_clock64();
    LdsPending pend = make_pending(lds, P.terrain);
    uint32_t gx = 0u, gy = 0u, tile;
    const bool active = tile_pixel<S>(P, gx, gy, tile, P.tile_order);
    if (tile == 0xFFFFFFFFu) return;  // a padding workgroup: no tile, no region
    wf_primary_lanes<S>(P, P.frame_index + blockIdx.y, tile, gx, gy, active, pend);
    if (threadIdx.x == 0u && blockIdx.y == 0u && P.tile_cost && tile != 0xFFFFFFFFu)
        P.tile_cost[tile] = (uint32_t)(wall_clock64() - t_start);
}

// One queue of occlusion rays through persistent waves (SUN: wave-uniform direction, curvature policy on).
struct WfOcclParams {
    TerrainDev terrain;
    float4 *trace;
    const float4 *ray_o, *ray_d;  // ray_d: IBL rays only
    const float *ray_stop;        // sun rays only
    const uint32_t *counts;       // per region (WfQueues::counts)
    uint32_t *cursor;             // chunks handed out beyond each wave's first one
    uint32_t regions;             // of this batch
    uint32_t kind;                // 0 sun rays from the front of a region, 1 sun rays from its back, 2 IBL rays
    V3 dir;                       // sun rays
    uint32_t quorum;
};
constexpr uint32_t kWfChunk = 8u;  // regions a wave takes at a time: its first chunk is its own index, the next ones come from the cursor
template <bool SUN>
struct WfSource {
    const WfOcclParams &W;
    uint32_t chunk, chunks;  // current chunk (>= chunks: exhausted)
    uint32_t k, used, have_n;  // region of the chunk, rays of it handed out, rays it holds
    uint32_t my_count;       // lane l < kWfChunk: the count of region l of the current chunk
    bool first;
    __device__ __forceinline__ void load_chunk(uint32_t lane) {
        const uint32_t region = chunk * kWfChunk + lane;
        const uint32_t word = (lane < kWfChunk && region < W.regions) ? W.counts[region] : 0u;
        my_count = (word >> (8u * W.kind)) & 0xFFu;
        k = 0u;
        used = 0u;
        have_n = (uint32_t)__shfl((int)my_count, 0, kWave);
    }
    __device__ __forceinline__ bool next_chunk(uint32_t lane) {
        uint32_t id = 0u;
        if (lane == 0u) id = gridDim.x + atomicAdd(W.cursor, 1u);
        chunk = (uint32_t)__shfl((int)id, 0, kWave);
        if (chunk >= chunks) return false;
        load_chunk(lane);
        return true;
    }
    __device__ __forceinline__ bool refill(bool &have, RayCtx &r, float &t_stop, uint32_t &tag, LdsPending &ctx) {
        const uint32_t lane = ctx.lane();
        if (first) {  // the wave's own chunk: no atomic for it (8 192 waves asking one counter at once: 60 ns each)
            first = false;
            chunk = blockIdx.x;
            if (chunk >= chunks) return false;
            load_chunk(lane);
        }
        const unsigned long long idle = __ballot(!have);
        const uint32_t n_idle = (uint32_t)__popcll(idle), rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
        const bool was_idle = !have;
        uint32_t given = 0u;
        while (given < n_idle) {  // wave-uniform
            if (used == have_n) {
                if (++k == kWfChunk) {
                    if (!next_chunk(lane)) return false;
                } else {
                    used = 0u;
                    have_n = (uint32_t)__shfl((int)my_count, (int)k, kWave);
                }
                continue;
            }
            const uint32_t give = have_n - used < n_idle - given ? have_n - used : n_idle - given;
            if (was_idle && rank >= given && rank < given + give) {
                const uint32_t j = used + (rank - given);
                const size_t slot = (size_t)(chunk * kWfChunk + k) * kWfRegion + (W.kind == 1u ? kWfRegion - 1u - j : j);
                const float4 o = W.ray_o[slot];
                V3 d = W.dir;
                if (SUN) {
                    t_stop = W.ray_stop[slot];
                } else {
                    const float4 dd = W.ray_d[slot];
                    d = V3{dd.x, dd.y, dd.z};
                    t_stop = dd.w;
                }
                tag = f_bits(o.w);
                r = make_ray(W.terrain, V3{o.x, o.y, o.z}, 1e-3f, d, 1e30f, SUN);  // occluded(): tmin 1e-3, max distance 1e30
                have = true;
            }
            used += give;
            given += give;
        }
        return true;
    }
    __device__ __forceinline__ void verdict(uint32_t tag, bool blocked) const {
        if (!blocked) return;
        float4 *rec = W.trace + 2u * (size_t)tag + (SUN ? 0u : 1u);
        float4 v = *rec;
        // the fused kernel: a = (y * vis) * reuse_w with vis = 0 (reuse_w = 1 in a trace batch); b = b0 * 0
        if (SUN) *rec = float4{(v.x * 0.0f) * 1.0f, (v.y * 0.0f) * 1.0f, (v.z * 0.0f) * 1.0f, v.w};
        else *rec = float4{v.x * 0.0f, v.y * 0.0f, v.z * 0.0f, v.w};
    }
};
template <bool SUN, int MIN_WAVES>
__global__ __launch_bounds__(kWave, MIN_WAVES) void k_wf_occl(const WfOcclParams W) {
    constexpr uint32_t kRows = kFifoWords * kLeafFifoRows;  // the leaf FIFO; no park rows, no verdict board
    __shared__ __attribute__((aligned(16))) uint32_t lds[kRows * kWave + 4 * kMaxLevels];
    LdsPending pend = make_pending(lds, W.terrain, kRows);
    WfSource<SUN> src{W, 0u, (W.regions + kWfChunk - 1u) / kWfChunk, 0u, 0u, 0u, 0u, true};
    march_stream<SUN>(W.terrain, src, pend, W.quorum ? W.quorum : (uint32_t)F3D_STREAM_QUORUM);
}

// A batch of frames is one launch (grid.y = frames).  The hardware starts workgroups with blockIdx.x running fastest, so
// taken literally frame 0's tiles would all start before frame 1's -- longest first WITHIN a frame, but the long tiles of
// the batch's last frame would start last and the launch would end in their tail (round 3).  The launch's linear order is
// therefore re-read (round 4): groups of 8 consecutive workgroups (one per XCD, which keeps a tile's row on its XCD) cycle
// through ALL frames of the batch before the next 8 slots of the per-frame order are touched -- longest first over the
// whole batch.  Which workgroup traces a (frame, tile) never changes what it computes.
__device__ __forceinline__ void batch_slot(uint32_t &frame_in_batch, uint32_t &slot) {
    const uint32_t linear = blockIdx.y * gridDim.x + blockIdx.x, group = linear / kNumXcd, xcd = linear % kNumXcd;  // (gridDim.x is a multiple of 8: frame_grid)
    frame_in_batch = group % gridDim.y;
    slot = (group / gridDim.y) * kNumXcd + xcd;
}
template <int MIN_WAVES, uint32_t S, bool MESH = false>
__global__ __launch_bounds__(kWave, MIN_WAVES) void k_trace(const FrameParams P) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLdsWords];
    const unsigned long long t_start = wall_clock64();
    typename PendingFor<MESH>::type pend{make_pending(lds, P.terrain)};
    uint32_t gx = 0u, gy = 0u, tile, frame_in_batch, slot;
    batch_slot(frame_in_batch, slot);
    const bool active = tile_pixel<S>(P, gx, gy, tile, P.tile_order, slot);
    trace_lanes<S>(P, P.frame_index + frame_in_batch, tile, active, pend);
    if (lane_now() == 0u && frame_in_batch == 0u && P.tile_cost && tile != 0xFFFFFFFFu)
        P.tile_cost[tile] = (uint32_t)(wall_clock64() - t_start);
}

// The ordered half of a frame (see above): one lane per pixel, 8x8 tiles.  A pixel-frame whose sun direction was
// mispredicted is traced again HERE, by its own lane (fix_pixel: head again -- idempotent --, the pixel's primary and sun
// rays with the direction the real head reads, the IBL terms from the records, the tail).  Round 3 listed such pixels and
// re-traced them 64 to a wave in a second kernel (k_fix): a launch per frame in the ordered chain (11 us of a 0.30 ms
// strip-frame, profiles/r04_strip_rocprofv3_summary.txt) for a list that is empty in all but a handful of frames.
__global__ __launch_bounds__(kWave) void k_merge(const FrameParams P) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLdsWords];
    uint32_t gx = 0u, gy = 0u;
    const bool active = tile_pixel(P, gx, gy);
    float m2 = 0.0f;
    const size_t pixels = (size_t)(P.row_end - P.row_begin) * P.cam.width;
    const size_t lp = active ? (size_t)(gy - P.row_begin) * P.cam.width + gx : 0u;
    const float4 *rec = P.trace + 2u * ((size_t)(P.frame_index - P.trace_first) * P.spp * pixels + lp);
    bool redo = false;
    if (active) {
        const FrameHead h = frame_head<true>(P, gx, gy);
        // the prediction for the frames traced next (frame 0 says nothing: there every head is invalid by definition)
        if (P.same_sun == 0u && P.frame_index > 0u) P.head[lp].y = h.prev_valid ? kHeadPrevValid : 0u;
        if (P.spp == 8u) {  // (wave-uniform) every record load of the pixel-frame in flight at once: merge_pixel_n
            m2 = merge_pixel_n<8u>(P, gx, gy, h, rec, pixels, P.same_sun == 0u, redo);
        } else {
            if (P.same_sun == 0u) redo = merge_mispredicted(P, h, rec, pixels);
            if (!redo) m2 = merge_pixel(P, gx, gy, h, rec, pixels);
        }
    }
    if (__ballot(redo) != 0ull) {  // rare (3 pixel-frames in the first 18 frames of the headline scene, none later)
        LdsPending pend = make_pending(lds, P.terrain);
        if (redo) {
            m2 = fix_pixel(P, gx, gy, rec, pixels, pend);
            atomicAdd(&P.fix_count[2], 1u);  // diagnostics: f3d_session_retraced_pixels
        }
    }
    if (P.collect_stats != 0u) publish_window_stats(P, active, m2);
}

// First prediction of "the merged reservoir is valid" for the frames traced before any merge: the centre ray hit a
// surface that faces the sun.
__global__ __launch_bounds__(kWave) void k_trace_init(const FrameParams P) {
    uint32_t gx, gy;
    if (!tile_pixel(P, gx, gy)) return;
    const size_t lp = (size_t)(gy - P.row_begin) * P.cam.width + gx;
    const float4 g = P.gbuffer_n[lp];
    P.head[lp] = uint2{0u, (g.w != 0.0f && dot(V3{g.x, g.y, g.z}, P.light.wi) > 0.0f) ? kHeadPrevValid : 0u};
}

// Does no reservoir the spatial pass of this 8x8 tile could read hold a sample (m == 0)?  The pass draws its neighbours
// from [gx - 3, gx + 4] x [gy - 3, gy + 4], clamped to the image -- PLUS four, not three: the offset is floor(u * 7) - 3
// and u is exactly 1.0 for the top 128 values of the generator (f3d_math.h rng_next; the same fact kHaloRows = 4 rests
// on, tests/test_halo_reach.py; round 4 looked three pixels out here and was wrong for 2^-25 of the draws).  At most
// 15 x 15 pixels for the tile, read here as one word each (the wave's lanes over a 16-wide raster of the box) and voted
// on.  Wave-uniform.
__device__ __forceinline__ bool head_neighbourhood_empty(const FrameParams &P, uint32_t gx, uint32_t gy) {
    const uint32_t lane = lane_now();
    const uint32_t x0 = gx - (lane & 7u), y0 = gy - (lane >> 3);  // the tile's first pixel
    const uint32_t x_last = (x0 + 7u < P.cam.width ? x0 + 7u : P.cam.width - 1u), y_last = (y0 + 7u < P.band_end ? y0 + 7u : P.band_end - 1u);
    const uint32_t xa = x0 >= kSpatialReachLo ? x0 - kSpatialReachLo : 0u, ya = y0 >= kSpatialReachLo ? y0 - kSpatialReachLo : 0u;
    static_assert(8u + kSpatialReachLo + kSpatialReachHi <= 16u, "the 16 x 16 raster below covers the tile's reach");
    const uint32_t xb = x_last + kSpatialReachHi < P.cam.width ? x_last + kSpatialReachHi : P.cam.width - 1u;
    const uint32_t yb = y_last + kSpatialReachHi < P.cam.height ? y_last + kSpatialReachHi : P.cam.height - 1u;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(P.res_in);
    uint32_t some = 0u;
#pragma unroll
    for (uint32_t i = 0u; i < 4u; i++) {
        const uint32_t k = lane + 64u * i, bx = xa + (k & 15u), by = ya + (k >> 4);
        if (bx <= xb && by <= yb) some |= words[4u * reservoir_index(P, bx, by) + 1u] & ~kLightTypeBit;
    }
    return __ballot(some != 0u) == 0ull;
}

// Frame head of the sample-lane form: one lane per pixel (8x8 tiles).
// A tile whose neighbourhood holds no sample -- the sky, 60 % of the headline frame -- skips the spatial pass: with m == 0
// everywhere no candidate is considered and no number drawn, the pass returns the pixel's own light type and target pdf
// around w_sum = 0, m = 0, weight = 0, and the head is "no usable history" (spatial_reuse / frame_head, f3d_shade.h, for
// that input; k_head 53 -> see profiles/README.md).
__global__ __launch_bounds__(kWave) void k_head(const FrameParams P) {
    uint32_t gx, gy, tile;
    const bool active = tile_pixel<1u>(P, gx, gy, tile);
    if (tile == 0xFFFFFFFFu) return;
#if !defined(F3D_NO_EMPTY_HEAD)  // A/B + test-of-the-tests switch
    if (P.frame_index > 0u && head_neighbourhood_empty(P, gx, gy)) {
        if (active) {
            const size_t ri = reservoir_index(P, gx, gy), lp = (size_t)(gy - P.row_begin) * P.cam.width + gx;
            const PackedReservoir self = P.res_in[ri];
            P.res_out[ri] = PackedReservoir{0.0f, self.m_lt, 0.0f, self.target_pdf};
            FrameHead h;
            h.centre_hit = P.gbuffer_n[lp].w != 0.0f;
            h.prev_valid = false;
            h.reuse_w = 1.0f;
            h.rng = 0u;  // (not part of the record)
            P.head[lp] = pack_head(h);
        }
        return;
    }
#endif
    if (active) P.head[(size_t)(gy - P.row_begin) * P.cam.width + gx] = pack_head(frame_head<true>(P, gx, gy));
}

// VARIANT is reserved for A/B builds (0 = the shipped kernel).
// MIN_WAVES: waves per SIMD the register allocator must leave room for (1 = unconstrained).
// S: sample lanes per pixel (1 = frame_pixel; 2, 4, 8 = frame_lanes).
// MESH: the scene may hold a mesh (FrameParams::mesh.traversal_mode == 0); terrain-only renders run the instantiation
// without the mesh walk.
template <int VARIANT, int MIN_WAVES = 1, uint32_t S = 1u, bool MESH = false>
__global__ __launch_bounds__(kWave, MIN_WAVES) void k_frame(const FrameParams P) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLdsWords];
    const unsigned long long t_start = wall_clock64();  // 100 MHz: the wave's cost for the next tile ordering
    typename PendingFor<MESH>::type pend{make_pending(lds, P.terrain)};
    uint32_t gx = 0u, gy = 0u, tile;
    const bool active = tile_pixel<S>(P, gx, gy, tile, P.tile_order);
    float m2 = 0.0f;
    if constexpr (S == 1u) {
        if (active) m2 = frame_pixel(P, gx, gy, pend);
        if (P.collect_stats != 0u) publish_window_stats(P, active, m2);
    } else {
        m2 = frame_lanes<S>(P, tile, active, pend);
        if (P.collect_stats != 0u) publish_window_stats(P, active && (lane_now() & (S - 1u)) == 0u, m2);
    }
    if (lane_now() == 0u) {  // (one wave per workgroup)
        const unsigned long long t_end = wall_clock64();
        if (P.tile_cost && tile != 0xFFFFFFFFu) P.tile_cost[tile] = (uint32_t)(t_end - t_start);
#if defined(F3D_WAVE_TIMES)  // diagnostics: when did this wave run? (tools/wave_times.py)
        if (P.wave_times) {
            P.wave_times[2u * blockIdx.x] = t_start;
            P.wave_times[2u * blockIdx.x + 1u] = t_end;
        }
#endif
    }
}

__global__ __launch_bounds__(kWave) void k_gbuffer(const FrameParams P, float4 *gbuffer_n, float *depth) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLdsWords];
    LdsPending pend = make_pending(lds, P.terrain);
    uint32_t gx, gy;
    if (tile_pixel(P, gx, gy)) gbuffer_pixel(P, gx, gy, gbuffer_n, depth, pend);
}

__global__ __launch_bounds__(kWave) void k_resolve(const ResolveParams R) {
    uint32_t gx, gy;
    const bool active = tile_pixel(R.frame, gx, gy);
    uint32_t flags = 0u;
    if (active) flags = resolve_pixel(R.frame, R.frames, gx, gy, R.rgba, R.albedo, R.normal, &R.aether, R.depth);
    const unsigned long long valid = __ballot((flags & 1u) != 0u), bad = __ballot((flags & 2u) != 0u);
    if (threadIdx.x == 0) {
        if (valid) atomicOr(&R.frame.stats[2], 1u);
        if (bad) atomicOr(&R.frame.stats[3], 1u);
    }
}

__global__ __launch_bounds__(kWave) void k_ray_batch(const RayBatchParams B) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLdsWords];
    LdsPending pend = make_pending(lds, B.terrain);
    const uint32_t i = blockIdx.x * kWave + threadIdx.x;
    if (i >= B.n) return;
    const float4 a = B.rays[2 * i], b = B.rays[2 * i + 1];
    const RayCtx r = make_ray(B.terrain, V3{a.x, a.y, a.z}, a.w, V3{b.x, b.y, b.z}, b.w, B.apply_curvature != 0u);
    TraceHit h;
    if (B.any_hit >= 2u) {  // the frame kernel's stackless march: 2 = any hit, 3 = closest hit
        h = march_ray(B.terrain, r, B.any_hit == 2u, B.start_in_cell != 0u, pend);
    } else {
        h = trace_terrain(B.terrain, r, B.any_hit != 0u, pend);
    }
    B.out_hit[i] = h.hit ? 1u : 0u;
    if (B.out_t) B.out_t[i] = h.t;
    if (B.out_normal) {
        B.out_normal[3 * i + 0] = h.hit ? h.n.x : 0.0f;
        B.out_normal[3 * i + 1] = h.hit ? h.n.y : 0.0f;
        B.out_normal[3 * i + 2] = h.hit ? h.n.z : 0.0f;
    }
}

// ---- acceleration-table builders (reference build_minmax_mips,
// terrain_heightfield.rs:132-202, runs single-threaded on the CPU) -------------------
__global__ void k_leaf_build(const PyramidBuildParams B) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x < B.leaf_dim_x && y < B.leaf_dim_y) leaf_build_at(B, x, y);
}

__global__ void k_band_build(const BandBuildParams B) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, z = blockIdx.y * blockDim.y + threadIdx.y;
    if (x < B.width && z < B.height) band_build_at(B, x, z);
}

__global__ void k_level_build(const LevelBuildParams B) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x < B.dst_dim_x && y < B.dst_dim_y) level_build_at(B, x, y);
}

// ---- far-horizon table of the IBL rays (f3d_cone.h): one lane per DEM block, 8 x 8 neighbouring blocks to a wave ------
struct HorizonBuildParams {
    TerrainDev terrain;
    uint32_t level, bx, bz;
    float *table;  // [bz][bx][kIblSectors]
};
__global__ __launch_bounds__(kWave) void k_horizon_build(const HorizonBuildParams B) {
    const uint32_t tiles_x = (B.bx + 7u) >> 3;
    const uint32_t x = (blockIdx.x % tiles_x) * 8u + (threadIdx.x & 7u), z = (blockIdx.x / tiles_x) * 8u + (threadIdx.x >> 3);
    if (x >= B.bx || z >= B.bz) return;
    float out[kIblSectors];
    horizon_block_build(B.terrain, B.level, x, z, out);
    float4 *dst = reinterpret_cast<float4 *>(B.table + ((size_t)z * B.bx + x) * kIblSectors);
    dst[0] = float4{out[0], out[1], out[2], out[3]};
    dst[1] = float4{out[4], out[5], out[6], out[7]};
}

}  // namespace f3d
