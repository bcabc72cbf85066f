// forge3d_amd/csrc/f3d_lds.h -- the per-wave LDS context of the terrain march on the device (gfx950 only).
// Shared by the terrain path tracer's kernels (f3d_kernels.hip) and the PBR path tracer's terrain primitive
// (f3d_wavefront.hip): leaf FIFO columns, park rows, the verdict board of the ray sharing and the level table.
#pragma once

#include "f3d_march.h"

namespace f3d {

constexpr int kWave = 64;

// Lane id from the hardware, never from a register that has to stay alive: volatile, so it is neither hoisted nor merged.
__device__ __forceinline__ uint32_t lane_now() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// Per-wave LDS scratch of the traversal: a copy of the per-level layout tables, so that a lane
// can look its (per-lane) level up with one ds_read_b64 instead of a vector load from the kernarg
// segment, and -- only for the sorted descent kept for the test hook / A-B builds -- the
// pending-sibling words as a [level][lane] column (bank = lane, conflict-free).
// Rows kParkRow.. of the column hold the sample-lane frame's accumulators between rounds and its frame-head record.
// 6 144 bytes a wave, and not a row more: LDS is handed out in 1 280-byte pieces on gfx950 (tools/experiments/lds_granule.hip),
// so 6 656 bytes cost 7 680 and a CU held 21 one-wave workgroups instead of 24 -- measured 4 % of the frame.  Hence the verdict board
// shares a word with the head's flags (its bit is set by other lanes: an LDS atomic) and the candidate's light-type flag
// rides in bit 31 of its sample count, as in the packed reservoir.
constexpr int kParkRow = kFifoWords * kLeafFifoRows, kParkHead = 6, kParkWords = 8;  // 6 accumulator words + the frame head's record {reuse weight, flags}
constexpr int kBoardRow = kParkRow + kParkHead + 1;  // verdict board of the ray sharing (f3d_march.h): bit 31 of the flags word
constexpr uint32_t kBoardBit = 0x80000000u;
constexpr int kLdsRows = kParkRow + kParkWords > kMaxLevels ? kParkRow + kParkWords : kMaxLevels;
constexpr int kLdsWords = kLdsRows * kWave + 4 * kMaxLevels;
template <int BOARD_ROW, bool MESH_STACK = true, bool SHARE_CLOSEST = false>
struct LdsPendingAt {
    // closest-hit rays are shared too (f3d_march.h march_shared_closest): the board's row is the board's alone, a lane's whole
    // word is free for the key of the nearest hit its ray's slices have found
    static constexpr bool kShareClosest = SHARE_CLOSEST;
    // may the scene hold a mesh?  (closest_hit / occluded, f3d_shade.h: the terrain-only frame kernels are compiled without
    // the mesh walk, so nothing the mesh path needs can move their register allocation -- and the other way round)
    // MESH_STACK = false: the user never walks a mesh 4 wide through stack_put / stack_get (the PBR tracer walks its instanced
    // meshes through the threaded BVH), so the leaf FIFO's rows need not hold kBvh4MaxLevels words.
    static constexpr bool kMesh = MESH_STACK;
    uint32_t *col;          // lds + lane
    const uint32_t *table;  // lds + kLdsRows * kWave: {band_offset, band_shift, node_offset, tiles_x} per level
    uint32_t leaf_quorum, share_below;
    __device__ __forceinline__ void put(uint32_t level, uint32_t word) { col[level * kWave] = word; }
    __device__ __forceinline__ uint32_t get(uint32_t level) const { return col[level * kWave]; }
    // per-level words of the 4-wide mesh walk (f3d_shade.h mesh_bvh4): the rows of the leaf FIFO, which is empty while a
    // mesh is walked (closest_hit walks the mesh before the terrain; occluded() after the terrain march has drained)
    // (asserted where the walk would use the rows: a translation unit with one-word FIFO entries and no 4-wide walk -- the PBR
    // tracer's -- may still name these types)
    template <bool M = MESH_STACK>
    __device__ __forceinline__ void stack_put(uint32_t level, uint32_t word) {
        static_assert(kFifoWords * kLeafFifoRows >= kBvh4MaxLevels || !M, "the 4-wide mesh walk keeps a word per level in the leaf FIFO's rows");
        col[level * kWave] = word;
    }
    __device__ __forceinline__ uint32_t stack_get(uint32_t level) const { return col[level * kWave]; }
    __device__ __forceinline__ void note(int) const {}  // step-statistics hook (host emulator only)
    __device__ __forceinline__ void feature(float) const {}
    __device__ __forceinline__ void hint(float) const {}
    // (leaf-gate A/B build only) may the lanes that hold a fat leaf solve it now?
    __device__ __forceinline__ bool leaf_gate(bool at_leaf) const {
        const unsigned long long leaf = __ballot(at_leaf), inner = __ballot(!at_leaf);
        return (uint32_t)__popcll(leaf) >= leaf_quorum || inner == 0ull;
    }
    // ---- deferred leaf FIFO of the march (f3d_march.h): kFifoWords words per entry in the lane's column ----
    __device__ __forceinline__ void fifo_put(uint32_t k, uint32_t cell, float lo, float hi) {
        uint32_t *e = col + mul24(k, kFifoWords * kWave);
        e[0] = cell;
        if (kFifoWords == 3u) {
            e[kWave] = f_bits(lo);
            e[2 * kWave] = f_bits(hi);
        }
    }
    __device__ __forceinline__ void fifo_retag(uint32_t k, uint32_t cell) { col[mul24(k, kFifoWords * kWave)] = cell; }
    __device__ __forceinline__ void fifo_get(uint32_t k, uint32_t &cell, float &lo, float &hi) const {
        const uint32_t *e = col + mul24(k, kFifoWords * kWave);
        cell = e[0];
        lo = hi = 0.0f;  // (one-word entries: the drain forms the interval again)
        if (kFifoWords == 3u) {
            lo = f_from_bits(e[kWave]);
            hi = f_from_bits(e[2 * kWave]);
        }
    }
    // drain now?  enough lanes have a leaf queued, or a FIFO is full, or nobody marches any more
    __device__ __forceinline__ bool flush_now(uint32_t queued, bool marching) const {
        const unsigned long long have = __ballot(queued != 0u);
        if (have == 0ull) return false;
        return (uint32_t)__popcll(have) >= leaf_quorum || __ballot(queued >= kLeafFifo) != 0ull ||
               __ballot(marching) == 0ull;
    }
    __device__ __forceinline__ bool any(bool pred) const { return __ballot(pred) != 0ull; }
    // ---- ray sharing (f3d_march.h march_shared) ----
    // The lane id is REMATERIALISED where it is used (two v_mbcnt, no input): as a value derived from threadIdx.x it was
    // hoisted out of every loop together with the masks built from it and lived in scratch across the whole kernel.
    __device__ __forceinline__ uint32_t lane() const { return lane_now(); }
    __device__ __forceinline__ bool share_now(bool marching) const { return share_now(marching, share_below); }
    __device__ __forceinline__ bool share_now(bool marching, uint32_t below) const {
        const uint32_t n = (uint32_t)__popcll(__ballot(marching));
        return n != 0u && n <= below && (uint32_t)__popcll(__ballot(true)) >= kShareAvail * n;
    }
    // the verdict board lives in row kBoardRow of the WAVE's columns: board[l] = col[l - lane]
    // (volatile: lanes talk to each other through it without a barrier -- one wave, LDS operations in order; the pointer
    // is an LDS pointer by TYPE: address-space inference skips volatile accesses, and as generic ones they were flat_load /
    // flat_store through a 64-bit address that sat in scratch)
    using LdsWord = __attribute__((address_space(3))) volatile uint32_t;
    __device__ __forceinline__ LdsWord *board() const { return (LdsWord *)(col - lane() + BOARD_ROW * kWave); }
    __device__ __forceinline__ void verdict_post(bool hit) const {  // every lane, before any verdict_set of the call
        LdsWord *mine = (LdsWord *)(col + BOARD_ROW * kWave);
        const uint32_t w = *mine;
        *mine = hit ? w | kBoardBit : w & ~kBoardBit;
    }
    __device__ __forceinline__ void verdict_set(uint32_t owner) const {  // (several lanes may name one owner; its other bits stay)
        __hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t *)(board() + (owner & (kWave - 1u))), kBoardBit, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __device__ __forceinline__ bool verdict_get(uint32_t owner) const { return (board()[owner & (kWave - 1u)] & kBoardBit) != 0u; }
    // nearest hit of a shared closest-hit ray: kBoardBit | the complement of the 27-bit key, so that an atomic MAX keeps the
    // smallest key and 0 says "none"
    __device__ __forceinline__ void nearest_post() const {
        static_assert(SHARE_CLOSEST, "the board's word is shared with other flags in this layout");
        *(LdsWord *)(col + BOARD_ROW * kWave) = 0u;
    }
    __device__ __forceinline__ void nearest_set(uint32_t owner, uint32_t key) const {
        __hip_atomic_fetch_max((__attribute__((address_space(3))) uint32_t *)(board() + (owner & (kWave - 1u))), kBoardBit | (0x07FFFFFFu - key),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __device__ __forceinline__ uint32_t nearest_get(uint32_t owner) const {
        const uint32_t w = board()[owner & (kWave - 1u)];
        return (w & kBoardBit) != 0u ? 0x07FFFFFFu - (w & 0x07FFFFFFu) : kNoNearest;
    }
    // wave primitives of march_deal (f3d_march.h)
    __device__ __forceinline__ unsigned long long ballot(bool pred) const { return __ballot(pred); }
    __device__ __forceinline__ float shfl(float v, int src) const { return __shfl(v, src, kWave); }
    __device__ __forceinline__ uint32_t shfl(uint32_t v, int src) const { return (uint32_t)__shfl((int)v, src, kWave); }
    __device__ __forceinline__ float fast_log2(float x) const { return __builtin_amdgcn_logf(x); }
    __device__ __forceinline__ float fast_exp2(float x) const { return __builtin_amdgcn_exp2f(x); }
    template <bool CURVED>
    __device__ __forceinline__ void deal(const TerrainDev &T, MarchSlice &s, MarchState &m) const {
        march_deal<CURVED>(T, s, m, *this);
    }
    __device__ __forceinline__ void band_entry(const TerrainDev &, uint32_t level, uint32_t &offset,
                                               uint32_t &shift) const {
        const uint2 e = *reinterpret_cast<const uint2 *>(table + 4u * level);
        offset = e.x;
        shift = e.y;
    }
    __device__ __forceinline__ void level_entry(const TerrainDev &, uint32_t level, uint32_t &offset,
                                                uint32_t &tiles_x) const {
        const uint2 e = *reinterpret_cast<const uint2 *>(table + 4u * level + 2u);
        offset = e.x;
        tiles_x = e.y;
    }
};
using LdsPending = LdsPendingAt<kBoardRow>;
// The PBR path tracer's terrain primitive: leaf FIFO rows, then the board's row, then kPathParkRows rows in which a lane
// parks its path's loop-carried state across the marches (f3d_wf_path.h, round 5), then the level table.  With 8 park
// rows the block is 6 400 bytes -- five of the 1 280-byte pieces LDS is handed out in: 25 one-wave workgroups a CU, room
// for six waves a SIMD; 13 rows (7 680 bytes, 21 workgroups) go with five waves.
#ifndef F3D_WF_PARK
#define F3D_WF_PARK 8  // (f3d_wavefront.hip asks for 18 together with one-word FIFO entries; other translation units never use the compact block)
#endif
constexpr int kPathParkRows = F3D_WF_PARK;
constexpr int kPathParkRow0 = kParkRow + 1;
constexpr int kCompactRows = kParkRow + 1 + kPathParkRows;
constexpr int kCompactLdsWords = kCompactRows * kWave + 4 * kMaxLevels;
#if !defined(F3D_WF_NO_SHARE_CLOSEST)  // (A/B: camera and bounce rays marched to their end by their own lanes, the round-5 form)
using LdsPendingCompact = LdsPendingAt<kParkRow, false, true>;
#else
using LdsPendingCompact = LdsPendingAt<kParkRow, false, false>;
#endif
// rows: lane-column rows in front of the level table (the occlusion-stream kernels need the leaf FIFO only)
struct LdsPendingTerrainOnly : LdsPending {
    static constexpr bool kMesh = false;
};
template <bool MESH>
struct PendingFor {
    using type = LdsPending;
};
template <>
struct PendingFor<false> {
    using type = LdsPendingTerrainOnly;
};

template <class Pending = LdsPending>
__device__ __forceinline__ Pending make_pending(uint32_t *lds, const TerrainDev &T, uint32_t rows = kLdsRows) {
    const uint32_t lane = threadIdx.x & (kWave - 1u);  // `lds` is this WAVE's block (workgroups may hold several)
    if (lane < kMaxLevels) {
        uint32_t *e = lds + rows * kWave + 4 * lane;
        e[0] = T.band_offset[lane];
        e[1] = T.band_shift[lane];
        e[2] = T.node_offset[lane];
        e[3] = T.tiles_x[lane];
    }
    __syncthreads();
    return Pending{lds + lane, lds + rows * kWave, T.leaf_quorum ? T.leaf_quorum : kDefaultLeafQuorum,
                   T.share_below ? (T.share_below < 64u ? T.share_below : 64u) : kShareBelow};
}

}  // namespace f3d
