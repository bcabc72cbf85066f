// forge3d_amd/csrc/f3d_smoke_sim.hip -- smoke transport solver on gfx950: one launch per pass of SmokeVolume::step
// (reference src/smoke/sim.rs:47-139), a lane per voxel; per-voxel arithmetic in f3d_smoke_sim.h.  C ABI:
// f3d_smoke_step (include/f3d_terrain_pt.h).  The state lives on the device for all `steps` of a call; BASELINE.json
// configs[4] (a 120-frame sequence) advances it frame by frame and renders each state with f3d_smoke_render.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <vector>

#include "../../include/f3d_terrain_pt.h"
#include "f3d_devmem.h"
#include "f3d_setup.h"
#include "f3d_smoke_sim.h"

using namespace f3d;
using namespace f3d::smoke;

namespace {

struct SimKernelArgs {
    SimGrid G;
    SimFields F;
    SimSettings S;
    SimEmitter E;
    float *a, *b, *c;   // pass-specific buffers (src / dst / aux)
    const float *sums;  // [0..2] total mass, sum x mass, sum z mass; [3] mass before the density advection
    float dt;
    uint32_t u0, u1;
};
enum Pass : uint32_t { kEmit, kForces, kAdvectVec, kDiffuse, kCurl, kConfine, kDivergence, kJacobi, kGradient, kBoundary, kLaneShear, kPredict,
                       kCorrect, kSubgrid, kDecay, kScale };

__device__ __forceinline__ bool voxel_of(const SimGrid &G, uint32_t &x, uint32_t &y, uint32_t &z) {
    x = blockIdx.x * blockDim.x + threadIdx.x;
    y = blockIdx.y;
    z = blockIdx.z;
    return x < G.nx;
}
template <uint32_t PASS>
__global__ __launch_bounds__(64) void k_sim(const SimKernelArgs A) {
    uint32_t x, y, z;
    if (!voxel_of(A.G, x, y, z)) return;
    const size_t i = sim_index(A.G, x, y, z);
    if (PASS == kEmit) sim_emit(A.G, A.F, A.E, A.dt, x, y, z);
    else if (PASS == kForces) sim_forces(A.G, A.F, A.S, x, y, z);
    else if (PASS == kAdvectVec) sim_advect_vector(A.G, A.a, A.b, A.dt, x, y, z);
    else if (PASS == kDiffuse) sim_diffuse(A.G, A.a, A.b, A.dt /* alpha */, A.u0, A.u1, x, y, z);
    else if (PASS == kCurl) sim_curl(A.G, A.F.velocity, A.a, A.b, x, y, z);
    else if (PASS == kConfine) sim_confine(A.G, A.a, A.b, A.F.velocity, A.S.vorticity, A.S.dt, x, y, z);
    else if (PASS == kDivergence) sim_divergence(A.G, A.F.velocity, A.a, x, y, z);
    else if (PASS == kJacobi) sim_jacobi(A.G, A.a, A.c, A.b, x, y, z);
    else if (PASS == kGradient) sim_subtract_gradient(A.G, A.a, A.F.velocity, x, y, z);
    else if (PASS == kBoundary) sim_boundary(A.G, A.F, A.S, x, y, z);
    else if (PASS == kLaneShear) sim_lane_shear(A.G, A.F, A.S, A.sums, x, y, z);
    else if (PASS == kPredict) sim_advect_predict(A.G, A.a, A.F.velocity, A.b, A.dt, x, y, z);
    else if (PASS == kCorrect) sim_advect_correct(A.G, A.a, A.F.velocity, A.b, A.c, A.dt, x, y, z);
    else if (PASS == kSubgrid) sim_subgrid(A.G, A.F, A.S, x, y, z);
    else if (PASS == kDecay) sim_decay(A.G, A.F, A.S, i);
    else if (PASS == kScale) {  // scale_to_mass, sim.rs:699-711: sums[3] = target, sums[0] = the mass now
        const float target = A.sums[3], mass = A.sums[0];
        if (target > 0.0f && mass > 1.0e-12f) A.F.density[i] *= target / mass;
    }
}
// grid sums: rows -> slabs -> total (f3d_smoke_sim.h); kinds packed as bits of `kinds`, results to out[kind slot]
struct SumArgs {
    SimGrid G;
    const float *density;
    float *rows, *slabs, *out;  // rows: [4][nz * ny], slabs: [4][nz], out: [4]
    uint32_t kinds[4], n_kinds, out_slot[4];
};
__global__ void k_sum_rows(const SumArgs A) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.G.ny * A.G.nz) return;
    for (uint32_t k = 0u; k < A.n_kinds; k++) A.rows[(size_t)k * A.G.ny * A.G.nz + r] = sim_sum_row(A.G, A.density, A.kinds[k], r % A.G.ny, r / A.G.ny);
}
__global__ void k_sum_slabs(const SumArgs A) {
    const uint32_t z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= A.G.nz) return;
    for (uint32_t k = 0u; k < A.n_kinds; k++) A.slabs[(size_t)k * A.G.nz + z] = sim_sum_seq(A.rows + (size_t)k * A.G.ny * A.G.nz + (size_t)z * A.G.ny, A.G.ny);
}
__global__ void k_sum_total(const SumArgs A) {
    if (threadIdx.x < A.n_kinds) A.out[A.out_slot[threadIdx.x]] = sim_sum_seq(A.slabs + (size_t)threadIdx.x * A.G.nz, A.G.nz);
}

// ---- the step as PHASES (round 5) -------------------------------------------------------------------------------------
// A step of the launch-per-pass form above is ~85 launches and copies of 8-11 us each over a grid whose every pass is 1-2 us
// of memory traffic (786 432 voxels in BASELINE.json configs[4]): 0.82 ms of launch ramps.  Per voxel the arithmetic of a
// pass is a function of the previous passes' fields (f3d_smoke_sim.h), so passes can share a launch whenever no voxel reads
// what ANOTHER voxel writes in it.  The phases below are that grouping -- 17 + the Jacobi sweeps instead of ~55 + the sweeps:
//   * pointwise passes ride with their neighbours in the step (emission clear + emitters + forces; gradient + boundary
//     conditions; mass scaling + sub-grid eddies; diffusion + decay);
//   * the three velocity components are diffused in one phase, the five scalar fields advected in one phase from ONE
//     back-trace (the reference re-derives the identical back-traced position per field) and diffused in one phase;
//   * no device-to-device copies and no clears: passes alternate between a field and its twin buffer, and the one pass
//     that would leave a result in the wrong place writes it home in the phase that follows;
//   * the four grid sums in front of the density advection (mass, centroid) are one row pass with four accumulators.
// Per voxel every phase calls the launch-per-pass form's functions with the same arguments, so all forms and the oracle
// agree bit for bit (tests/test_smoke_sim.py).  Two drivers run the phases:
//   fused       (default) one launch per phase, (x, y) flattened so that a 96-wide row does not leave a third of a wave idle;
//   persistent  ONE cooperative launch for all steps, a grid barrier between phases -- built to get rid of the launches
//               altogether and measured (tools/experiments/sim_phases.py): on this chip a grid barrier costs MORE than a
//               launch, 10 us of same-address atomics and polling + 13 us of cache maintenance (an agent-scope release
//               writes the XCD's L2 back, an acquire invalidates it: the eight L2s are not coherent with each other) against
//               ~3.5 us of work per phase; 1.49 ms a step.  Kept as a tested form (F3D_SMOKE_SOLVER=persistent).
enum Phase : uint32_t { kPhEmitForces, kPhAdvectVec, kPhDiffuseVec, kPhCurl, kPhConfine, kPhDivergence, kPhJacobi, kPhGradientBoundary,
                        kPhLaneShear, kPhAdvectScalars, kPhCorrectScalars, kPhScaleSubgrid, kPhDiffuseDecay, kPhJacobiTwice };
constexpr uint32_t kInlineEmitters = 4u;  // emitters that travel in the kernel arguments (a host-to-device copy of pageable memory waits for the stream)
struct StepArgs {
    SimGrid G;
    SimFields F;
    SimSettings S;
    const SimEmitter *emitters;  // more than kInlineEmitters: a device copy
    SimEmitter inline_emitters[kInlineEmitters];
    uint32_t emitter_count, steps;
    float *vel_b, *curl, *mag, *div, *pres_b;
    float *adv[5], *adv2[5];  // advected scalars (predictor; MacCormack-corrected)
    float *rows, *slabs, *sums;
    unsigned int *barrier;  // persistent form: [0] arrivals, [1] generation, [2] abort
};
struct PhaseCtx {
    float *cur, *next;   // Jacobi: pressure in, pressure out; gradient: the final pressure
    uint32_t corrected;  // the advected scalars are in adv2 (MacCormack) instead of adv
};
template <uint32_t PH>
__device__ __forceinline__ void phase_voxel(const StepArgs &A, const SimGrid &G, const PhaseCtx &C, uint32_t v, uint32_t x, uint32_t y, uint32_t z) {
    const SimFields &F = A.F;
    const SimSettings &S = A.S;
    if (PH == kPhEmitForces) {
        F.emission_rate[v] = 0.0f;
        for (uint32_t e = 0u; e < A.emitter_count; e++) {
            const SimEmitter E = A.emitter_count <= kInlineEmitters ? A.inline_emitters[e] : A.emitters[e];
            if (G.time_seconds >= E.start_time && G.time_seconds <= E.end_time) sim_emit(G, F, E, S.dt, x, y, z);
        }
        sim_forces(G, F, S, x, y, z);
    } else if (PH == kPhAdvectVec) {
        sim_advect_vector(G, F.velocity, A.vel_b, S.dt, x, y, z);
    } else if (PH == kPhDiffuseVec) {  // diffuse_vector (sim.rs:739-754), or the advected velocity as it is: home either way
        if (S.diffusion > 0.0f) {
            for (uint32_t c = 0u; c < 3u; c++) sim_diffuse(G, A.vel_b, F.velocity, S.diffusion * S.dt, 3u, c, x, y, z);
        } else {
            for (uint32_t c = 0u; c < 3u; c++) F.velocity[3u * (size_t)v + c] = A.vel_b[3u * (size_t)v + c];
        }
    } else if (PH == kPhCurl) {
        sim_curl(G, F.velocity, A.curl, A.mag, x, y, z);
    } else if (PH == kPhConfine) {
        sim_confine(G, A.curl, A.mag, F.velocity, S.vorticity, S.dt, x, y, z);
    } else if (PH == kPhDivergence) {
        sim_divergence(G, F.velocity, A.div, x, y, z);
        F.pressure[v] = 0.0f;
    } else if (PH == kPhJacobi) {
        sim_jacobi(G, C.cur, A.div, C.next, x, y, z);
    } else if (PH == kPhJacobiTwice) {
        sim_jacobi_twice(G, C.cur, A.div, C.next, x, y, z);
    } else if (PH == kPhGradientBoundary) {
        sim_subtract_gradient(G, C.cur, F.velocity, x, y, z);
        if (C.cur != F.pressure) F.pressure[v] = C.cur[v];  // (an odd number of sweeps: the field goes home here)
        sim_boundary(G, F, S, x, y, z);
    } else if (PH == kPhLaneShear) {
        sim_lane_shear(G, F, S, A.sums, x, y, z);
    } else if (PH == kPhAdvectScalars) {  // advect_scalar x 5, first pass (sim.rs:603-613): one back-trace serves all five
        const float *old[5] = {F.density, F.temperature, F.fuel, F.soot, F.humidity};
        float bx, by, bz;
        sim_back(G, F.velocity, S.dt, x, y, z, bx, by, bz);
#pragma unroll
        for (uint32_t f = 0u; f < 5u; f++) A.adv[f][v] = f_max(sim_sample(G, old[f], bx, by, bz, 1u, 0u), 0.0f);
    } else if (PH == kPhCorrectScalars) {
        const float *old[5] = {F.density, F.temperature, F.fuel, F.soot, F.humidity};
#pragma unroll
        for (uint32_t f = 0u; f < 5u; f++) sim_advect_correct(G, old[f], F.velocity, A.adv[f], A.adv2[f], S.dt, x, y, z);
    } else if (PH == kPhScaleSubgrid) {
        float *const *adv = C.corrected ? A.adv2 : A.adv;
        SimFields T = F;  // the advected fields, where they are: scale_to_mass (sim.rs:699-711) and the sub-grid eddies act on them
        T.density = adv[0];
        T.temperature = adv[1];
        T.fuel = adv[2];
        T.soot = adv[3];
        T.humidity = adv[4];
        if (S.mass_conservation != 0) {
            const float target = A.sums[3], mass = A.sums[0];
            if (target > 0.0f && mass > 1.0e-12f) T.density[v] *= target / mass;
        }
        sim_subgrid(G, T, S, x, y, z);
    } else if (PH == kPhDiffuseDecay) {  // apply_scalar_diffusion (sim.rs:236-245) home, then decay and ageing of the voxel
        float *const *adv = C.corrected ? A.adv2 : A.adv;
        float *home[5] = {F.density, F.temperature, F.fuel, F.soot, F.humidity};
#pragma unroll
        for (uint32_t f = 0u; f < 5u; f++) {
            if (S.diffusion > 0.0f) sim_diffuse(G, adv[f], home[f], S.diffusion * S.dt, 1u, 0u, x, y, z);
            else home[f][v] = adv[f][v];
        }
        sim_decay(G, F, S, v);
    }
}
// grid sums of `density`: one lane a row, every kind in one sweep over x (the adds of a kind in sim_sum_row's order)
struct SumKinds {
    uint32_t n, kind[4], slot[4];
};
__device__ __forceinline__ void sums_row(const StepArgs &A, const SimGrid &G, const float *density, const SumKinds &K, uint32_t r) {
    const uint32_t n_rows = G.ny * G.nz, y = r % G.ny, z = r / G.ny;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    auto term = [&](uint32_t kind, float d, float m, uint32_t x) {
        return kind == 0u ? d : (kind == 1u ? m : (kind == 2u ? (float)x * m : (float)z * m));
    };
    for (uint32_t x = 0u; x < G.nx; x++) {
        const float d = density[sim_index(G, x, y, z)], m = f_max(d, 0.0f);
        a0 += term(K.kind[0], d, m, x);
        if (K.n > 1u) a1 += term(K.kind[1], d, m, x);
        if (K.n > 2u) a2 += term(K.kind[2], d, m, x);
        if (K.n > 3u) a3 += term(K.kind[3], d, m, x);
    }
    A.rows[r] = a0;
    if (K.n > 1u) A.rows[(size_t)n_rows + r] = a1;
    if (K.n > 2u) A.rows[2u * (size_t)n_rows + r] = a2;
    if (K.n > 3u) A.rows[3u * (size_t)n_rows + r] = a3;
}
// ... then the rows of a slab and the slabs by ONE workgroup, in the order of sim_sum_seq
__device__ __forceinline__ void sums_finish(const StepArgs &A, const SimGrid &G, const SumKinds &K) {
    const uint32_t n_rows = G.ny * G.nz;
    for (uint32_t j = threadIdx.x; j < K.n * G.nz; j += blockDim.x) {
        const uint32_t k = j / G.nz, z = j - k * G.nz;
        A.slabs[(size_t)k * G.nz + z] = sim_sum_seq(A.rows + (size_t)k * n_rows + (size_t)z * G.ny, G.ny);
    }
    __syncthreads();
    if (threadIdx.x < K.n) A.sums[K.slot[threadIdx.x]] = sim_sum_seq(A.slabs + (size_t)threadIdx.x * G.nz, G.nz);
}
__device__ __forceinline__ SumKinds sums_before_advection(const SimSettings &S) {
    // mass and centroid for the lane shear, mass before the density advection (the shear moves velocity only, so the
    // density these are taken from is the same: one pass)
    SumKinds K{};
    if (S.turbulence_strength > 0.0f) {
        K.kind[K.n] = 1u, K.slot[K.n++] = 0u;
        K.kind[K.n] = 2u, K.slot[K.n++] = 1u;
        K.kind[K.n] = 3u, K.slot[K.n++] = 2u;
    }
    if (S.mass_conservation != 0) K.kind[K.n] = 0u, K.slot[K.n++] = 3u;
    return K;
}

// fused driver: one launch per phase, a lane per voxel
constexpr uint32_t kPhaseBlock = 256u;
template <uint32_t PH>
__global__ __launch_bounds__(kPhaseBlock) void k_phase(const StepArgs A, const PhaseCtx C) {
    const uint32_t n = A.G.nx * A.G.ny * A.G.nz, v = blockIdx.x * kPhaseBlock + threadIdx.x;
    if (v >= n) return;
    const uint32_t plane = A.G.nx * A.G.ny, z = v / plane, y = (v - z * plane) / A.G.nx, x = v - z * plane - y * A.G.nx;
    phase_voxel<PH>(A, A.G, C, v, x, y, z);
}
// ---- K Jacobi sweeps a launch, the tile and its halo in LDS (round 6) ------------------------------------------------------
// A sweep of the pressure solve (sim_jacobi) is 3 MB in, 3 MB out and ~5 us of launch for BASELINE configs[4]'s grid, thirty times
// a step: 45 % of the step, issuing 12 % of its vector slots.  Here a workgroup owns a tile of kJacTX x kJacTY x kJacTZ voxels,
// loads it with a halo of K voxels (y, z; x too when a row does not fit one tile -- otherwise the row's two border voxels) into
// LDS and runs K sweeps there, each over the voxels whose K-sweep history the halo still covers (the region shrinks by one ring a
// sweep), the divergence read from L2 each time; the tile's voxels go out after the K-th.  Per voxel and sweep the expression is
// sim_jacobi's -- the six neighbours added in its order, minus the divergence, divided by 6; 0 on the border -- on the values the
// one-sweep launches would have read: bit-identical (tests/test_smoke_sim.py runs both).  K = 4: a 98 x 16 x 12 region twice =
// 150 KB of the CU's 160 KB of LDS, one workgroup of 1 024 lanes a CU, 256 tiles for the 96 x 64 x 128 grid; 20 + 10 sweeps are
// 5 + 3 launches instead of 30.  Measured slower than the one-sweep launches (see where it is launched): opt-in, F3D_SMOKE_JACOBI=tiled.
constexpr uint32_t kJacTX = 96u, kJacTY = 8u, kJacTZ = 4u, kJacK = 4u, kJacLanesX = 128u, kJacRows = 8u;
constexpr uint32_t kJacRegion = (kJacTX + 2u * kJacK) * (kJacTY + 2u * kJacK) * (kJacTZ + 2u * kJacK);  // 104 x 16 x 12
static_assert(kJacTX + 2u * kJacK <= kJacLanesX, "a lane per voxel of a region row");
__global__ __launch_bounds__(kJacLanesX * kJacRows) void k_jacobi_tiled(const SimGrid G, const float *cur, const float *div, float *next, uint32_t sweeps,
                                                                          uint32_t tiles_x, uint32_t tiles_y) {
    __shared__ float buf[2][kJacRegion];
    const uint32_t tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, tz = blockIdx.x / (tiles_x * tiles_y);
    const uint32_t hx = tiles_x == 1u ? 1u : sweeps, h = sweeps;  // halo in x; in y and z
    const uint32_t ex = kJacTX + 2u * hx, ey = kJacTY + 2u * h, ez = kJacTZ + 2u * h;
    // region voxel (rx, ry, rz) is grid voxel (x0 + rx, y0 + ry, z0 + rz) (unsigned wrap-around below 0: outside the grid like beyond it)
    const uint32_t x0 = tx * kJacTX - hx, y0 = ty * kJacTY - h, z0 = tz * kJacTZ - h;
    // (a wave lies in one row: the row's number and everything derived from it are scalars)
    const uint32_t rx = threadIdx.x & (kJacLanesX - 1u), row0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kJacLanesX));
    const uint32_t gx = x0 + rx;
    const bool col = rx < ex;
    // sim_interior(G, x, y, z, 1) as one unsigned comparison an axis: x - 1 < nx - 2 (a coordinate "below 0" has wrapped around)
    const uint32_t inx = G.nx > 2u ? G.nx - 2u : 0u, iny = G.ny > 2u ? G.ny - 2u : 0u, inz = G.nz > 2u ? G.nz - 2u : 0u;
    // Everything that comes from memory is asked for at once, before the first sweep: the region's pressures (a lane takes the
    // voxel rx of every kJacRows-th row) and the divergences of the voxels it will update -- the same voxels in every sweep (the
    // inner rows of the region, numbered once; a sweep skips the rows its ring has given up), so they stay in registers.  (A first
    // form fetched the divergence inside the sweeps: a trip to L2 per row and sweep, 28 us a launch.)
    constexpr uint32_t kLoads = ((kJacTY + 2u * kJacK) * (kJacTZ + 2u * kJacK) + kJacRows - 1u) / kJacRows;            // 24
    constexpr uint32_t kInner = ((kJacTY + 2u * kJacK - 2u) * (kJacTZ + 2u * kJacK - 2u) + kJacRows - 1u) / kJacRows;  // 18
    float p[kLoads], d[kInner];
    const uint32_t iy = ey - 2u, iz = ez - 2u;  // the inner rows: ry in [1, ey - 2], rz in [1, ez - 2]
    // (row -> (y, z) without a division: rows advance by kJacRows <= the rows of a slab, so y wraps at most once a step)
    static_assert(kJacRows <= kJacTY, "a step of kJacRows rows crosses at most one slab");
    {
        uint32_t ry = row0, rz = 0u;
#pragma unroll
        for (uint32_t j = 0u; j < kLoads; j++) {
            const uint32_t gy = y0 + ry, gz = z0 + rz;
            p[j] = (col && rz < ez && gx < G.nx && gy < G.ny && gz < G.nz) ? cur[sim_index(G, gx, gy, gz)] : 0.0f;
            ry += kJacRows;
            if (ry >= ey) ry -= ey, rz++;
        }
    }
    {
        uint32_t ry = 1u + row0, rz = 1u;
#pragma unroll
        for (uint32_t j = 0u; j < kInner; j++) {
            const uint32_t gy = y0 + ry, gz = z0 + rz;
            d[j] = (col && rz <= iz && gx - 1u < inx && gy - 1u < iny && gz - 1u < inz) ? div[sim_index(G, gx, gy, gz)] : 0.0f;
            ry += kJacRows;
            if (ry > iy) ry -= iy, rz++;
        }
    }
#pragma unroll
    for (uint32_t j = 0u; j < kLoads; j++) {
        const uint32_t row = row0 + j * kJacRows;
        if (col && row < ey * ez) buf[0][row * ex + rx] = p[j];
    }
    __syncthreads();
    for (uint32_t s = 1u; s <= sweeps; s++) {
        const float *in = buf[(s - 1u) & 1u];
        float *out = buf[s & 1u];
        // rows whose y and z are at least s from the region's faces; in x every voxel but the region's first and last (with a
        // one-voxel x halo those are outside the grid or on its border: 0 either way; with a K-voxel one they are never read again
        // by a voxel that is still owed a right value)
        uint32_t ry = 1u + row0, rz = 1u;
#pragma unroll
        for (uint32_t j = 0u; j < kInner; j++, ry += kJacRows, ry > iy ? (ry -= iy, rz++) : 0u) {
            const uint32_t gy = y0 + ry, gz = z0 + rz;
            if (rz <= iz && ry >= s && ry + s < ey && rz >= s && rz + s < ez && col && rx >= 1u && rx + 1u < ex) {
                const uint32_t i = (rz * ey + ry) * ex + rx;
                float v = 0.0f;
                if (gx - 1u < inx && gy - 1u < iny && gz - 1u < inz) {
                    const float sum = in[i - 1u] + in[i + 1u] + in[i - ex] + in[i + ex] + in[i - ex * ey] + in[i + ex * ey];
                    v = (sum - d[j]) / 6.0f;
                }
                out[i] = v;
            }
        }
        __syncthreads();
    }
    const float *res = buf[sweeps & 1u];
    for (uint32_t r = row0; r < kJacTY * kJacTZ; r += kJacRows) {  // the tile goes out
        const uint32_t ry = h + r % kJacTY, rz = h + r / kJacTY, gy = y0 + ry, gz = z0 + rz;
        if (rx >= hx && rx < hx + kJacTX && gx < G.nx && gy < G.ny && gz < G.nz) next[sim_index(G, gx, gy, gz)] = res[(rz * ey + ry) * ex + rx];
    }
}

__global__ __launch_bounds__(64) void k_phase_sum_rows(const StepArgs A, const SumKinds K, const float *density) {
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r < A.G.ny * A.G.nz) sums_row(A, A.G, density, K, r);
}
__global__ __launch_bounds__(256) void k_phase_sum_finish(const StepArgs A, const SumKinds K) { sums_finish(A, A.G, K); }
// The same sums in the same order with the slab staged in LDS (round 5: a lane a row reading global memory touched 64 cache
// lines per load and the single finishing workgroup walked rows and slabs from L2 -- 24 + 19 us, twice a step, a fifth of
// the step).  A workgroup per z slab: coalesced copy into LDS (rows padded by one float: a lane a row would otherwise
// hit one bank), a wave per kind adds its rows along x, one lane per kind adds the slab's rows; a second, one-wave launch
// adds the slabs.
__global__ __launch_bounds__(256) void k_phase_sum_slabs(const StepArgs A, const SumKinds K, const float *density) {
    extern __shared__ float slab_lds[];
    const SimGrid &G = A.G;
    const uint32_t z = blockIdx.x, pitch = G.ny + 1u;  // x-major with an odd pitch: the copy's writes and the rows' reads both spread over the banks
    float *tile = slab_lds, *row_sum = slab_lds + (size_t)G.nx * pitch;
    for (uint32_t i = threadIdx.x; i < G.nx * G.ny; i += 256u) {
        const uint32_t y = i / G.nx, x = i - y * G.nx;
        tile[x * pitch + y] = density[sim_index(G, x, y, z)];
    }
    __syncthreads();
    const uint32_t k = threadIdx.x >> 6;
    if (k < K.n) {
        const uint32_t kind = K.kind[k];
        for (uint32_t r = threadIdx.x & 63u; r < G.ny; r += 64u) {
            float a = 0.0f;
#pragma unroll 8
            for (uint32_t x = 0u; x < G.nx; x++) {  // (sums_row's terms and order; the loads of eight steps go out together)
                const float d = tile[x * pitch + r], m = f_max(d, 0.0f);
                a += kind == 0u ? d : (kind == 1u ? m : (kind == 2u ? (float)x * m : (float)z * m));
            }
            row_sum[k * G.ny + r] = a;
        }
    }
    __syncthreads();
    if (threadIdx.x < K.n) {
        const float *v = row_sum + threadIdx.x * G.ny;
        float sum = 0.0f;
#pragma unroll 8
        for (uint32_t r = 0u; r < G.ny; r++) sum += v[r];  // (sim_sum_seq)
        A.slabs[(size_t)threadIdx.x * G.nz + z] = sum;
    }
}
__global__ __launch_bounds__(256) void k_phase_sum_total(const StepArgs A, const SumKinds K) {
    extern __shared__ float slab_lds[];
    const uint32_t n = K.n * A.G.nz;
    for (uint32_t i = threadIdx.x; i < n; i += 256u) slab_lds[i] = A.slabs[i];  // (one round trip for all of them instead of one per addend)
    __syncthreads();
    if (threadIdx.x < K.n) {
        const float *v = slab_lds + threadIdx.x * A.G.nz;
        float sum = 0.0f;
#pragma unroll 8
        for (uint32_t zz = 0u; zz < A.G.nz; zz++) sum += v[zz];  // (sim_sum_seq)
        A.sums[K.slot[threadIdx.x]] = sum;
    }
}

// persistent driver: the grid barrier.  Cache maintenance is most of its cost (see above), so the fences are executed by
// ONE wave of the workgroup -- with all sixteen executing them a barrier took 37 us and a Jacobi sweep of three voxels a lane
// 32 us.  The other waves are ordered behind that wave by the workgroup barriers on either side (they share its CU's L1,
// which its buffer_inv has emptied, and its XCD's L2).  A barrier that is not released within about a second (a workgroup
// that never arrived: cannot happen under a cooperative launch) sets the abort word and lets every lane run to the end, so
// a broken launch ends in an error, not in a hung GPU.
constexpr uint32_t kStepBlock = 1024u, kWave = 64u;
__device__ __forceinline__ void grid_barrier(unsigned int *bar) {
#if defined(F3D_SIM_PHASE_TIMES)  // diagnostics build (tools/experiments/sim_phases.py): when did workgroup 0 reach each barrier, when did it leave? (100 MHz)
    if (blockIdx.x == 0u && threadIdx.x == 0u) {
        const unsigned int k = bar[3];
        if (k < 2000u) reinterpret_cast<unsigned long long *>(bar + 4)[2u * k] = wall_clock64();
    }
#endif
    __syncthreads();  // every wave's stores of the phase have left it
    if (threadIdx.x < kWave) {  // (wave 0, all lanes: a fence is a wave's instruction)
#if !defined(F3D_SIM_BARRIER_NOFENCE)  // (timing experiment, wrong results)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
        if (threadIdx.x == 0u) {
            const unsigned int gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int arrived = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived == gridDim.x - 1u) {
                __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                uint32_t polls = 0u;
                while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                    if (++polls > (1u << 22)) {
                        __hip_atomic_store(&bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
#if !defined(F3D_SIM_BARRIER_NOFENCE)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    }
    __syncthreads();
#if defined(F3D_SIM_PHASE_TIMES)
    if (blockIdx.x == 0u && threadIdx.x == 0u) {
        const unsigned int k = bar[3];
        if (k < 2000u) reinterpret_cast<unsigned long long *>(bar + 4)[2u * k + 1u] = wall_clock64();
        bar[3] = k + 1u;
    }
#endif
}
__global__ __launch_bounds__(kStepBlock) void k_sim_step(const StepArgs A) {
    SimGrid G = A.G;
    const SimSettings &S = A.S;
    const uint32_t n = G.nx * G.ny * G.nz, lanes = gridDim.x * kStepBlock, lane = blockIdx.x * kStepBlock + threadIdx.x;
    const uint32_t plane = G.nx * G.ny;
    PhaseCtx C{};
#define F3D_SIM_PHASE(PH)                                                                                                     \
    {                                                                                                                         \
        for (uint32_t v = lane; v < n; v += lanes) {                                                                          \
            const uint32_t z = v / plane, y = (v - z * plane) / G.nx, x = v - z * plane - y * G.nx;                           \
            phase_voxel<PH>(A, G, C, v, x, y, z);                                                                             \
        }                                                                                                                     \
        grid_barrier(A.barrier);                                                                                              \
    }
    auto grid_sums = [&](const float *density, const SumKinds &K) {
        for (uint32_t r = lane; r < G.ny * G.nz; r += lanes) sums_row(A, G, density, K, r);
        grid_barrier(A.barrier);
        if (blockIdx.x == 0u) sums_finish(A, G, K);
        grid_barrier(A.barrier);
    };
    auto project = [&](uint32_t iterations) {  // sim.rs:270-317; ends with the boundary conditions that follow it in the step
        F3D_SIM_PHASE(kPhDivergence)
        C.cur = A.F.pressure;
        C.next = A.pres_b;
        for (uint32_t it = 0u; it < iterations; it++) {
            F3D_SIM_PHASE(kPhJacobi)
            float *t = C.cur;
            C.cur = C.next;
            C.next = t;
        }
        F3D_SIM_PHASE(kPhGradientBoundary)
    };
    for (uint32_t step = 0u; step < A.steps; step++) {  // SmokeVolume::step, sim.rs:47-139
        F3D_SIM_PHASE(kPhEmitForces)
        F3D_SIM_PHASE(kPhAdvectVec)
        F3D_SIM_PHASE(kPhDiffuseVec)
        if (S.vorticity > 0.0f) {
            F3D_SIM_PHASE(kPhCurl)
            F3D_SIM_PHASE(kPhConfine)
        }
        project(S.pressure_iterations > 1u ? S.pressure_iterations : 1u);
        const SumKinds before = sums_before_advection(S);
        if (before.n != 0u) grid_sums(A.F.density, before);
        if (S.turbulence_strength > 0.0f) F3D_SIM_PHASE(kPhLaneShear)
        F3D_SIM_PHASE(kPhAdvectScalars)
        C.corrected = 0u;
        if (S.mac_cormack != 0) {
            F3D_SIM_PHASE(kPhCorrectScalars)
            C.corrected = 1u;
        }
        if (S.mass_conservation != 0) {
            SumKinds after{};
            after.n = 1u;  // kind 0 -> slot 0
            grid_sums(C.corrected ? A.adv2[0] : A.adv[0], after);
        }
        F3D_SIM_PHASE(kPhScaleSubgrid)
        F3D_SIM_PHASE(kPhDiffuseDecay)
        project((S.pressure_iterations / 2u) > 1u ? S.pressure_iterations / 2u : 1u);
        G.time_seconds += S.dt;
        G.frame_index += 1u;
    }
#undef F3D_SIM_PHASE
}

void ok(hipError_t e, const char *what) {
    if (e != hipSuccess) fail(F3D_STATUS_DEVICE, "HIP failure in %s: %s", what, hipGetErrorString(e));
}

struct Sim {
    SimGrid G;
    SimFields F{};
    float *tmp_a = nullptr, *tmp_b = nullptr, *vec_a = nullptr, *curl = nullptr, *div = nullptr, *rows = nullptr, *slabs = nullptr, *sums = nullptr;
    size_t n = 0;
    dim3 grid, block;
    uint32_t n_scratch = 0;
    void *alloc(size_t bytes, const char *tag = nullptr) {  // stream-ordered scratch that stays for the next call (f3d_devmem.h workspace)
        char name[48];
        if (!tag) snprintf(name, sizeof(name), "smoke.sim.%u", n_scratch++);
        void *p = nullptr;
        ok(workspace(&p, tag ? tag : name, bytes), "smoke solver allocation");
        return p;
    }
    template <uint32_t PASS>
    void run(SimKernelArgs A) {
        A.G = G;
        A.F = F;
        A.sums = sums;
        hipLaunchKernelGGL(k_sim<PASS>, grid, block, 0, call_stream(), A);
    }
    void sum(std::initializer_list<std::pair<uint32_t, uint32_t>> kinds_slots) {  // (kind, slot of `sums`)
        SumArgs A{};
        A.G = G;
        A.density = F.density;
        A.rows = rows;
        A.slabs = slabs;
        A.out = sums;
        for (auto ks : kinds_slots) {
            A.kinds[A.n_kinds] = ks.first;
            A.out_slot[A.n_kinds++] = ks.second;
        }
        const uint32_t r = G.ny * G.nz;
        hipLaunchKernelGGL(k_sum_rows, dim3((r + 63u) / 64u), dim3(64), 0, call_stream(), A);
        hipLaunchKernelGGL(k_sum_slabs, dim3((G.nz + 63u) / 64u), dim3(64), 0, call_stream(), A);
        hipLaunchKernelGGL(k_sum_total, dim3(1), dim3(64), 0, call_stream(), A);
    }
};

void validate_step(const f3d_smoke_step_settings &s) {  // SmokeStepSettings::validate, types.rs:182-226
    const float vals[9] = {s.dt, s.density_decay, s.temperature_decay, s.velocity_damping, s.diffusion, s.buoyancy, s.vorticity, s.turbulence_strength,
                           s.boundary_damping};
    const char *names[9] = {"dt", "density_decay", "temperature_decay", "velocity_damping", "diffusion", "buoyancy", "vorticity", "turbulence_strength",
                            "boundary_damping"};
    for (int i = 0; i < 9; i++)
        if (!std::isfinite(vals[i])) fail(F3D_STATUS_VALUE, "%s must be finite", names[i]);
    if (s.dt <= 0.0f) fail(F3D_STATUS_VALUE, "dt must be > 0");
    if (s.density_decay < 0.0f || s.temperature_decay < 0.0f || s.velocity_damping < 0.0f || s.diffusion < 0.0f || s.vorticity < 0.0f ||
        s.turbulence_strength < 0.0f)
        fail(F3D_STATUS_VALUE, "decay, damping, diffusion, vorticity, and turbulence must be >= 0");
    if (!(s.boundary_damping >= 0.0f && s.boundary_damping <= 1.0f)) fail(F3D_STATUS_VALUE, "boundary_damping must be in [0, 1]");
    for (int a = 0; a < 3; a++)
        if (!std::isfinite(s.wind[a])) fail(F3D_STATUS_VALUE, "wind[%d] must be finite", a);
}
void validate_emitter(const f3d_smoke_emitter &e) {  // SmokeEmitter::validate, types.rs:101-134
    for (int a = 0; a < 3; a++)
        if (!std::isfinite(e.center[a])) fail(F3D_STATUS_VALUE, "center[%d] must be finite", a);
    if (!std::isfinite(e.radius) || e.radius <= 0.0f) fail(F3D_STATUS_VALUE, "radius must be finite and > 0");
    const float vals[8] = {e.density_rate, e.temperature_rate, e.fuel_rate, e.soot_rate, e.humidity_rate, e.emission_rate, e.start_time, e.end_time};
    const char *names[8] = {"density_rate", "temperature_rate", "fuel_rate", "soot_rate", "humidity_rate", "emission_rate", "start_time", "end_time"};
    for (int i = 0; i < 8; i++)
        if (!std::isfinite(vals[i])) fail(F3D_STATUS_VALUE, "%s must be finite", names[i]);
    for (int a = 0; a < 3; a++)
        if (!std::isfinite(e.velocity[a])) fail(F3D_STATUS_VALUE, "velocity[%d] must be finite", a);
    if (e.end_time < e.start_time) fail(F3D_STATUS_VALUE, "end_time must be >= start_time");
}

// project, sim.rs:270-317
void project(Sim &s, uint32_t iterations) {
    SimKernelArgs A{};
    A.a = s.div;
    s.run<kDivergence>(A);
    ok(hipMemsetAsync(s.F.pressure, 0, s.n * sizeof(float), call_stream()), "pressure clear");
    float *cur = s.F.pressure, *next = s.tmp_a;
    for (uint32_t it = 0; it < iterations; it++) {
        A = SimKernelArgs{};
        A.a = cur;
        A.b = next;
        A.c = s.div;
        s.run<kJacobi>(A);
        std::swap(cur, next);
    }
    if (cur != s.F.pressure) ok(hipMemcpyAsync(s.F.pressure, cur, s.n * sizeof(float), hipMemcpyDeviceToDevice, call_stream()), "pressure copy");
    A = SimKernelArgs{};
    A.a = s.F.pressure;
    s.run<kGradient>(A);
}
// advect_scalar into the same field, sim.rs:594-636
void advect(Sim &s, float *field, const SimSettings &S) {
    SimKernelArgs A{};
    A.a = field;
    A.b = s.tmp_a;
    A.dt = S.dt;
    s.run<kPredict>(A);
    if (S.mac_cormack) {
        A.c = s.tmp_b;  // a = old, b = predicted, c = corrected
        s.run<kCorrect>(A);
        ok(hipMemcpyAsync(field, s.tmp_b, s.n * sizeof(float), hipMemcpyDeviceToDevice, call_stream()), "advected field");
    } else {
        ok(hipMemcpyAsync(field, s.tmp_a, s.n * sizeof(float), hipMemcpyDeviceToDevice, call_stream()), "advected field");
    }
}
void diffuse(Sim &s, float *field, float alpha, uint32_t stride, uint32_t comp, float *scratch) {
    SimKernelArgs A{};
    A.a = field;
    A.b = scratch;
    A.dt = alpha;
    A.u0 = stride;
    A.u1 = comp;
    s.run<kDiffuse>(A);
}

}  // namespace

extern "C" int f3d_smoke_step(f3d_smoke_state *st, const f3d_smoke_step_settings *settings, const f3d_smoke_emitter *emitters,
                              uint32_t emitter_count, uint32_t steps, double *device_seconds, char *err, size_t errlen) {
    if (err && errlen) err[0] = 0;
    int rc = F3D_STATUS_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    try {
        if (!st || !settings || (emitter_count && !emitters)) fail(F3D_STATUS_VALUE, "null argument");
        for (int a = 0; a < 3; a++) {
            if (st->dims[a] < 2u) fail(F3D_STATUS_VALUE, "dims[%d] must be >= 2", a);
            if (!std::isfinite(st->voxel_size[a]) || st->voxel_size[a] <= 0.0f) fail(F3D_STATUS_VALUE, "voxel_size[%d] must be finite and > 0", a);
            if (!std::isfinite(st->origin[a])) fail(F3D_STATUS_VALUE, "origin[%d] must be finite", a);
        }
        const uint64_t n64 = (uint64_t)st->dims[0] * st->dims[1] * st->dims[2];
        if (n64 > (1ull << 24)) fail(F3D_STATUS_VALUE, "smoke domain has %llu voxels, exceeding CPU reference limit %llu", (unsigned long long)n64, 1ull << 24);
        float *host[9] = {st->density, st->temperature, st->fuel, st->soot, st->humidity, st->emission_rate, st->particle_age, st->velocity, st->pressure};
        for (float *p : host)
            if (!p) fail(F3D_STATUS_VALUE, "all nine smoke state fields are required");
        validate_step(*settings);
        for (uint32_t e = 0; e < emitter_count; e++) validate_emitter(emitters[e]);
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) fail(F3D_STATUS_DEVICE, "no HIP device available: libf3dhip has no CPU fallback");

        std::lock_guard<std::mutex> workspace_guard(workspace_lock());  // one smoke call at a time enqueues (f3d_devmem.h)
        Sim s;
        s.n = (size_t)n64;
        s.G = SimGrid{st->dims[0], st->dims[1], st->dims[2], st->voxel_size[0], st->voxel_size[1], st->voxel_size[2], st->origin[0], st->origin[1],
                      st->origin[2], st->sparse_threshold, st->time_seconds, st->frame_index};
        s.block = dim3(64);
        s.grid = dim3((s.G.nx + 63u) / 64u, s.G.ny, s.G.nz);
        float **dev[9] = {&s.F.density, &s.F.temperature, &s.F.fuel, &s.F.soot, &s.F.humidity, &s.F.emission_rate, &s.F.particle_age, &s.F.velocity, &s.F.pressure};
        // The nine fields may be DEVICE arrays (all nine, or none): the solver then works on them in place and nothing crosses
        // the bus -- a smoke sequence keeps its state on the GPU (forge3d_amd.smoke.SmokeSequence; BASELINE.json configs[4]).
        int on_device = 0;
        for (int f = 0; f < 9; f++) {
            hipPointerAttribute_t attr{};
            if (hipPointerGetAttributes(&attr, host[f]) == hipSuccess && attr.type == hipMemoryTypeDevice) on_device++;
            (void)hipGetLastError();
        }
        if (on_device != 0 && on_device != 9) fail(F3D_STATUS_VALUE, "the nine smoke state fields must be all host or all device arrays");
        const bool resident = on_device == 9;
        for (int f = 0; f < 9; f++) {
            const size_t bytes = s.n * sizeof(float) * (f == 7 ? 3u : 1u);
            if (resident) {
                *dev[f] = host[f];
                continue;
            }
            static const char *const kFieldTags[9] = {"smoke.sim.f0", "smoke.sim.f1", "smoke.sim.f2", "smoke.sim.f3", "smoke.sim.f4", "smoke.sim.f5", "smoke.sim.f6", "smoke.sim.f7", "smoke.sim.f8"};
            *dev[f] = (float *)s.alloc(bytes, kFieldTags[f]);
            ok(hipStreamSynchronize(call_stream()), "smoke state upload");
            ok(hipMemcpy(*dev[f], host[f], bytes, hipMemcpyHostToDevice), "smoke state upload");
        }
        s.tmp_a = (float *)s.alloc(s.n * sizeof(float), "smoke.sim.tmp_a");
        s.tmp_b = (float *)s.alloc(s.n * sizeof(float), "smoke.sim.tmp_b");
        s.vec_a = (float *)s.alloc(3 * s.n * sizeof(float), "smoke.sim.vec_a");
        s.curl = (float *)s.alloc(3 * s.n * sizeof(float), "smoke.sim.curl");
        s.div = (float *)s.alloc(s.n * sizeof(float), "smoke.sim.div");
        s.rows = (float *)s.alloc(4 * (size_t)s.G.ny * s.G.nz * sizeof(float), "smoke.sim.rows");
        s.slabs = (float *)s.alloc(4 * (size_t)s.G.nz * sizeof(float), "smoke.sim.slabs");
        s.sums = (float *)s.alloc(4 * sizeof(float), "smoke.sim.sums");
        SimSettings S{};
        S.dt = settings->dt; S.density_decay = settings->density_decay; S.temperature_decay = settings->temperature_decay;
        S.velocity_damping = settings->velocity_damping; S.diffusion = settings->diffusion; S.buoyancy = settings->buoyancy;
        S.vorticity = settings->vorticity; S.pressure_iterations = settings->pressure_iterations; S.turbulence_strength = settings->turbulence_strength;
        S.turbulence_seed = settings->turbulence_seed; S.mac_cormack = settings->mac_cormack; S.mass_conservation = settings->mass_conservation;
        S.terrain_collision = settings->terrain_collision; S.boundary_damping = settings->boundary_damping;
        for (int a = 0; a < 3; a++) S.wind[a] = settings->wind[a];

        // A call whose state stays on the device and whose caller does not ask for the device time returns as soon as its
        // launches are enqueued (a resident sequence: the next call's work is behind this call's in the stream).
        const bool timed = device_seconds != nullptr || !resident || !current_smoke_context()->async_ok;  // (handle-less calls on the null stream wait, as in ABI 4)
        ok(hipEventCreate(&e0), "event");
        ok(hipEventCreate(&e1), "event");
        // Which driver (see "the step as PHASES" above): fused launches by default; F3D_SMOKE_SOLVER=persistent / launches
        // select the one cooperative launch and the round-3 launch-per-pass form (both kept as tested A/Bs).
        const char *form = getenv("F3D_SMOKE_SOLVER");
        bool persistent = form && strcmp(form, "persistent") == 0;
        const bool per_pass = form && strcmp(form, "launches") == 0;
        int device = 0, cus = 0, cooperative = 0, per_cu = 0;
        ok(hipGetDevice(&device), "device");
        if (persistent) {
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
            (void)hipDeviceGetAttribute(&cooperative, hipDeviceAttributeCooperativeLaunch, device);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sim_step, (int)kStepBlock, 0) != hipSuccess) per_cu = 0;
            (void)hipGetLastError();
            if (!cooperative || cus <= 0 || per_cu <= 0) fail(F3D_STATUS_DEVICE, "F3D_SMOKE_SOLVER=persistent: this device cannot co-schedule the step kernel");
        }
        if (!per_pass) {
            StepArgs K{};
            K.G = s.G;
            K.F = s.F;
            K.S = S;
            K.emitter_count = emitter_count;
            K.steps = steps;
            static_assert(sizeof(SimEmitter) == sizeof(f3d_smoke_emitter), "emitter layouts differ");
            if (emitter_count <= kInlineEmitters) {
                if (emitter_count) memcpy(K.inline_emitters, emitters, emitter_count * sizeof(SimEmitter));
            } else {
                SimEmitter *d_em = (SimEmitter *)s.alloc(emitter_count * sizeof(SimEmitter), "smoke.sim.emitters");
                ok(hipMemcpyAsync(d_em, emitters, emitter_count * sizeof(SimEmitter), hipMemcpyHostToDevice, call_stream()), "emitters");
                K.emitters = d_em;
            }
            K.vel_b = s.vec_a;
            K.curl = s.curl;
            K.mag = s.tmp_a;
            K.div = s.div;
            K.pres_b = s.tmp_b;
            static const char *const kAdvTags[10] = {"smoke.sim.adv0", "smoke.sim.adv1", "smoke.sim.adv2", "smoke.sim.adv3", "smoke.sim.adv4", "smoke.sim.cor0", "smoke.sim.cor1", "smoke.sim.cor2", "smoke.sim.cor3", "smoke.sim.cor4"};
            for (int f = 0; f < 5; f++) K.adv[f] = (float *)s.alloc(s.n * sizeof(float), kAdvTags[f]);
            if (S.mac_cormack)
                for (int f = 0; f < 5; f++) K.adv2[f] = (float *)s.alloc(s.n * sizeof(float), kAdvTags[5 + f]);
            K.rows = s.rows;
            K.slabs = s.slabs;
            K.sums = s.sums;
            if (persistent) {
#if defined(F3D_SIM_PHASE_TIMES)
                constexpr size_t kBarrierBytes = 4 * sizeof(unsigned int) + 2 * 2000 * sizeof(unsigned long long);
#else
                constexpr size_t kBarrierBytes = 4 * sizeof(unsigned int);
#endif
                K.barrier = (unsigned int *)s.alloc(kBarrierBytes, "smoke.sim.barrier");
                ok(hipMemsetAsync(K.barrier, 0, kBarrierBytes, call_stream()), "barrier clear");
                const uint32_t wanted = (uint32_t)((s.n + kStepBlock - 1u) / kStepBlock);
                const uint32_t blocks = std::min<uint32_t>((uint32_t)cus, std::max<uint32_t>(wanted, 1u));  // one workgroup a CU: co-resident by construction
                void *params[1] = {&K};
                ok(hipEventRecord(e0, call_stream()), "event");
                ok(hipLaunchCooperativeKernel((const void *)k_sim_step, dim3(blocks), dim3(kStepBlock), params, 0u, call_stream()), "cooperative launch of the smoke solver");
                ok(hipEventRecord(e1, call_stream()), "event");
                ok(hipEventSynchronize(e1), "smoke solver");
                unsigned int bar[4] = {};
                ok(hipMemcpy(bar, K.barrier, sizeof(bar), hipMemcpyDeviceToHost), "barrier read-back");
#if defined(F3D_SIM_PHASE_TIMES)
                if (const char *path = getenv("F3D_SIM_PHASE_FILE")) {
                    std::vector<unsigned long long> t(2 * 2000);
                    ok(hipMemcpy(t.data(), K.barrier + 4, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost), "phase times");
                    if (FILE *f = fopen(path, "w")) {
                        for (unsigned int k = 0; k < bar[3] && k < 2000u; k++) fprintf(f, "%u %llu %llu\n", k, t[2 * k], t[2 * k + 1]);
                        fclose(f);
                    }
                }
#endif
                if (bar[2] != 0u) fail(F3D_STATUS_DEVICE, "smoke solver: a grid barrier of the persistent step kernel was not released (%u workgroups)", blocks);
                for (uint32_t step = 0; step < steps; step++) {
                    s.G.time_seconds += S.dt;
                    s.G.frame_index += 1u;
                }
            } else {
                // fused: one launch per phase, in stream order (k_phase); the step's time and frame advance on the host
                const dim3 grid((unsigned)((s.n + kPhaseBlock - 1u) / kPhaseBlock)), block(kPhaseBlock);
                const dim3 row_grid((s.G.ny * s.G.nz + 63u) / 64u);
                PhaseCtx C{};
                // MEASURED, NOT ADOPTED: two Jacobi sweeps a launch (sim_jacobi_twice: the intermediate field formed in registers, 32
                // cached loads a voxel instead of 2 x 8) -- bit-identical, 15 launches fewer a step, and SLOWER: 0.411-0.416 ms
                // against 0.386-0.397 (the doubled sweep costs more than the 5-us launch it saves).  F3D_SMOKE_DOUBLE_SWEEPS=1 runs it.
                const bool single_sweeps = getenv("F3D_SMOKE_DOUBLE_SWEEPS") == nullptr;
                // MEASURED, NOT ADOPTED (round 6): K sweeps a launch on LDS tiles (k_jacobi_tiled) -- bit-identical, 8 launches instead of
                // 30 a step, and SLOWER: 0.45 ms a step against 0.33 (a launch of four sweeps takes 43 us, of two 22 us: ~80 vector
                // instructions per voxel and sweep once the halo's redundancy, the row predicates and the IEEE division are paid by one
                // workgroup a CU, against 5 us for a sweep that the whole chip shares).  F3D_SMOKE_JACOBI=tiled runs it.
                const char *jacobi_form = getenv("F3D_SMOKE_JACOBI");
                const bool jacobi_tiled = single_sweeps && jacobi_form && strcmp(jacobi_form, "tiled") == 0;
                const size_t slab_lds = ((size_t)s.G.nx * (s.G.ny + 1u) + 4u * (size_t)s.G.ny) * sizeof(float);
                const bool slab_sums = slab_lds <= 60u * 1024u && 4u * (size_t)s.G.nz * sizeof(float) <= 60u * 1024u &&
                                       getenv("F3D_SMOKE_ROW_SUMS") == nullptr;  // (else: a lane a row from global memory)
                auto sums = [&](const float *density, const SumKinds &kinds) {
                    if (slab_sums) {
                        hipLaunchKernelGGL(k_phase_sum_slabs, dim3(s.G.nz), dim3(256), slab_lds, call_stream(), K, kinds, density);
                        hipLaunchKernelGGL(k_phase_sum_total, dim3(1), dim3(256), 4u * (size_t)s.G.nz * sizeof(float), call_stream(), K, kinds);
                    } else {
                        hipLaunchKernelGGL(k_phase_sum_rows, row_grid, dim3(64), 0, call_stream(), K, kinds, density);
                        hipLaunchKernelGGL(k_phase_sum_finish, dim3(1), dim3(256), 0, call_stream(), K, kinds);
                    }
                };
                auto project_fused = [&](uint32_t iterations) {
                    hipLaunchKernelGGL(k_phase<kPhDivergence>, grid, block, 0, call_stream(), K, C);
                    C.cur = K.F.pressure;
                    C.next = K.pres_b;
                    uint32_t it = 0;
                    if (jacobi_tiled) {  // K sweeps a launch in LDS (k_jacobi_tiled)
                        const uint32_t tiles_x = (s.G.nx + kJacTX - 1u) / kJacTX, tiles_y = (s.G.ny + kJacTY - 1u) / kJacTY,
                                       tiles_z = (s.G.nz + kJacTZ - 1u) / kJacTZ;
                        while (it < iterations) {
                            const uint32_t k = std::min(kJacK, iterations - it);
                            hipLaunchKernelGGL(k_jacobi_tiled, dim3(tiles_x * tiles_y * tiles_z), dim3(kJacLanesX * kJacRows), 0, call_stream(), K.G,
                                               (const float *)C.cur, (const float *)K.div, C.next, k, tiles_x, tiles_y);
                            std::swap(C.cur, C.next);
                            it += k;
                        }
                    }
                    for (; it + 2u <= iterations && !single_sweeps; it += 2u) {  // two sweeps a launch (sim_jacobi_twice)
                        hipLaunchKernelGGL(k_phase<kPhJacobiTwice>, grid, block, 0, call_stream(), K, C);
                        std::swap(C.cur, C.next);
                    }
                    for (; it < iterations; it++) {
                        hipLaunchKernelGGL(k_phase<kPhJacobi>, grid, block, 0, call_stream(), K, C);
                        std::swap(C.cur, C.next);
                    }
                    hipLaunchKernelGGL(k_phase<kPhGradientBoundary>, grid, block, 0, call_stream(), K, C);
                };
                if (timed) ok(hipEventRecord(e0, call_stream()), "event");
                for (uint32_t step = 0; step < steps; step++) {  // SmokeVolume::step, sim.rs:47-139
                    hipLaunchKernelGGL(k_phase<kPhEmitForces>, grid, block, 0, call_stream(), K, C);
                    hipLaunchKernelGGL(k_phase<kPhAdvectVec>, grid, block, 0, call_stream(), K, C);
                    hipLaunchKernelGGL(k_phase<kPhDiffuseVec>, grid, block, 0, call_stream(), K, C);
                    if (S.vorticity > 0.0f) {
                        hipLaunchKernelGGL(k_phase<kPhCurl>, grid, block, 0, call_stream(), K, C);
                        hipLaunchKernelGGL(k_phase<kPhConfine>, grid, block, 0, call_stream(), K, C);
                    }
                    project_fused(std::max(1u, S.pressure_iterations));
                    SumKinds before{};
                    if (S.turbulence_strength > 0.0f) {
                        before.kind[before.n] = 1u, before.slot[before.n++] = 0u;
                        before.kind[before.n] = 2u, before.slot[before.n++] = 1u;
                        before.kind[before.n] = 3u, before.slot[before.n++] = 2u;
                    }
                    if (S.mass_conservation) before.kind[before.n] = 0u, before.slot[before.n++] = 3u;
                    if (before.n != 0u) sums(K.F.density, before);
                    if (S.turbulence_strength > 0.0f) hipLaunchKernelGGL(k_phase<kPhLaneShear>, grid, block, 0, call_stream(), K, C);
                    hipLaunchKernelGGL(k_phase<kPhAdvectScalars>, grid, block, 0, call_stream(), K, C);
                    C.corrected = 0u;
                    if (S.mac_cormack) {
                        hipLaunchKernelGGL(k_phase<kPhCorrectScalars>, grid, block, 0, call_stream(), K, C);
                        C.corrected = 1u;
                    }
                    if (S.mass_conservation) {
                        SumKinds after{};
                        after.n = 1u;  // kind 0 -> slot 0
                        sums(C.corrected ? K.adv2[0] : K.adv[0], after);
                    }
                    hipLaunchKernelGGL(k_phase<kPhScaleSubgrid>, grid, block, 0, call_stream(), K, C);
                    hipLaunchKernelGGL(k_phase<kPhDiffuseDecay>, grid, block, 0, call_stream(), K, C);
                    project_fused(std::max(1u, S.pressure_iterations / 2u));
                    s.G.time_seconds += S.dt;
                    s.G.frame_index += 1u;
                    K.G = s.G;
                }
                ok(hipGetLastError(), "smoke solver kernels");
                if (timed) {
                    ok(hipEventRecord(e1, call_stream()), "event");
                    ok(hipEventSynchronize(e1), "smoke solver");
                }
            }
        } else {
        ok(hipEventRecord(e0, call_stream()), "event");
        for (uint32_t step = 0; step < steps; step++) {  // SmokeVolume::step, sim.rs:47-139
            ok(hipMemsetAsync(s.F.emission_rate, 0, s.n * sizeof(float), call_stream()), "emission clear");
            for (uint32_t e = 0; e < emitter_count; e++)
                if (s.G.time_seconds >= emitters[e].start_time && s.G.time_seconds <= emitters[e].end_time) {
                    SimKernelArgs A{};
                    A.S = S;
                    static_assert(sizeof(SimEmitter) == sizeof(f3d_smoke_emitter), "emitter layouts differ");
                    memcpy(&A.E, &emitters[e], sizeof(SimEmitter));
                    A.dt = S.dt;
                    s.run<kEmit>(A);
                }
            SimKernelArgs A{};
            A.S = S;
            s.run<kForces>(A);
            // velocity = advect_vector(velocity_before, velocity_before)
            ok(hipMemcpyAsync(s.vec_a, s.F.velocity, 3 * s.n * sizeof(float), hipMemcpyDeviceToDevice, call_stream()), "velocity copy");
            A = SimKernelArgs{};
            A.a = s.vec_a;
            A.b = s.F.velocity;
            A.dt = S.dt;
            s.run<kAdvectVec>(A);
            if (S.diffusion > 0.0f) {  // diffuse_vector, sim.rs:739-754: component by component over a copy
                ok(hipMemcpyAsync(s.vec_a, s.F.velocity, 3 * s.n * sizeof(float), hipMemcpyDeviceToDevice, call_stream()), "velocity copy");
                for (uint32_t c = 0; c < 3u; c++) {
                    A = SimKernelArgs{};
                    A.a = s.vec_a;
                    A.b = s.F.velocity;
                    A.dt = S.diffusion * S.dt;
                    A.u0 = 3u;
                    A.u1 = c;
                    s.run<kDiffuse>(A);
                }
            }
            if (S.vorticity > 0.0f) {
                A = SimKernelArgs{};
                A.S = S;
                A.a = s.curl;
                A.b = s.tmp_a;
                s.run<kCurl>(A);
                s.run<kConfine>(A);
            }
            project(s, std::max(1u, S.pressure_iterations));
            A = SimKernelArgs{};
            A.S = S;
            s.run<kBoundary>(A);
            if (S.turbulence_strength > 0.0f) {
                s.sum({{1u, 0u}, {2u, 1u}, {3u, 2u}});
                s.run<kLaneShear>(A);
            }
            if (S.mass_conservation) s.sum({{0u, 3u}});  // density_mass_before
            advect(s, s.F.density, S);
            if (S.mass_conservation) {
                s.sum({{0u, 0u}});
                s.run<kScale>(A);
            }
            advect(s, s.F.temperature, S);
            advect(s, s.F.fuel, S);
            advect(s, s.F.soot, S);
            advect(s, s.F.humidity, S);
            s.run<kSubgrid>(A);
            if (S.diffusion > 0.0f) {  // apply_scalar_diffusion, sim.rs:236-245
                float *fields[5] = {s.F.density, s.F.temperature, s.F.fuel, s.F.soot, s.F.humidity};
                for (float *f : fields) {
                    diffuse(s, f, S.diffusion * S.dt, 1u, 0u, s.tmp_a);
                    ok(hipMemcpyAsync(f, s.tmp_a, s.n * sizeof(float), hipMemcpyDeviceToDevice, call_stream()), "diffused field");
                }
            }
            s.run<kDecay>(A);
            project(s, std::max(1u, S.pressure_iterations / 2u));
            s.run<kBoundary>(A);
            s.G.time_seconds += S.dt;
            s.G.frame_index += 1u;
        }
        ok(hipEventRecord(e1, call_stream()), "event");
        ok(hipEventSynchronize(e1), "smoke solver");
        }
        ok(hipGetLastError(), "smoke solver kernels");
        float ms = 0.0f;
        if (timed || persistent || per_pass) ok(hipEventElapsedTime(&ms, e0, e1), "event");
        if (device_seconds) *device_seconds = ms * 1e-3;
        if (!resident)
            for (int f = 0; f < 9; f++) ok(hipMemcpy(host[f], *dev[f], s.n * sizeof(float) * (f == 7 ? 3u : 1u), hipMemcpyDeviceToHost), "smoke state read-back");
        st->time_seconds = s.G.time_seconds;
        st->frame_index = s.G.frame_index;
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
