// forge3d_amd/csrc/f3d_smoke_sim.hip -- smoke transport solver on gfx950: one launch per pass of SmokeVolume::step
// (reference src/smoke/sim.rs:47-139), a lane per voxel; per-voxel arithmetic in f3d_smoke_sim.h.  C ABI:
// f3d_smoke_step (include/f3d_terrain_pt.h).  The state lives on the device for all `steps` of a call; BASELINE.json
// configs[4] (a 120-frame sequence) advances it frame by frame and renders each state with f3d_smoke_render.
#include <hip/hip_runtime.h>

#include <cmath>
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

void ok(hipError_t e, const char *what) {
    if (e != hipSuccess) fail(F3D_STATUS_DEVICE, "HIP failure in %s: %s", what, hipGetErrorString(e));
}

struct Sim {
    SimGrid G;
    SimFields F{};
    float *tmp_a = nullptr, *tmp_b = nullptr, *vec_a = nullptr, *curl = nullptr, *div = nullptr, *rows = nullptr, *slabs = nullptr, *sums = nullptr;
    std::vector<void *> owned;
    size_t n = 0;
    dim3 grid, block;
    void *alloc(size_t bytes) {
        void *p = nullptr;
        ok(device_alloc(&p, bytes), "smoke solver allocation");
        owned.push_back(p);
        return p;
    }
    ~Sim() {
        for (void *p : owned) (void)device_free(p);
    }
    template <uint32_t PASS>
    void run(SimKernelArgs A) {
        A.G = G;
        A.F = F;
        A.sums = sums;
        hipLaunchKernelGGL(k_sim<PASS>, grid, block, 0, nullptr, A);
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
        hipLaunchKernelGGL(k_sum_rows, dim3((r + 63u) / 64u), dim3(64), 0, nullptr, A);
        hipLaunchKernelGGL(k_sum_slabs, dim3((G.nz + 63u) / 64u), dim3(64), 0, nullptr, A);
        hipLaunchKernelGGL(k_sum_total, dim3(1), dim3(64), 0, nullptr, A);
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
    ok(hipMemsetAsync(s.F.pressure, 0, s.n * sizeof(float), nullptr), "pressure clear");
    float *cur = s.F.pressure, *next = s.tmp_a;
    for (uint32_t it = 0; it < iterations; it++) {
        A = SimKernelArgs{};
        A.a = cur;
        A.b = next;
        A.c = s.div;
        s.run<kJacobi>(A);
        std::swap(cur, next);
    }
    if (cur != s.F.pressure) ok(hipMemcpyAsync(s.F.pressure, cur, s.n * sizeof(float), hipMemcpyDeviceToDevice, nullptr), "pressure copy");
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
        ok(hipMemcpyAsync(field, s.tmp_b, s.n * sizeof(float), hipMemcpyDeviceToDevice, nullptr), "advected field");
    } else {
        ok(hipMemcpyAsync(field, s.tmp_a, s.n * sizeof(float), hipMemcpyDeviceToDevice, nullptr), "advected field");
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
            *dev[f] = (float *)s.alloc(bytes);
            ok(hipMemcpy(*dev[f], host[f], bytes, hipMemcpyHostToDevice), "smoke state upload");
        }
        s.tmp_a = (float *)s.alloc(s.n * sizeof(float));
        s.tmp_b = (float *)s.alloc(s.n * sizeof(float));
        s.vec_a = (float *)s.alloc(3 * s.n * sizeof(float));
        s.curl = (float *)s.alloc(3 * s.n * sizeof(float));
        s.div = (float *)s.alloc(s.n * sizeof(float));
        s.rows = (float *)s.alloc(4 * (size_t)s.G.ny * s.G.nz * sizeof(float));
        s.slabs = (float *)s.alloc(4 * (size_t)s.G.nz * sizeof(float));
        s.sums = (float *)s.alloc(4 * sizeof(float));
        SimSettings S{};
        S.dt = settings->dt; S.density_decay = settings->density_decay; S.temperature_decay = settings->temperature_decay;
        S.velocity_damping = settings->velocity_damping; S.diffusion = settings->diffusion; S.buoyancy = settings->buoyancy;
        S.vorticity = settings->vorticity; S.pressure_iterations = settings->pressure_iterations; S.turbulence_strength = settings->turbulence_strength;
        S.turbulence_seed = settings->turbulence_seed; S.mac_cormack = settings->mac_cormack; S.mass_conservation = settings->mass_conservation;
        S.terrain_collision = settings->terrain_collision; S.boundary_damping = settings->boundary_damping;
        for (int a = 0; a < 3; a++) S.wind[a] = settings->wind[a];

        ok(hipEventCreate(&e0), "event");
        ok(hipEventCreate(&e1), "event");
        ok(hipEventRecord(e0, nullptr), "event");
        for (uint32_t step = 0; step < steps; step++) {  // SmokeVolume::step, sim.rs:47-139
            ok(hipMemsetAsync(s.F.emission_rate, 0, s.n * sizeof(float), nullptr), "emission clear");
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
            ok(hipMemcpyAsync(s.vec_a, s.F.velocity, 3 * s.n * sizeof(float), hipMemcpyDeviceToDevice, nullptr), "velocity copy");
            A = SimKernelArgs{};
            A.a = s.vec_a;
            A.b = s.F.velocity;
            A.dt = S.dt;
            s.run<kAdvectVec>(A);
            if (S.diffusion > 0.0f) {  // diffuse_vector, sim.rs:739-754: component by component over a copy
                ok(hipMemcpyAsync(s.vec_a, s.F.velocity, 3 * s.n * sizeof(float), hipMemcpyDeviceToDevice, nullptr), "velocity copy");
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
                    ok(hipMemcpyAsync(f, s.tmp_a, s.n * sizeof(float), hipMemcpyDeviceToDevice, nullptr), "diffused field");
                }
            }
            s.run<kDecay>(A);
            project(s, std::max(1u, S.pressure_iterations / 2u));
            s.run<kBoundary>(A);
            s.G.time_seconds += S.dt;
            s.G.frame_index += 1u;
        }
        ok(hipEventRecord(e1, nullptr), "event");
        ok(hipEventSynchronize(e1), "smoke solver");
        ok(hipGetLastError(), "smoke solver kernels");
        float ms = 0.0f;
        ok(hipEventElapsedTime(&ms, e0, e1), "event");
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
