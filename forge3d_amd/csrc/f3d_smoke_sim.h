// forge3d_amd/csrc/f3d_smoke_sim.h -- per-voxel arithmetic of the smoke transport solver (host + device).
//
// Reference: SmokeVolume::step and its helpers (src/smoke/sim.rs:47-799, sampling.rs:1-94), single-threaded CPU Rust:
// every pass is a triple loop over the grid.  Here every pass is a function of ONE voxel that reads the fields of the
// previous pass and writes its own entry, so a pass is one launch with a lane per voxel (f3d_smoke.hip) -- the passes
// that the reference runs in place over a clone (diffusion, the Jacobi sweeps, the advections) read a `src` and write
// a `dst` buffer.  The three grid sums of a step (mass before / after the density advection, the smoke's centroid) are
// taken row by row, then over the rows of a slab, then over the slabs (sum_rows / sum_slabs / sum_total): a fixed
// order that ny * nz lanes can follow; oracle/smoke_sim_oracle.c sums the same way, so device == oracle bit for bit.
// sin / cos / exp are the fixed polynomials of f3d_math.h.  The test emulator compiles this header for the host.
#pragma once

#include "f3d_math.h"

namespace f3d {
namespace smoke {

struct SimSettings {  // SmokeStepSettings, types.rs:142-158
    float dt, density_decay, temperature_decay, velocity_damping, diffusion, buoyancy, vorticity;
    uint32_t pressure_iterations;
    float turbulence_strength;
    uint32_t turbulence_seed;
    int32_t mac_cormack, mass_conservation, terrain_collision;
    float boundary_damping;
    float wind[3];
};
struct SimEmitter {  // SmokeEmitter, types.rs:69-81
    float center[3], radius, density_rate, temperature_rate, fuel_rate, soot_rate, humidity_rate, emission_rate, velocity[3], start_time,
        end_time;
};
struct SimGrid {
    uint32_t nx, ny, nz;
    float vx, vy, vz;  // voxel size
    float ox, oy, oz;  // origin
    float sparse_threshold, time_seconds;
    uint32_t frame_index;
};

F3D_HD size_t sim_index(const SimGrid &G, uint32_t x, uint32_t y, uint32_t z) { return ((size_t)z * G.ny + y) * G.nx + x; }
F3D_HD float sim_lerp(float a, float b, float t) { return a + (b - a) * t; }
F3D_HD float sim_smoothstep(float e0, float e1, float x) {
    const float t = f_clamp((x - e0) / f_max(e1 - e0, 1.0e-6f), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
F3D_HD void sim_sincos(float a, float &s, float &c) {  // the angle in turns, then sincos_turn (as f3d_wf_path.h sincos_rad)
    float u = a * 0.15915494309189533577f;
    u = u - f_floor(u);
    sincos_turn(u, s, c);
}
F3D_HD float sim_sin(float a) {
    float s, c;
    sim_sincos(a, s, c);
    return s;
}
F3D_HD float sim_cos(float a) {
    float s, c;
    sim_sincos(a, s, c);
    return c;
}
F3D_HD float sim_axis_unit(uint32_t i, uint32_t n) {
    const uint32_t m = n > 0u ? n - 1u : 0u;
    return (float)i / (float)(m > 1u ? m : 1u);
}
// sample_scalar / sample_vector_component (sampling.rs:1-81): stride 1 / 3
F3D_HD float sim_sample(const SimGrid &G, const float *f, float px, float py, float pz, uint32_t stride, uint32_t comp) {
    const float x = f_clamp(px, 0.0f, (float)(G.nx - 1u)), y = f_clamp(py, 0.0f, (float)(G.ny - 1u)), z = f_clamp(pz, 0.0f, (float)(G.nz - 1u));
    const uint32_t x0 = (uint32_t)f_floor(x), y0 = (uint32_t)f_floor(y), z0 = (uint32_t)f_floor(z);
    const uint32_t x1 = x0 + 1u < G.nx - 1u ? x0 + 1u : G.nx - 1u, y1 = y0 + 1u < G.ny - 1u ? y0 + 1u : G.ny - 1u,
                   z1 = z0 + 1u < G.nz - 1u ? z0 + 1u : G.nz - 1u;
    const float fx = x - (float)x0, fy = y - (float)y0, fz = z - (float)z0;
    auto rd = [&](uint32_t X, uint32_t Y, uint32_t Z) F3D_LAMBDA { return f[sim_index(G, X, Y, Z) * stride + comp]; };
    const float c00 = sim_lerp(rd(x0, y0, z0), rd(x1, y0, z0), fx), c10 = sim_lerp(rd(x0, y1, z0), rd(x1, y1, z0), fx);
    const float c01 = sim_lerp(rd(x0, y0, z1), rd(x1, y0, z1), fx), c11 = sim_lerp(rd(x0, y1, z1), rd(x1, y1, z1), fx);
    return sim_lerp(sim_lerp(c00, c10, fy), sim_lerp(c01, c11, fy), fz);
}

// add_emitter, sim.rs:7-45
struct SimFields {
    float *density, *temperature, *fuel, *soot, *humidity, *emission_rate, *particle_age, *velocity, *pressure;
};
F3D_HD void sim_emit(const SimGrid &G, const SimFields &F, const SimEmitter &E, float dt, uint32_t x, uint32_t y, uint32_t z) {
    const size_t i = sim_index(G, x, y, z);
    const float radius = f_max(E.radius, 1.0e-6f);
    const float px = G.ox + ((float)x + 0.5f) * G.vx, py = G.oy + ((float)y + 0.5f) * G.vy, pz = G.oz + ((float)z + 0.5f) * G.vz;
    const float dx = px - E.center[0], dy = py - E.center[1], dz = pz - E.center[2];
    const float dist = f_sqrt((dx * dx + dy * dy) + dz * dz);
    if (dist > radius) return;
    const float falloff = 1.0f - sim_smoothstep(0.0f, radius, dist), amount = dt * falloff;
    F.density[i] = f_max(F.density[i] + E.density_rate * amount, 0.0f);
    F.temperature[i] = f_max(F.temperature[i] + E.temperature_rate * amount, 0.0f);
    F.fuel[i] = f_max(F.fuel[i] + E.fuel_rate * amount, 0.0f);
    F.soot[i] = f_max(F.soot[i] + E.soot_rate * amount, 0.0f);
    F.humidity[i] = f_max(F.humidity[i] + E.humidity_rate * amount, 0.0f);
    F.emission_rate[i] += E.emission_rate * falloff;
    F.particle_age[i] = 0.0f;
    for (uint32_t c = 0u; c < 3u; c++) F.velocity[3u * i + c] += E.velocity[c] * amount;
}

// apply_forces, sim.rs:162-234
F3D_HD void sim_forces(const SimGrid &G, const SimFields &F, const SimSettings &S, uint32_t x, uint32_t y, uint32_t z) {
    const size_t i = sim_index(G, x, y, z);
    float *v = F.velocity + 3u * i;
    float v0 = v[0], v1 = v[1], v2 = v[2];
    v0 += S.wind[0] * S.dt;
    v1 += (S.wind[1] + F.temperature[i] * S.buoyancy) * S.dt;
    v2 += S.wind[2] * S.dt;
    if (S.velocity_damping > 0.0f) {
        const float damping = exp_det(-S.velocity_damping * S.dt);
        v0 *= damping;
        v1 *= damping;
        v2 *= damping;
    }
    if (S.turbulence_strength > 0.0f) {
        const float xf = sim_axis_unit(x, G.nx), yf = sim_axis_unit(y, G.ny), zf = sim_axis_unit(z, G.nz);
        const float seed_phase = (float)S.turbulence_seed * 0.000137f, t = G.time_seconds;
        const float amp = S.turbulence_strength * S.dt;
        const float altitude_gain = f_clamp(0.45f + 0.75f * yf, 0.35f, 1.20f);
        const float lane_a = sim_sin(xf * 9.6f + zf * 4.2f + yf * 1.6f + t * 0.52f + seed_phase);
        const float lane_b = sim_cos(zf * 7.4f - xf * 5.1f + yf * 2.7f - t * 0.37f + seed_phase * 1.7f);
        const float roll = sim_sin((xf + zf) * 3.9f - yf * 5.2f + t * 0.29f + seed_phase * 0.6f);
        v0 += (0.62f * lane_a + 0.28f * roll) * amp * altitude_gain;
        v1 += (0.08f * lane_b - 0.05f * roll) * amp;
        v2 += (-0.56f * lane_b + 0.26f * lane_a) * amp * altitude_gain;
        const float wind_len = f_sqrt(S.wind[0] * S.wind[0] + S.wind[2] * S.wind[2]);
        if (wind_len > 1.0e-6f) {
            const float wind_x = S.wind[0] / wind_len, wind_z = S.wind[2] / wind_len, cross_x = -wind_z, cross_z = wind_x;
            const float along = (float)x * wind_x + (float)z * wind_z, cross_coord = (float)x * cross_x + (float)z * cross_z;
            const float lane_phase = along * 0.34f + cross_coord * 0.72f + t * 0.34f + seed_phase * 11.0f;
            const float lane_force = sim_sin(lane_phase) + 0.45f * sim_sin(lane_phase * 0.53f + (float)z * 0.29f);
            const float speed_lane = 0.5f + 0.5f * sim_cos(lane_phase * 0.41f + (float)x * 0.18f);
            v0 += cross_x * lane_force * amp * 0.82f * altitude_gain + wind_x * speed_lane * amp * 0.30f * altitude_gain;
            v2 += cross_z * lane_force * amp * 0.82f * altitude_gain + wind_z * speed_lane * amp * 0.30f * altitude_gain;
            const float shear = (yf - 0.42f) * amp * 1.35f;
            v0 += cross_x * shear;
            v2 += cross_z * shear;
        }
    }
    v[0] = v0;
    v[1] = v1;
    v[2] = v2;
}

// backtraced position of voxel (x, y, z) in the velocity field `vel` (backtrace, sim.rs:659-665)
F3D_HD void sim_back(const SimGrid &G, const float *vel, float dt, uint32_t x, uint32_t y, uint32_t z, float &bx, float &by, float &bz) {
    const float px = (float)x, py = (float)y, pz = (float)z;
    bx = px - sim_sample(G, vel, px, py, pz, 3u, 0u) * dt / G.vx;
    by = py - sim_sample(G, vel, px, py, pz, 3u, 1u) * dt / G.vy;
    bz = pz - sim_sample(G, vel, px, py, pz, 3u, 2u) * dt / G.vz;
}
// advect_vector, sim.rs:638-657
F3D_HD void sim_advect_vector(const SimGrid &G, const float *old, float *dst, float dt, uint32_t x, uint32_t y, uint32_t z) {
    float bx, by, bz;
    sim_back(G, old, dt, x, y, z, bx, by, bz);
    const size_t i = sim_index(G, x, y, z);
    for (uint32_t c = 0u; c < 3u; c++) dst[3u * i + c] = sim_sample(G, old, bx, by, bz, 3u, c);
}
// advect_scalar, first pass (sim.rs:603-613) and the MacCormack correction (:619-634)
F3D_HD void sim_advect_predict(const SimGrid &G, const float *old, const float *vel, float *pred, float dt, uint32_t x, uint32_t y, uint32_t z) {
    float bx, by, bz;
    sim_back(G, vel, dt, x, y, z, bx, by, bz);
    pred[sim_index(G, x, y, z)] = f_max(sim_sample(G, old, bx, by, bz, 1u, 0u), 0.0f);
}
F3D_HD void sim_advect_correct(const SimGrid &G, const float *old, const float *vel, const float *pred, float *dst, float dt, uint32_t x,
                               uint32_t y, uint32_t z) {
    float bx, by, bz;
    sim_back(G, vel, dt, x, y, z, bx, by, bz);
    const float fx = bx + sim_sample(G, vel, bx, by, bz, 3u, 0u) * dt / G.vx, fy = by + sim_sample(G, vel, bx, by, bz, 3u, 1u) * dt / G.vy,
                fz = bz + sim_sample(G, vel, bx, by, bz, 3u, 2u) * dt / G.vz;
    const size_t i = sim_index(G, x, y, z);
    const float recovered = sim_sample(G, pred, fx, fy, fz, 1u, 0u);
    const float candidate = pred[i] + 0.5f * (old[i] - recovered);
    // local_min_max, sim.rs:675-697
    const uint32_t x0 = (uint32_t)f_clamp(f_floor(bx), 0.0f, (float)(G.nx - 1u)), y0 = (uint32_t)f_clamp(f_floor(by), 0.0f, (float)(G.ny - 1u)),
                   z0 = (uint32_t)f_clamp(f_floor(bz), 0.0f, (float)(G.nz - 1u));
    const uint32_t x1 = x0 + 1u < G.nx - 1u ? x0 + 1u : G.nx - 1u, y1 = y0 + 1u < G.ny - 1u ? y0 + 1u : G.ny - 1u,
                   z1 = z0 + 1u < G.nz - 1u ? z0 + 1u : G.nz - 1u;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (uint32_t zz = z0; zz <= z1; zz++)
        for (uint32_t yy = y0; yy <= y1; yy++)
            for (uint32_t xx = x0; xx <= x1; xx++) {
                const float v = old[sim_index(G, xx, yy, zz)];
                lo = f_min(lo, v);
                hi = f_max(hi, v);
            }
    dst[i] = f_max(f_clamp(candidate, lo, hi), 0.0f);
}
F3D_HD bool sim_interior(const SimGrid &G, uint32_t x, uint32_t y, uint32_t z, uint32_t m) {
    return x >= m && y >= m && z >= m && x + m < G.nx && y + m < G.ny && z + m < G.nz;
}
// diffuse_scalar_in_place, sim.rs:713-737: dst = the diffused interior, the border copied
F3D_HD void sim_diffuse(const SimGrid &G, const float *src, float *dst, float alpha, uint32_t stride, uint32_t comp, uint32_t x, uint32_t y,
                        uint32_t z) {
    const size_t i = sim_index(G, x, y, z);
    auto rd = [&](uint32_t X, uint32_t Y, uint32_t Z) F3D_LAMBDA { return src[sim_index(G, X, Y, Z) * stride + comp]; };
    float out = src[i * stride + comp];
    if (sim_interior(G, x, y, z, 1u)) {
        const float sum = rd(x - 1u, y, z) + rd(x + 1u, y, z) + rd(x, y - 1u, z) + rd(x, y + 1u, z) + rd(x, y, z - 1u) + rd(x, y, z + 1u);
        out = (out + alpha * sum) / (1.0f + 6.0f * alpha);
    }
    dst[i * stride + comp] = out;
}
// compute_divergence, sim.rs:756-777
F3D_HD void sim_divergence(const SimGrid &G, const float *v, float *div, uint32_t x, uint32_t y, uint32_t z) {
    float out = 0.0f;
    if (sim_interior(G, x, y, z, 1u)) {
        const float du = (v[sim_index(G, x + 1u, y, z) * 3u] - v[sim_index(G, x - 1u, y, z) * 3u]) / (2.0f * G.vx);
        const float dv = (v[sim_index(G, x, y + 1u, z) * 3u + 1u] - v[sim_index(G, x, y - 1u, z) * 3u + 1u]) / (2.0f * G.vy);
        const float dw = (v[sim_index(G, x, y, z + 1u) * 3u + 2u] - v[sim_index(G, x, y, z - 1u) * 3u + 2u]) / (2.0f * G.vz);
        out = du + dv + dw;
    }
    div[sim_index(G, x, y, z)] = out;
}
// one Jacobi sweep of project (sim.rs:276-292): interior from `cur`, the border stays 0
F3D_HD void sim_jacobi(const SimGrid &G, const float *cur, const float *div, float *next, uint32_t x, uint32_t y, uint32_t z) {
    const size_t i = sim_index(G, x, y, z);
    float out = 0.0f;
    if (sim_interior(G, x, y, z, 1u)) {
        const float sum = cur[sim_index(G, x - 1u, y, z)] + cur[sim_index(G, x + 1u, y, z)] + cur[sim_index(G, x, y - 1u, z)] +
                          cur[sim_index(G, x, y + 1u, z)] + cur[sim_index(G, x, y, z - 1u)] + cur[sim_index(G, x, y, z + 1u)];
        out = (sum - div[i]) / 6.0f;
    }
    next[i] = out;
}
// TWO sweeps in one pass (round 5): next2[i] from `cur` without the intermediate field ever being stored -- each of the six
// neighbours' first-sweep values is formed here, by sim_jacobi's own expression (0 on the border), and the second sweep
// adds them in sim_jacobi's order.  The same operations on the same values as two launches of the sweep: bit-identical;
// 32 cached loads instead of 2 x 8 and one launch (5 us on this chip for a 786 432-voxel field) fewer per pair -- measured
// slower than two launches (f3d_smoke_sim.hip: opt-in), kept under test.
F3D_HD float sim_jacobi_value(const SimGrid &G, const float *cur, const float *div, uint32_t x, uint32_t y, uint32_t z) {
    if (!sim_interior(G, x, y, z, 1u)) return 0.0f;
    const float sum = cur[sim_index(G, x - 1u, y, z)] + cur[sim_index(G, x + 1u, y, z)] + cur[sim_index(G, x, y - 1u, z)] +
                      cur[sim_index(G, x, y + 1u, z)] + cur[sim_index(G, x, y, z - 1u)] + cur[sim_index(G, x, y, z + 1u)];
    return (sum - div[sim_index(G, x, y, z)]) / 6.0f;
}
F3D_HD void sim_jacobi_twice(const SimGrid &G, const float *cur, const float *div, float *next2, uint32_t x, uint32_t y, uint32_t z) {
    const size_t i = sim_index(G, x, y, z);
    float out = 0.0f;
    if (sim_interior(G, x, y, z, 1u)) {
        const float sum = sim_jacobi_value(G, cur, div, x - 1u, y, z) + sim_jacobi_value(G, cur, div, x + 1u, y, z) +
                          sim_jacobi_value(G, cur, div, x, y - 1u, z) + sim_jacobi_value(G, cur, div, x, y + 1u, z) +
                          sim_jacobi_value(G, cur, div, x, y, z - 1u) + sim_jacobi_value(G, cur, div, x, y, z + 1u);
        out = (sum - div[i]) / 6.0f;
    }
    next2[i] = out;
}
// the gradient subtraction of project (sim.rs:294-316)
F3D_HD void sim_subtract_gradient(const SimGrid &G, const float *P, float *vel, uint32_t x, uint32_t y, uint32_t z) {
    if (!sim_interior(G, x, y, z, 1u)) return;
    const size_t vi = 3u * sim_index(G, x, y, z);
    vel[vi] -= (P[sim_index(G, x + 1u, y, z)] - P[sim_index(G, x - 1u, y, z)]) / (2.0f * G.vx);
    vel[vi + 1u] -= (P[sim_index(G, x, y + 1u, z)] - P[sim_index(G, x, y - 1u, z)]) / (2.0f * G.vy);
    vel[vi + 2u] -= (P[sim_index(G, x, y, z + 1u)] - P[sim_index(G, x, y, z - 1u)]) / (2.0f * G.vz);
}
// curl + magnitude (sim.rs:553-566, curl_at :779-799): zero on the border
F3D_HD void sim_curl(const SimGrid &G, const float *v, float *curl, float *mag, uint32_t x, uint32_t y, uint32_t z) {
    const size_t i = sim_index(G, x, y, z);
    float cx = 0.0f, cy = 0.0f, cz = 0.0f, m = 0.0f;
    if (sim_interior(G, x, y, z, 1u)) {
        auto rv = [&](uint32_t X, uint32_t Y, uint32_t Z, uint32_t C) F3D_LAMBDA { return v[sim_index(G, X, Y, Z) * 3u + C]; };
        const float dw_dy = (rv(x, y + 1u, z, 2u) - rv(x, y - 1u, z, 2u)) / (2.0f * G.vy), dv_dz = (rv(x, y, z + 1u, 1u) - rv(x, y, z - 1u, 1u)) / (2.0f * G.vz);
        const float du_dz = (rv(x, y, z + 1u, 0u) - rv(x, y, z - 1u, 0u)) / (2.0f * G.vz), dw_dx = (rv(x + 1u, y, z, 2u) - rv(x - 1u, y, z, 2u)) / (2.0f * G.vx);
        const float dv_dx = (rv(x + 1u, y, z, 1u) - rv(x - 1u, y, z, 1u)) / (2.0f * G.vx), du_dy = (rv(x, y + 1u, z, 0u) - rv(x, y - 1u, z, 0u)) / (2.0f * G.vy);
        cx = dw_dy - dv_dz;
        cy = du_dz - dw_dx;
        cz = dv_dx - du_dy;
        m = f_sqrt((cx * cx + cy * cy) + cz * cz);
    }
    curl[3u * i] = cx;
    curl[3u * i + 1u] = cy;
    curl[3u * i + 2u] = cz;
    mag[i] = m;
}
// the confinement force (sim.rs:567-590)
F3D_HD void sim_confine(const SimGrid &G, const float *curl, const float *mag, float *vel, float strength, float dt, uint32_t x, uint32_t y,
                        uint32_t z) {
    if (!sim_interior(G, x, y, z, 2u)) return;
    const size_t i = sim_index(G, x, y, z);
    const float gx = mag[sim_index(G, x + 1u, y, z)] - mag[sim_index(G, x - 1u, y, z)], gy = mag[sim_index(G, x, y + 1u, z)] - mag[sim_index(G, x, y - 1u, z)],
                gz = mag[sim_index(G, x, y, z + 1u)] - mag[sim_index(G, x, y, z - 1u)];
    const float len2 = (gx * gx + gy * gy) + gz * gz;
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;
    if (len2 > 1.0e-12f) {
        const float inv = 1.0f / f_sqrt(len2);
        nx = gx * inv;
        ny = gy * inv;
        nz = gz * inv;
    }
    const float cx = curl[3u * i], cy = curl[3u * i + 1u], cz = curl[3u * i + 2u];
    vel[3u * i] += ((ny * cz - cy * nz) * strength) * dt;
    vel[3u * i + 1u] += ((nz * cx - cz * nx) * strength) * dt;
    vel[3u * i + 2u] += ((nx * cy - cx * ny) * strength) * dt;
}
// apply_boundary_conditions, sim.rs:520-551
F3D_HD void sim_boundary(const SimGrid &G, const SimFields &F, const SimSettings &S, uint32_t x, uint32_t y, uint32_t z) {
    const size_t i = sim_index(G, x, y, z), vi = 3u * i;
    const float keep = 1.0f - S.boundary_damping;
    if (x == 0u || x == G.nx - 1u) {
        F.velocity[vi] = 0.0f;
        F.density[i] *= 0.58f;
        F.temperature[i] *= 0.70f;
    } else if (x == 1u || x == G.nx - 2u) {
        F.density[i] *= 0.78f;
        F.temperature[i] *= 0.86f;
    }
    if (y == 0u || y == G.ny - 1u) F.velocity[vi + 1u] = 0.0f;
    if (z == 0u || z == G.nz - 1u) {
        F.velocity[vi + 2u] = 0.0f;
        F.density[i] *= 0.58f;
        F.temperature[i] *= 0.70f;
    } else if (z == 1u || z == G.nz - 2u) {
        F.density[i] *= 0.78f;
        F.temperature[i] *= 0.86f;
    }
    if (S.terrain_collision != 0 && y == 0u) {
        F.density[i] *= keep;
        F.temperature[i] *= keep;
    }
}
// apply_lane_advection_shear, sim.rs:319-423 (sums[0..2] = total mass, sum x mass, sum z mass)
F3D_HD void sim_lane_shear(const SimGrid &G, const SimFields &F, const SimSettings &S, const float *sums, uint32_t x, uint32_t y, uint32_t z) {
    const float wind_len = f_sqrt(S.wind[0] * S.wind[0] + S.wind[2] * S.wind[2]);
    if (S.turbulence_strength <= 0.0f || wind_len <= 1.0e-6f) return;
    const size_t i = sim_index(G, x, y, z), vi = 3u * i;
    const float active = sim_smoothstep(G.sparse_threshold, f_max(G.sparse_threshold * 60.0f, 0.012f), F.density[i]);
    if (active <= 0.0f) return;
    const float wind_x = S.wind[0] / wind_len, wind_z = S.wind[2] / wind_len, cross_x = -wind_z, cross_z = wind_x;
    const float amp = S.turbulence_strength * S.dt, seed_phase = (float)S.turbulence_seed * 0.0027f, fi = (float)G.frame_index;
    const float total_mass = sums[0];
    float centroid_x = sums[1], centroid_z = sums[2];
    if (total_mass > 1.0e-6f) {
        centroid_x /= total_mass;
        centroid_z /= total_mass;
    }
    float v0 = F.velocity[vi], v2 = F.velocity[vi + 2u];
    const float along = (float)x * wind_x + (float)z * wind_z, cross_coord = (float)x * cross_x + (float)z * cross_z;
    const float lane_phase = along * 0.23f + cross_coord * 0.49f + fi * 0.105f + seed_phase;
    const float lane_force = sim_sin(lane_phase) + 0.58f * sim_sin(lane_phase * 0.41f + (float)y * 0.74f);
    const float altitude = sim_axis_unit(y, G.ny);
    const float altitude_shear = (altitude - 0.44f) * 0.95f;
    const float force = (lane_force * 2.75f + altitude_shear * 1.45f) * active * amp;
    v0 += cross_x * force;
    v2 += cross_z * force;
    const float slab_phase = along * 0.17f - cross_coord * 0.31f + (float)y * 1.12f + fi * 0.043f + (float)S.turbulence_seed * 0.0021f;
    const float slab_lane = sim_sin(slab_phase) + 0.42f * sim_sin(slab_phase * 0.53f + along * 0.09f);
    const float slab_split = ((altitude - 0.50f) * 2.55f + slab_lane * 0.58f) * active * amp;
    v0 += cross_x * slab_split * 1.90f;
    v2 += cross_z * slab_split * 1.90f;
    const float speed_split = sim_sin(slab_phase * 0.39f + (float)y * 0.67f) * active * amp;
    v0 += wind_x * speed_split * 0.52f;
    v2 += wind_z * speed_split * 0.52f;
    if (total_mass > 1.0e-6f) {
        const float eddies[4][4] = {{5.5f, 5.4f, 1.0f, 1.85f}, {11.5f, 7.6f, -1.0f, 1.58f}, {19.0f, 10.2f, 1.0f, 1.30f}, {28.0f, 13.0f, -1.0f, 1.05f}};
        const float altitude_gain = 0.55f + 0.75f * altitude;
        for (uint32_t e = 0u; e < 4u; e++) {
            const float distance = eddies[e][0], radius = eddies[e][1], side = eddies[e][2], strength = eddies[e][3];
            const float phase = fi * (0.035f + (float)e * 0.006f) + (float)S.turbulence_seed * 0.0013f;
            const float center_x = centroid_x + wind_x * distance + cross_x * side * radius * (0.40f + 0.20f * sim_sin(phase));
            const float center_z = centroid_z + wind_z * distance + cross_z * side * radius * (0.40f + 0.20f * sim_cos(phase));
            const float dx = (float)x - center_x, dz = (float)z - center_z, r2 = dx * dx + dz * dz;
            const float envelope = exp_det(-r2 / (2.0f * radius * radius)) * active;
            const float inv_r = 1.0f / f_sqrt(r2 + 1.0f);
            const float spin = side * strength * amp * envelope * altitude_gain;
            v0 += -dz * inv_r * spin;
            v2 += dx * inv_r * spin;
        }
    }
    F.velocity[vi] = v0;
    F.velocity[vi + 2u] = v2;
}
// apply_subgrid_density_eddies, sim.rs:425-518
F3D_HD void sim_subgrid(const SimGrid &G, const SimFields &F, const SimSettings &S, uint32_t x, uint32_t y, uint32_t z) {
    if (S.turbulence_strength <= 0.0f) return;
    const size_t i = sim_index(G, x, y, z);
    const float active = sim_smoothstep(G.sparse_threshold, f_max(G.sparse_threshold * 90.0f, 0.018f), F.density[i]);
    if (active <= 0.0f) return;
    const float wind_len = f_max(f_sqrt(S.wind[0] * S.wind[0] + S.wind[2] * S.wind[2]), 1.0e-6f);
    const float wind_x = S.wind[0] / wind_len, wind_z = S.wind[2] / wind_len, cross_x = -wind_z, cross_z = wind_x;
    const float seed_phase = (float)S.turbulence_seed * 0.0019f, t = (float)G.frame_index * 0.046f;
    const float xf = (float)x, yf = (float)y, zf = (float)z;
    const float wob = sim_sin(xf * 0.043f + zf * 0.071f + t + seed_phase);
    const float phase = xf * 0.18f + zf * 0.27f + yf * 0.72f + wob * 1.7f + t + seed_phase;
    const float ribbons = 0.5f + 0.5f * sim_sin(phase);
    const float sheets = 0.5f + 0.5f * sim_sin(phase * 0.47f - zf * 0.16f + yf * 0.51f);
    const float voids = sim_smoothstep(0.45f, 0.84f, 1.0f - ribbons) * sim_smoothstep(0.34f, 0.76f, 1.0f - sheets) * active;
    const float ridges = sim_smoothstep(0.62f, 0.94f, ribbons) * sim_smoothstep(0.48f, 0.90f, sheets) * active;
    const float age_t = sim_smoothstep(2.0f, 28.0f, f_max(F.particle_age[i], 0.0f));
    const float void_strength = 0.62f + 0.32f * age_t, ridge_strength = 0.075f - 0.045f * age_t;
    const float along = xf * wind_x + zf * wind_z, cross_coord = xf * cross_x + zf * cross_z;
    const float broad = 0.5f + 0.5f * wob;
    const float channel_phase = along * 0.115f + cross_coord * 0.52f + broad * 5.4f + sim_sin(yf * 0.62f + along * 0.035f) * 0.85f +
                                (float)G.frame_index * 0.033f + (float)S.turbulence_seed * 0.0023f;
    const float channel_wave = 0.5f + 0.5f * sim_sin(channel_phase) + 0.28f * sim_sin(channel_phase * 0.47f - cross_coord * 0.19f + yf * 0.34f);
    const float entrainment = sim_smoothstep(0.58f, 1.06f, channel_wave);
    const float lateral_slots = sim_smoothstep(0.50f, 0.94f, 1.0f - (0.62f * ribbons + 0.38f * sheets));
    const float core_protect = 1.0f - 0.56f * sim_smoothstep(0.72f, 1.75f, F.density[i]);
    const float aged_sheet = (0.28f + 0.72f * age_t) * active * core_protect;
    const float clear_air = f_clamp(entrainment * (0.54f + 0.46f * lateral_slots) * aged_sheet, 0.0f, 1.0f);
    const float channel_void = f_clamp(sim_smoothstep(0.42f, 0.86f, 1.0f - channel_wave) * (0.55f + 0.45f * lateral_slots) * active *
                                           (0.42f + 0.58f * age_t) * core_protect, 0.0f, 1.0f);
    const float gain = (1.0f - void_strength * voids + ridge_strength * ridges) * (1.0f - (0.024f + 0.055f * age_t) * clear_air) *
                       (1.0f - (0.045f + 0.070f * age_t) * channel_void);
    F.density[i] = f_clamp(F.density[i] * gain, 0.0f, 8.0f);
    F.humidity[i] = f_max(F.humidity[i] * (1.0f - (0.15f + 0.10f * age_t) * voids - (0.024f + 0.055f * age_t) * clear_air -
                                           (0.045f + 0.070f * age_t) * channel_void), 0.0f);
}
// apply_decay_and_age, sim.rs:247-268
F3D_HD void sim_decay(const SimGrid &G, const SimFields &F, const SimSettings &S, size_t i) {
    const float temperature_decay = exp_det(-S.temperature_decay * S.dt);
    const float age_t = sim_smoothstep(7.0f, 36.0f, f_max(F.particle_age[i], 0.0f));
    const float density_decay = exp_det(-S.density_decay * S.dt * (1.0f + 3.0f * age_t));
    const float soot_decay = exp_det(-S.density_decay * S.dt * (0.42f + 1.15f * age_t));
    F.density[i] *= density_decay;
    F.temperature[i] *= temperature_decay;
    F.fuel[i] *= density_decay;
    F.soot[i] *= soot_decay;
    if (F.density[i] > G.sparse_threshold) F.particle_age[i] = F.particle_age[i] < 0.0f ? 0.0f : F.particle_age[i] + S.dt;
    else F.particle_age[i] = -1.0f;
}

// ---- the grid sums: rows (one lane per row), the rows of a slab (one lane per slab), the slabs (one lane) ----------
// kind 0: density; 1: max(density, 0); 2: x * max(density, 0); 3: z * max(density, 0)
F3D_HD float sim_sum_row(const SimGrid &G, const float *density, uint32_t kind, uint32_t y, uint32_t z) {
    float row = 0.0f;
    for (uint32_t x = 0u; x < G.nx; x++) {
        const float d = density[sim_index(G, x, y, z)], m = f_max(d, 0.0f);
        row += kind == 0u ? d : (kind == 1u ? m : (kind == 2u ? (float)x * m : (float)z * m));
    }
    return row;
}
F3D_HD float sim_sum_seq(const float *v, uint32_t n) {
    float s = 0.0f;
    for (uint32_t k = 0u; k < n; k++) s += v[k];
    return s;
}

}  // namespace smoke
}  // namespace f3d
