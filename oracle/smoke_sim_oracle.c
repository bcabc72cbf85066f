/* oracle/smoke_sim_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked or imported by the product).
 *
 * Restates the reference's smoke transport solver, one function per reference function (src/smoke/sim.rs):
 *   SmokeVolume::add_emitter                  :7-45         apply_forces                   :162-234
 *   SmokeVolume::step                         :47-139       apply_scalar_diffusion         :236-245
 *   apply_decay_and_age                       :247-268      project                        :270-317
 *   apply_lane_advection_shear                :319-423      apply_subgrid_density_eddies   :425-518
 *   apply_boundary_conditions                 :520-551      apply_vorticity_confinement    :553-591
 *   advect_scalar / advect_vector             :594-657      backtrace / forwardtrace / local_min_max :659-697
 *   scale_to_mass / diffuse_* / compute_divergence / curl_at :699-799
 *   sample_scalar / sample_vector_component / lerp / smoothstep   src/smoke/sampling.rs:1-94
 *   SmokeStepSettings / SmokeEmitter defaults + validate          src/smoke/types.rs:69-226
 * PARITY PIN: the reference (Rust) cannot be built here and ships no golden state for the solver; this oracle is pinned
 * by the reference's own four solver tests restated (sim.rs:801-899: emitter adds the required fields; smoke advects
 * with the wind and preserves mass to 2 %; a buoyant plume rises; the projection reduces the divergence) in
 * tests/test_smoke_sim.py -- "parity pinned by KAT properties only".
 * Two deliberate numerical choices (results differ from a Rust build in the last bits either way -- it calls libm):
 *   * sin / cos / exp are the fixed polynomials shared (restated, not included) with the HIP kernels;
 *   * sums over the grid (mass, centroid) are taken row by row, then over the rows of a slab, then over the slabs --
 *     the reference adds all voxels in one sequential chain (iter().sum()), an order no parallel machine can follow;
 *     the difference is a rounding of the total (1e-7 relative) and enters only through the mass rescale and the
 *     eddy centres. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float *density, *temperature, *fuel, *soot, *humidity, *emission_rate, *particle_age, *velocity, *pressure;
    uint32_t dims[3];
    float voxel_size[3], origin[3];
    float sparse_threshold;
    float time_seconds;
    uint32_t frame_index;
} sim_volume;

typedef struct { /* SmokeStepSettings, types.rs:142-158 */
    float dt, density_decay, temperature_decay, velocity_damping, diffusion, buoyancy, vorticity;
    uint32_t pressure_iterations;
    float turbulence_strength;
    uint32_t turbulence_seed;
    int32_t mac_cormack, mass_conservation, terrain_collision;
    float boundary_damping;
    float wind[3];
} sim_settings;

typedef struct { /* SmokeEmitter, types.rs:69-81 */
    float center[3], radius, density_rate, temperature_rate, fuel_rate, soot_rate, humidity_rate, emission_rate, velocity[3],
        start_time, end_time;
} sim_emitter;

/* ---- deterministic transcendentals (shared with csrc/f3d_math.h, restated) ---------------------------- */
static float s_exp(float x) {
    if (x > 88.0f) return INFINITY;
    if (x < -103.0f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float z = r * r;
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, z, r) + 1.0f;
    int e = (int)n;
    union { uint32_t u; float f; } sc;
    if (e < -126) {
        sc.u = (uint32_t)(e + 64 + 127) << 23;
        return (y * sc.f) * 5.42101086242752217e-20f;
    }
    sc.u = (uint32_t)(e + 127) << 23;
    return y * sc.f;
}
static void s_sincos(float a, float *s_out, float *c_out) { /* f3d_wf_path.h sincos_rad: the angle in turns, then quadrants */
    float u = a * 0.15915494309189533577f;
    u = u - floorf(u);
    float q4 = 4.0f * u, k = rintf(q4), x = (q4 - k) * 1.57079632679489661923f;
    float z = x * x;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = fmaf(ps * z, x, x);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
    int q = ((int)k) & 3;
    *s_out = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    *c_out = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}
static float s_sin(float a) { float s, c; s_sincos(a, &s, &c); return s; }
static float s_cos(float a) { float s, c; s_sincos(a, &s, &c); return c; }

static float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static float lerpf(float a, float b, float t) { return a + (b - a) * t; }                       /* sampling.rs:83-85 */
static float smoothstep(float e0, float e1, float x) {                                          /* sampling.rs:87-90 */
    float t = clampf((x - e0) / fmaxf(e1 - e0, 1.0e-6f), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
static size_t idx3(const uint32_t d[3], uint32_t x, uint32_t y, uint32_t z) { return ((size_t)z * d[1] + y) * d[0] + x; }
static uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
static uint32_t sat_sub1(uint32_t n) { return n > 0u ? n - 1u : 0u; }
static float axis_unit(uint32_t i, uint32_t n) { uint32_t m = sat_sub1(n); return (float)i / (float)(m > 1u ? m : 1u); } /* x / (n-1).max(1) */

/* sample_scalar / sample_vector_component, sampling.rs:1-81 (stride 1 / 3) */
static float sample_strided(const float *f, const uint32_t d[3], const float p[3], uint32_t stride, uint32_t comp) {
    float x = clampf(p[0], 0.0f, (float)(d[0] - 1u)), y = clampf(p[1], 0.0f, (float)(d[1] - 1u)), z = clampf(p[2], 0.0f, (float)(d[2] - 1u));
    uint32_t x0 = (uint32_t)floorf(x), y0 = (uint32_t)floorf(y), z0 = (uint32_t)floorf(z);
    uint32_t x1 = umin(x0 + 1u, d[0] - 1u), y1 = umin(y0 + 1u, d[1] - 1u), z1 = umin(z0 + 1u, d[2] - 1u);
    float fx = x - (float)x0, fy = y - (float)y0, fz = z - (float)z0;
#define RD(X, Y, Z) f[idx3(d, X, Y, Z) * stride + comp]
    float c00 = lerpf(RD(x0, y0, z0), RD(x1, y0, z0), fx), c10 = lerpf(RD(x0, y1, z0), RD(x1, y1, z0), fx);
    float c01 = lerpf(RD(x0, y0, z1), RD(x1, y0, z1), fx), c11 = lerpf(RD(x0, y1, z1), RD(x1, y1, z1), fx);
#undef RD
    return lerpf(lerpf(c00, c10, fy), lerpf(c01, c11, fy), fz);
}
static void sample_vector(const float *v, const uint32_t d[3], const float p[3], float out[3]) {
    for (uint32_t c = 0; c < 3u; c++) out[c] = sample_strided(v, d, p, 3u, c);
}

/* the grid sums (see the header): rows, then the rows of a slab, then the slabs */
static float grid_sum(const sim_volume *V, float (*term)(const sim_volume *, uint32_t, uint32_t, uint32_t)) {
    float total = 0.0f;
    for (uint32_t z = 0; z < V->dims[2]; z++) {
        float slab = 0.0f;
        for (uint32_t y = 0; y < V->dims[1]; y++) {
            float row = 0.0f;
            for (uint32_t x = 0; x < V->dims[0]; x++) row += term(V, x, y, z);
            slab += row;
        }
        total += slab;
    }
    return total;
}
static float term_density(const sim_volume *V, uint32_t x, uint32_t y, uint32_t z) { return V->density[idx3(V->dims, x, y, z)]; }
static float term_mass(const sim_volume *V, uint32_t x, uint32_t y, uint32_t z) { return fmaxf(V->density[idx3(V->dims, x, y, z)], 0.0f); }
static float term_mass_x(const sim_volume *V, uint32_t x, uint32_t y, uint32_t z) { return (float)x * fmaxf(V->density[idx3(V->dims, x, y, z)], 0.0f); }
static float term_mass_z(const sim_volume *V, uint32_t x, uint32_t y, uint32_t z) { return (float)z * fmaxf(V->density[idx3(V->dims, x, y, z)], 0.0f); }
float smoke_sim_mass(const sim_volume *V) { return grid_sum(V, term_density); } /* SmokeVolume::mass, types.rs:407-409 */

/* add_emitter, sim.rs:7-45 */
static void add_emitter(sim_volume *V, const sim_emitter *E, float dt) {
    const uint32_t *d = V->dims;
    const float radius = fmaxf(E->radius, 1.0e-6f);
    for (uint32_t z = 0; z < d[2]; z++)
        for (uint32_t y = 0; y < d[1]; y++)
            for (uint32_t x = 0; x < d[0]; x++) {
                const size_t i = idx3(d, x, y, z);
                const float px = V->origin[0] + ((float)x + 0.5f) * V->voxel_size[0], py = V->origin[1] + ((float)y + 0.5f) * V->voxel_size[1],
                            pz = V->origin[2] + ((float)z + 0.5f) * V->voxel_size[2];
                const float dx = px - E->center[0], dy = py - E->center[1], dz = pz - E->center[2];
                const float dist = sqrtf((dx * dx + dy * dy) + dz * dz); /* glam distance: length of the difference */
                if (dist > radius) continue;
                const float falloff = 1.0f - smoothstep(0.0f, radius, dist), amount = dt * falloff;
                V->density[i] = fmaxf(V->density[i] + E->density_rate * amount, 0.0f);
                V->temperature[i] = fmaxf(V->temperature[i] + E->temperature_rate * amount, 0.0f);
                V->fuel[i] = fmaxf(V->fuel[i] + E->fuel_rate * amount, 0.0f);
                V->soot[i] = fmaxf(V->soot[i] + E->soot_rate * amount, 0.0f);
                V->humidity[i] = fmaxf(V->humidity[i] + E->humidity_rate * amount, 0.0f);
                V->emission_rate[i] += E->emission_rate * falloff;
                V->particle_age[i] = 0.0f;
                for (uint32_t c = 0; c < 3u; c++) V->velocity[3 * i + c] += E->velocity[c] * amount;
            }
}

/* apply_forces, sim.rs:162-234 */
static void apply_forces(sim_volume *V, const sim_settings *S) {
    const uint32_t *d = V->dims;
    for (uint32_t z = 0; z < d[2]; z++)
        for (uint32_t y = 0; y < d[1]; y++)
            for (uint32_t x = 0; x < d[0]; x++) {
                const size_t i = idx3(d, x, y, z), vi = 3 * i;
                float *v = V->velocity + vi;
                v[0] += S->wind[0] * S->dt;
                v[1] += (S->wind[1] + V->temperature[i] * S->buoyancy) * S->dt;
                v[2] += S->wind[2] * S->dt;
                if (S->velocity_damping > 0.0f) {
                    const float damping = s_exp(-S->velocity_damping * S->dt);
                    v[0] *= damping; v[1] *= damping; v[2] *= damping;
                }
                if (S->turbulence_strength > 0.0f) {
                    const float xf = axis_unit(x, d[0]), yf = axis_unit(y, d[1]), zf = axis_unit(z, d[2]);
                    const float seed_phase = (float)S->turbulence_seed * 0.000137f, t = V->time_seconds;
                    const float amp = S->turbulence_strength * S->dt;
                    const float altitude_gain = clampf(0.45f + 0.75f * yf, 0.35f, 1.20f);
                    const float lane_a = s_sin(xf * 9.6f + zf * 4.2f + yf * 1.6f + t * 0.52f + seed_phase);
                    const float lane_b = s_cos(zf * 7.4f - xf * 5.1f + yf * 2.7f - t * 0.37f + seed_phase * 1.7f);
                    const float roll = s_sin((xf + zf) * 3.9f - yf * 5.2f + t * 0.29f + seed_phase * 0.6f);
                    v[0] += (0.62f * lane_a + 0.28f * roll) * amp * altitude_gain;
                    v[1] += (0.08f * lane_b - 0.05f * roll) * amp;
                    v[2] += (-0.56f * lane_b + 0.26f * lane_a) * amp * altitude_gain;
                    const float wind_len = sqrtf(S->wind[0] * S->wind[0] + S->wind[2] * S->wind[2]);
                    if (wind_len > 1.0e-6f) {
                        const float wind_x = S->wind[0] / wind_len, wind_z = S->wind[2] / wind_len, cross_x = -wind_z, cross_z = wind_x;
                        const float along = (float)x * wind_x + (float)z * wind_z, cross_coord = (float)x * cross_x + (float)z * cross_z;
                        const float lane_phase = along * 0.34f + cross_coord * 0.72f + t * 0.34f + seed_phase * 11.0f;
                        const float lane_force = s_sin(lane_phase) + 0.45f * s_sin(lane_phase * 0.53f + (float)z * 0.29f);
                        const float speed_lane = 0.5f + 0.5f * s_cos(lane_phase * 0.41f + (float)x * 0.18f);
                        v[0] += cross_x * lane_force * amp * 0.82f * altitude_gain + wind_x * speed_lane * amp * 0.30f * altitude_gain;
                        v[2] += cross_z * lane_force * amp * 0.82f * altitude_gain + wind_z * speed_lane * amp * 0.30f * altitude_gain;
                        const float shear = (yf - 0.42f) * amp * 1.35f;
                        v[0] += cross_x * shear;
                        v[2] += cross_z * shear;
                    }
                }
            }
}

/* backtrace / forwardtrace, sim.rs:659-673 */
static void trace(const float p[3], const float v[3], const float vs[3], float dt, float sign, float out[3]) {
    for (int c = 0; c < 3; c++) out[c] = sign < 0.0f ? p[c] - v[c] * dt / vs[c] : p[c] + v[c] * dt / vs[c];
}
/* local_min_max, sim.rs:675-697 */
static void local_min_max(const float *f, const uint32_t d[3], const float p[3], float *lo_out, float *hi_out) {
    uint32_t x0 = (uint32_t)clampf(floorf(p[0]), 0.0f, (float)(d[0] - 1u)), y0 = (uint32_t)clampf(floorf(p[1]), 0.0f, (float)(d[1] - 1u)),
             z0 = (uint32_t)clampf(floorf(p[2]), 0.0f, (float)(d[2] - 1u));
    uint32_t x1 = umin(x0 + 1u, d[0] - 1u), y1 = umin(y0 + 1u, d[1] - 1u), z1 = umin(z0 + 1u, d[2] - 1u);
    float lo = INFINITY, hi = -INFINITY;
    for (uint32_t z = z0; z <= z1; z++)
        for (uint32_t y = y0; y <= y1; y++)
            for (uint32_t x = x0; x <= x1; x++) {
                const float v = f[idx3(d, x, y, z)];
                lo = fminf(lo, v);
                hi = fmaxf(hi, v);
            }
    *lo_out = lo;
    *hi_out = hi;
}
/* advect_scalar, sim.rs:594-636: returns a new field (caller frees) */
static float *advect_scalar(const float *old, const float *vel, const uint32_t d[3], const float vs[3], float dt, int mac_cormack) {
    const size_t n = (size_t)d[0] * d[1] * d[2];
    float *pred = (float *)malloc(n * sizeof(float));
    for (uint32_t z = 0; z < d[2]; z++)
        for (uint32_t y = 0; y < d[1]; y++)
            for (uint32_t x = 0; x < d[0]; x++) {
                const float p[3] = {(float)x, (float)y, (float)z};
                float v[3], back[3];
                sample_vector(vel, d, p, v);
                trace(p, v, vs, dt, -1.0f, back);
                pred[idx3(d, x, y, z)] = fmaxf(sample_strided(old, d, back, 1u, 0u), 0.0f);
            }
    if (!mac_cormack) return pred;
    float *corr = (float *)malloc(n * sizeof(float));
    memcpy(corr, pred, n * sizeof(float));
    for (uint32_t z = 0; z < d[2]; z++)
        for (uint32_t y = 0; y < d[1]; y++)
            for (uint32_t x = 0; x < d[0]; x++) {
                const size_t i = idx3(d, x, y, z);
                const float p[3] = {(float)x, (float)y, (float)z};
                float v[3], back[3], vb[3], fwd[3], lo, hi;
                sample_vector(vel, d, p, v);
                trace(p, v, vs, dt, -1.0f, back);
                sample_vector(vel, d, back, vb);
                trace(back, vb, vs, dt, 1.0f, fwd);
                const float recovered = sample_strided(pred, d, fwd, 1u, 0u);
                const float candidate = pred[i] + 0.5f * (old[i] - recovered);
                local_min_max(old, d, back, &lo, &hi);
                corr[i] = fmaxf(clampf(candidate, lo, hi), 0.0f);
            }
    free(pred);
    return corr;
}
/* advect_vector, sim.rs:638-657 */
static float *advect_vector(const float *old, const float *vel, const uint32_t d[3], const float vs[3], float dt) {
    const size_t n = (size_t)d[0] * d[1] * d[2];
    float *out = (float *)malloc(3 * n * sizeof(float));
    for (uint32_t z = 0; z < d[2]; z++)
        for (uint32_t y = 0; y < d[1]; y++)
            for (uint32_t x = 0; x < d[0]; x++) {
                const float p[3] = {(float)x, (float)y, (float)z};
                float v[3], back[3];
                sample_vector(vel, d, p, v);
                trace(p, v, vs, dt, -1.0f, back);
                for (uint32_t c = 0; c < 3u; c++) out[3 * idx3(d, x, y, z) + c] = sample_strided(old, d, back, 3u, c);
            }
    return out;
}
/* scale_to_mass, sim.rs:699-711 (with the grid sum of the header) */
static void scale_to_mass(sim_volume *V, float target) {
    if (target <= 0.0f) return;
    const float mass = smoke_sim_mass(V);
    if (mass > 1.0e-12f) {
        const float scale = target / mass;
        const size_t n = (size_t)V->dims[0] * V->dims[1] * V->dims[2];
        for (size_t i = 0; i < n; i++) V->density[i] *= scale;
    }
}
/* diffuse_scalar_in_place on a strided field, sim.rs:713-737 (+ diffuse_vector :739-754) */
static void diffuse_strided(float *f, const uint32_t d[3], float rate, float dt, uint32_t stride, uint32_t comp) {
    if (rate <= 0.0f) return;
    const size_t n = (size_t)d[0] * d[1] * d[2];
    float *old = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) old[i] = f[i * stride + comp];
    const float alpha = rate * dt;
    for (uint32_t z = 1; z + 1 < d[2]; z++)
        for (uint32_t y = 1; y + 1 < d[1]; y++)
            for (uint32_t x = 1; x + 1 < d[0]; x++) {
                const float sum = old[idx3(d, x - 1, y, z)] + old[idx3(d, x + 1, y, z)] + old[idx3(d, x, y - 1, z)] + old[idx3(d, x, y + 1, z)] +
                                  old[idx3(d, x, y, z - 1)] + old[idx3(d, x, y, z + 1)];
                f[idx3(d, x, y, z) * stride + comp] = (old[idx3(d, x, y, z)] + alpha * sum) / (1.0f + 6.0f * alpha);
            }
    free(old);
}
/* compute_divergence, sim.rs:756-777 */
static float *compute_divergence(const float *v, const uint32_t d[3], const float vs[3]) {
    float *div = (float *)calloc((size_t)d[0] * d[1] * d[2], sizeof(float));
    for (uint32_t z = 1; z + 1 < d[2]; z++)
        for (uint32_t y = 1; y + 1 < d[1]; y++)
            for (uint32_t x = 1; x + 1 < d[0]; x++) {
                const float du = (v[idx3(d, x + 1, y, z) * 3] - v[idx3(d, x - 1, y, z) * 3]) / (2.0f * vs[0]);
                const float dv = (v[idx3(d, x, y + 1, z) * 3 + 1] - v[idx3(d, x, y - 1, z) * 3 + 1]) / (2.0f * vs[1]);
                const float dw = (v[idx3(d, x, y, z + 1) * 3 + 2] - v[idx3(d, x, y, z - 1) * 3 + 2]) / (2.0f * vs[2]);
                div[idx3(d, x, y, z)] = du + dv + dw;
            }
    return div;
}
float smoke_sim_divergence_l2(const sim_volume *V) { /* divergence_l2, sim.rs:140-145 (one sequential chain: a diagnostic, not part of a step) */
    float *div = compute_divergence(V->velocity, V->dims, V->voxel_size);
    const size_t n = (size_t)V->dims[0] * V->dims[1] * V->dims[2];
    float sum = 0.0f;
    for (size_t i = 0; i < n; i++) sum += div[i] * div[i];
    free(div);
    return sqrtf(sum / (float)(n > 1 ? n : 1));
}
/* project, sim.rs:270-317 */
static void project(sim_volume *V, uint32_t iterations) {
    const uint32_t *d = V->dims;
    const size_t n = (size_t)d[0] * d[1] * d[2];
    float *div = compute_divergence(V->velocity, d, V->voxel_size);
    memset(V->pressure, 0, n * sizeof(float));
    float *cur = V->pressure, *next = (float *)calloc(n, sizeof(float));
    for (uint32_t it = 0; it < iterations; it++) {
        for (uint32_t z = 1; z + 1 < d[2]; z++)
            for (uint32_t y = 1; y + 1 < d[1]; y++)
                for (uint32_t x = 1; x + 1 < d[0]; x++) {
                    const float sum = cur[idx3(d, x - 1, y, z)] + cur[idx3(d, x + 1, y, z)] + cur[idx3(d, x, y - 1, z)] + cur[idx3(d, x, y + 1, z)] +
                                      cur[idx3(d, x, y, z - 1)] + cur[idx3(d, x, y, z + 1)];
                    next[idx3(d, x, y, z)] = (sum - div[idx3(d, x, y, z)]) / 6.0f;
                }
        float *t = cur; cur = next; next = t; /* std::mem::swap */
    }
    if (cur != V->pressure) { memcpy(V->pressure, cur, n * sizeof(float)); next = cur; }
    free(next);
    const float *P = V->pressure;
    for (uint32_t z = 1; z + 1 < d[2]; z++)
        for (uint32_t y = 1; y + 1 < d[1]; y++)
            for (uint32_t x = 1; x + 1 < d[0]; x++) {
                const size_t vi = 3 * idx3(d, x, y, z);
                V->velocity[vi] -= (P[idx3(d, x + 1, y, z)] - P[idx3(d, x - 1, y, z)]) / (2.0f * V->voxel_size[0]);
                V->velocity[vi + 1] -= (P[idx3(d, x, y + 1, z)] - P[idx3(d, x, y - 1, z)]) / (2.0f * V->voxel_size[1]);
                V->velocity[vi + 2] -= (P[idx3(d, x, y, z + 1)] - P[idx3(d, x, y, z - 1)]) / (2.0f * V->voxel_size[2]);
            }
    free(div);
}
/* apply_vorticity_confinement, sim.rs:553-591 (+ curl_at :779-799) */
static void vorticity(sim_volume *V, float strength, float dt) {
    const uint32_t *d = V->dims;
    const size_t n = (size_t)d[0] * d[1] * d[2];
    const float *v = V->velocity, *vs = V->voxel_size;
    float *curl = (float *)calloc(3 * n, sizeof(float)), *mag = (float *)calloc(n, sizeof(float));
#define RV(X, Y, Z, C) v[idx3(d, X, Y, Z) * 3 + C]
    for (uint32_t z = 1; z + 1 < d[2]; z++)
        for (uint32_t y = 1; y + 1 < d[1]; y++)
            for (uint32_t x = 1; x + 1 < d[0]; x++) {
                const float dw_dy = (RV(x, y + 1, z, 2) - RV(x, y - 1, z, 2)) / (2.0f * vs[1]), dv_dz = (RV(x, y, z + 1, 1) - RV(x, y, z - 1, 1)) / (2.0f * vs[2]);
                const float du_dz = (RV(x, y, z + 1, 0) - RV(x, y, z - 1, 0)) / (2.0f * vs[2]), dw_dx = (RV(x + 1, y, z, 2) - RV(x - 1, y, z, 2)) / (2.0f * vs[0]);
                const float dv_dx = (RV(x + 1, y, z, 1) - RV(x - 1, y, z, 1)) / (2.0f * vs[0]), du_dy = (RV(x, y + 1, z, 0) - RV(x, y - 1, z, 0)) / (2.0f * vs[1]);
                const size_t i = idx3(d, x, y, z);
                const float cx = dw_dy - dv_dz, cy = du_dz - dw_dx, cz = dv_dx - du_dy;
                curl[3 * i] = cx; curl[3 * i + 1] = cy; curl[3 * i + 2] = cz;
                mag[i] = sqrtf((cx * cx + cy * cy) + cz * cz); /* glam Vec3::length */
            }
#undef RV
    for (uint32_t z = 2; z + 2 < d[2]; z++)
        for (uint32_t y = 2; y + 2 < d[1]; y++)
            for (uint32_t x = 2; x + 2 < d[0]; x++) {
                const size_t i = idx3(d, x, y, z);
                const float gx = mag[idx3(d, x + 1, y, z)] - mag[idx3(d, x - 1, y, z)], gy = mag[idx3(d, x, y + 1, z)] - mag[idx3(d, x, y - 1, z)],
                            gz = mag[idx3(d, x, y, z + 1)] - mag[idx3(d, x, y, z - 1)];
                const float len2 = (gx * gx + gy * gy) + gz * gz;
                float nx = 0.0f, ny = 0.0f, nz = 0.0f;
                if (len2 > 1.0e-12f) { /* glam normalize: v * (1 / length) */
                    const float inv = 1.0f / sqrtf(len2);
                    nx = gx * inv; ny = gy * inv; nz = gz * inv;
                }
                const float cx = curl[3 * i], cy = curl[3 * i + 1], cz = curl[3 * i + 2];
                /* n.cross(curl) * strength * dt */
                V->velocity[3 * i] += ((ny * cz - cy * nz) * strength) * dt;
                V->velocity[3 * i + 1] += ((nz * cx - cz * nx) * strength) * dt;
                V->velocity[3 * i + 2] += ((nx * cy - cx * ny) * strength) * dt;
            }
    free(curl);
    free(mag);
}
/* apply_boundary_conditions, sim.rs:520-551 */
static void boundary(sim_volume *V, const sim_settings *S) {
    const uint32_t *d = V->dims;
    const float keep = 1.0f - S->boundary_damping;
    for (uint32_t z = 0; z < d[2]; z++)
        for (uint32_t y = 0; y < d[1]; y++)
            for (uint32_t x = 0; x < d[0]; x++) {
                const size_t i = idx3(d, x, y, z), vi = 3 * i;
                if (x == 0u || x == d[0] - 1u) { V->velocity[vi] = 0.0f; V->density[i] *= 0.58f; V->temperature[i] *= 0.70f; }
                else if (x == 1u || x == d[0] - 2u) { V->density[i] *= 0.78f; V->temperature[i] *= 0.86f; }
                if (y == 0u || y == d[1] - 1u) V->velocity[vi + 1] = 0.0f;
                if (z == 0u || z == d[2] - 1u) { V->velocity[vi + 2] = 0.0f; V->density[i] *= 0.58f; V->temperature[i] *= 0.70f; }
                else if (z == 1u || z == d[2] - 2u) { V->density[i] *= 0.78f; V->temperature[i] *= 0.86f; }
                if (S->terrain_collision && y == 0u) { V->density[i] *= keep; V->temperature[i] *= keep; }
            }
}
/* apply_lane_advection_shear, sim.rs:319-423 */
static void lane_shear(sim_volume *V, const sim_settings *S) {
    if (S->turbulence_strength <= 0.0f) return;
    const uint32_t *d = V->dims;
    const float wind_len = sqrtf(S->wind[0] * S->wind[0] + S->wind[2] * S->wind[2]);
    if (wind_len <= 1.0e-6f) return;
    const float wind_x = S->wind[0] / wind_len, wind_z = S->wind[2] / wind_len, cross_x = -wind_z, cross_z = wind_x;
    const float amp = S->turbulence_strength * S->dt, seed_phase = (float)S->turbulence_seed * 0.0027f;
    const float total_mass = grid_sum(V, term_mass);
    float centroid_x = grid_sum(V, term_mass_x), centroid_z = grid_sum(V, term_mass_z);
    if (total_mass > 1.0e-6f) { centroid_x /= total_mass; centroid_z /= total_mass; }
    static const float eddies[4][4] = {{5.5f, 5.4f, 1.0f, 1.85f}, {11.5f, 7.6f, -1.0f, 1.58f}, {19.0f, 10.2f, 1.0f, 1.30f}, {28.0f, 13.0f, -1.0f, 1.05f}};
    const float fi = (float)V->frame_index;
    for (uint32_t z = 0; z < d[2]; z++)
        for (uint32_t y = 0; y < d[1]; y++)
            for (uint32_t x = 0; x < d[0]; x++) {
                const size_t i = idx3(d, x, y, z), vi = 3 * i;
                const float active = smoothstep(V->sparse_threshold, fmaxf(V->sparse_threshold * 60.0f, 0.012f), V->density[i]);
                if (active <= 0.0f) continue;
                const float along = (float)x * wind_x + (float)z * wind_z, cross_coord = (float)x * cross_x + (float)z * cross_z;
                const float lane_phase = along * 0.23f + cross_coord * 0.49f + fi * 0.105f + seed_phase;
                const float lane_force = s_sin(lane_phase) + 0.58f * s_sin(lane_phase * 0.41f + (float)y * 0.74f);
                const float altitude = axis_unit(y, d[1]);
                const float altitude_shear = (altitude - 0.44f) * 0.95f;
                const float force = (lane_force * 2.75f + altitude_shear * 1.45f) * active * amp;
                V->velocity[vi] += cross_x * force;
                V->velocity[vi + 2] += cross_z * force;
                const float slab_phase = along * 0.17f - cross_coord * 0.31f + (float)y * 1.12f + fi * 0.043f + (float)S->turbulence_seed * 0.0021f;
                const float slab_lane = s_sin(slab_phase) + 0.42f * s_sin(slab_phase * 0.53f + along * 0.09f);
                const float slab_split = ((altitude - 0.50f) * 2.55f + slab_lane * 0.58f) * active * amp;
                V->velocity[vi] += cross_x * slab_split * 1.90f;
                V->velocity[vi + 2] += cross_z * slab_split * 1.90f;
                const float speed_split = s_sin(slab_phase * 0.39f + (float)y * 0.67f) * active * amp;
                V->velocity[vi] += wind_x * speed_split * 0.52f;
                V->velocity[vi + 2] += wind_z * speed_split * 0.52f;
                if (total_mass > 1.0e-6f) {
                    const float altitude_gain = 0.55f + 0.75f * altitude;
                    for (uint32_t e = 0; e < 4u; e++) {
                        const float distance = eddies[e][0], radius = eddies[e][1], side = eddies[e][2], strength = eddies[e][3];
                        const float phase = fi * (0.035f + (float)e * 0.006f) + (float)S->turbulence_seed * 0.0013f;
                        const float center_x = centroid_x + wind_x * distance + cross_x * side * radius * (0.40f + 0.20f * s_sin(phase));
                        const float center_z = centroid_z + wind_z * distance + cross_z * side * radius * (0.40f + 0.20f * s_cos(phase));
                        const float dx = (float)x - center_x, dz = (float)z - center_z, r2 = dx * dx + dz * dz;
                        const float envelope = s_exp(-r2 / (2.0f * radius * radius)) * active;
                        const float inv_r = 1.0f / sqrtf(r2 + 1.0f);
                        const float spin = side * strength * amp * envelope * altitude_gain;
                        V->velocity[vi] += -dz * inv_r * spin;
                        V->velocity[vi + 2] += dx * inv_r * spin;
                    }
                }
            }
}
/* apply_subgrid_density_eddies, sim.rs:425-518 */
static void subgrid_eddies(sim_volume *V, const sim_settings *S) {
    if (S->turbulence_strength <= 0.0f) return;
    const uint32_t *d = V->dims;
    const float wind_len = fmaxf(sqrtf(S->wind[0] * S->wind[0] + S->wind[2] * S->wind[2]), 1.0e-6f);
    const float wind_x = S->wind[0] / wind_len, wind_z = S->wind[2] / wind_len, cross_x = -wind_z, cross_z = wind_x;
    const float seed_phase = (float)S->turbulence_seed * 0.0019f, t = (float)V->frame_index * 0.046f;
    for (uint32_t z = 0; z < d[2]; z++)
        for (uint32_t y = 0; y < d[1]; y++)
            for (uint32_t x = 0; x < d[0]; x++) {
                const size_t i = idx3(d, x, y, z);
                const float active = smoothstep(V->sparse_threshold, fmaxf(V->sparse_threshold * 90.0f, 0.018f), V->density[i]);
                if (active <= 0.0f) continue;
                const float xf = (float)x, yf = (float)y, zf = (float)z;
                const float wob = s_sin(xf * 0.043f + zf * 0.071f + t + seed_phase);
                const float phase = xf * 0.18f + zf * 0.27f + yf * 0.72f + wob * 1.7f + t + seed_phase;
                const float ribbons = 0.5f + 0.5f * s_sin(phase);
                const float sheets = 0.5f + 0.5f * s_sin(phase * 0.47f - zf * 0.16f + yf * 0.51f);
                const float voids = smoothstep(0.45f, 0.84f, 1.0f - ribbons) * smoothstep(0.34f, 0.76f, 1.0f - sheets) * active;
                const float ridges = smoothstep(0.62f, 0.94f, ribbons) * smoothstep(0.48f, 0.90f, sheets) * active;
                const float age_t = smoothstep(2.0f, 28.0f, fmaxf(V->particle_age[i], 0.0f));
                const float void_strength = 0.62f + 0.32f * age_t, ridge_strength = 0.075f - 0.045f * age_t;
                const float along = xf * wind_x + zf * wind_z, cross_coord = xf * cross_x + zf * cross_z;
                const float broad = 0.5f + 0.5f * wob;
                const float channel_phase = along * 0.115f + cross_coord * 0.52f + broad * 5.4f + s_sin(yf * 0.62f + along * 0.035f) * 0.85f +
                                            (float)V->frame_index * 0.033f + (float)S->turbulence_seed * 0.0023f;
                const float channel_wave = 0.5f + 0.5f * s_sin(channel_phase) + 0.28f * s_sin(channel_phase * 0.47f - cross_coord * 0.19f + yf * 0.34f);
                const float entrainment = smoothstep(0.58f, 1.06f, channel_wave);
                const float lateral_slots = smoothstep(0.50f, 0.94f, 1.0f - (0.62f * ribbons + 0.38f * sheets));
                const float core_protect = 1.0f - 0.56f * smoothstep(0.72f, 1.75f, V->density[i]);
                const float aged_sheet = (0.28f + 0.72f * age_t) * active * core_protect;
                const float clear_air = clampf(entrainment * (0.54f + 0.46f * lateral_slots) * aged_sheet, 0.0f, 1.0f);
                const float channel_void = clampf(smoothstep(0.42f, 0.86f, 1.0f - channel_wave) * (0.55f + 0.45f * lateral_slots) * active *
                                                      (0.42f + 0.58f * age_t) * core_protect, 0.0f, 1.0f);
                const float gain = (1.0f - void_strength * voids + ridge_strength * ridges) * (1.0f - (0.024f + 0.055f * age_t) * clear_air) *
                                   (1.0f - (0.045f + 0.070f * age_t) * channel_void);
                V->density[i] = clampf(V->density[i] * gain, 0.0f, 8.0f);
                V->humidity[i] = fmaxf(V->humidity[i] * (1.0f - (0.15f + 0.10f * age_t) * voids - (0.024f + 0.055f * age_t) * clear_air -
                                                         (0.045f + 0.070f * age_t) * channel_void), 0.0f);
            }
}
/* apply_decay_and_age, sim.rs:247-268 */
static void decay_and_age(sim_volume *V, const sim_settings *S) {
    const float temperature_decay = s_exp(-S->temperature_decay * S->dt);
    const size_t n = (size_t)V->dims[0] * V->dims[1] * V->dims[2];
    for (size_t i = 0; i < n; i++) {
        const float age_t = smoothstep(7.0f, 36.0f, fmaxf(V->particle_age[i], 0.0f));
        const float density_decay = s_exp(-S->density_decay * S->dt * (1.0f + 3.0f * age_t));
        const float soot_decay = s_exp(-S->density_decay * S->dt * (0.42f + 1.15f * age_t));
        V->density[i] *= density_decay;
        V->temperature[i] *= temperature_decay;
        V->fuel[i] *= density_decay;
        V->soot[i] *= soot_decay;
        if (V->density[i] > V->sparse_threshold) V->particle_age[i] = V->particle_age[i] < 0.0f ? 0.0f : V->particle_age[i] + S->dt;
        else V->particle_age[i] = -1.0f;
    }
}
static void replace(float **field, float *fresh, size_t n) {
    memcpy(*field, fresh, n * sizeof(float));
    free(fresh);
}

/* add_emitter as the binding exposes it (py.rs:459-464): validation is the caller's */
void smoke_sim_add_emitter(sim_volume *V, const sim_emitter *E, float dt) { add_emitter(V, E, dt); }

/* SmokeVolume::step, sim.rs:47-139.  Returns 0; the validation of settings / emitters (types.rs) is the caller's. */
int smoke_sim_step(sim_volume *V, const sim_settings *S, const sim_emitter *emitters, uint32_t emitter_count) {
    const uint32_t *d = V->dims;
    const size_t n = (size_t)d[0] * d[1] * d[2];
    memset(V->emission_rate, 0, n * sizeof(float));
    for (uint32_t e = 0; e < emitter_count; e++)
        if (V->time_seconds >= emitters[e].start_time && V->time_seconds <= emitters[e].end_time) add_emitter(V, &emitters[e], S->dt);
    apply_forces(V, S);
    {
        float *before = (float *)malloc(3 * n * sizeof(float));
        memcpy(before, V->velocity, 3 * n * sizeof(float));
        float *adv = advect_vector(before, before, d, V->voxel_size, S->dt);
        memcpy(V->velocity, adv, 3 * n * sizeof(float));
        free(adv);
        free(before);
    }
    for (uint32_t c = 0; c < 3u; c++) diffuse_strided(V->velocity, d, S->diffusion, S->dt, 3u, c);
    if (S->vorticity > 0.0f) vorticity(V, S->vorticity, S->dt);
    project(V, S->pressure_iterations > 1u ? S->pressure_iterations : 1u);
    boundary(V, S);
    lane_shear(V, S);
    const float mass_before = smoke_sim_mass(V);
    replace(&V->density, advect_scalar(V->density, V->velocity, d, V->voxel_size, S->dt, S->mac_cormack), n);
    if (S->mass_conservation) scale_to_mass(V, mass_before);
    replace(&V->temperature, advect_scalar(V->temperature, V->velocity, d, V->voxel_size, S->dt, S->mac_cormack), n);
    replace(&V->fuel, advect_scalar(V->fuel, V->velocity, d, V->voxel_size, S->dt, S->mac_cormack), n);
    replace(&V->soot, advect_scalar(V->soot, V->velocity, d, V->voxel_size, S->dt, S->mac_cormack), n);
    replace(&V->humidity, advect_scalar(V->humidity, V->velocity, d, V->voxel_size, S->dt, S->mac_cormack), n);
    subgrid_eddies(V, S);
    if (S->diffusion > 0.0f) { /* apply_scalar_diffusion, sim.rs:236-245 */
        diffuse_strided(V->density, d, S->diffusion, S->dt, 1u, 0u);
        diffuse_strided(V->temperature, d, S->diffusion, S->dt, 1u, 0u);
        diffuse_strided(V->fuel, d, S->diffusion, S->dt, 1u, 0u);
        diffuse_strided(V->soot, d, S->diffusion, S->dt, 1u, 0u);
        diffuse_strided(V->humidity, d, S->diffusion, S->dt, 1u, 0u);
    }
    decay_and_age(V, S);
    project(V, S->pressure_iterations / 2u > 1u ? S->pressure_iterations / 2u : 1u);
    boundary(V, S);
    V->time_seconds += S->dt;
    V->frame_index += 1u;
    return 0;
}
