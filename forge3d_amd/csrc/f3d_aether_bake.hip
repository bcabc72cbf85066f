// forge3d_amd/csrc/f3d_aether_bake.hip -- the AETHER atmosphere LUT baker on gfx950 (offline half).
//
// Reference: bake_atmosphere_luts and what it calls (src/core/atmosphere/bake.rs:776-1666, spectral.rs) -- single-thread
// Rust on the host: transmittance (32 x 8), single scattering by ray marching (64 x 64 steps per entry), then orders
// 2..n of multiple scattering, each a gather over a 16 x 32 sphere quadrature at 16 points along every view ray of the
// 17 x 17 x 8 x 16 table with a quadrilinear fetch of the previous order per direction (~4e6 operations per entry and
// order), ground bounce included; 11 wavelengths throughout.  Offline there (the shipped anchors are its output).
//
// Here: one LANE per table entry for transmittance / single scattering / aerial (short dependent marches), one WAVE per
// entry for the scattering orders: the 512 quadrature directions are dealt 8 to a lane (q = lane + 64 k), every lane
// sums its own in a fixed order and a butterfly (xor 32 ... 1) adds the lanes, so a bake is deterministic; the previous
// order (36 992 x 11 floats = 1.6 MB) stays in L2.  The per-order convergence deltas are summed by one thread in table
// order.  Results: the oracle's (oracle/aether_bake_oracle.c, which reproduces the reference's shipped anchors to the
// last f16 bit in all but ~1 value per table) up to the summation order of the gathers and the device's expf / powf /
// expm1f -- tests bound the difference in f16 ulps.
#include <hip/hip_runtime.h>

#include <cmath>
#include <exception>
#include <vector>

#include "f3d_setup.h"
#include "f3d_devmem.h"

using namespace f3d;

namespace {

constexpr int NW = 11, NQ = 512, kOrderSteps = 16;
__constant__ float kCie[NW][3] = {{0.001368f, 0.000039f, 0.006450f}, {0.134380f, 0.004000f, 0.645600f}, {0.290800f, 0.060000f, 1.669200f},
                                  {0.004900f, 0.323000f, 0.272000f}, {0.290400f, 0.954000f, 0.020300f}, {0.916300f, 0.870000f, 0.001650f},
                                  {0.854450f, 0.381000f, 0.000190f}, {0.164900f, 0.061000f, 0.000000f}, {0.011359f, 0.004102f, 0.000000f},
                                  {0.000690f, 0.000249f, 0.000000f}, {0.000042f, 0.000015f, 0.000000f}};
constexpr float kPiF = 3.14159265358979323846f;

struct BakeParams {
    f3d_aether_bake_config c;
    float H;                // atmosphere height
    float rayleigh[NW], mie_ext[NW], mie_sca[NW], ozone[NW];  // per-wavelength coefficients (spectral.rs:62-92, bake.rs:796)
    float white[3];         // xyz_to_rgb(integrate_xyz(1))
    const float4 *quad;     // NQ directions (xyz) + weight
    float cos_norm;         // pi / sum of cos * weight over the upper hemisphere (bake.rs:1312-1317)
    float *previous, *next, *accumulated, *single;  // [entry][NW]
    uint16_t *t_out, *single_out, *acc_out, *aerial_out;
    float *deltas;
    uint32_t count, order;
};

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ float dot3f(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ float rayleigh_phase(float ct) {
    const float c = clampf(ct, -1.0f, 1.0f);
    return 3.0f * (1.0f + c * c) / (16.0f * kPiF);
}
__device__ __forceinline__ float mie_phase(float ct, float g_in) {
    const float c = clampf(ct, -1.0f, 1.0f), g = clampf(g_in, -0.999f, 0.999f);
    const float den = powf(fmaxf(1.0f + g * g - 2.0f * g * c, 1.0e-6f), 1.5f);
    return 3.0f * (1.0f - g * g) * (1.0f + c * c) / (8.0f * kPiF * (2.0f + g * g) * den);
}
__device__ __forceinline__ float mu_from_unit(float u) {
    const float x = 2.0f * clampf(u, 0.0f, 1.0f) - 1.0f;
    return (signbit(x) ? -1.0f : 1.0f) * (fabsf(x) * fabsf(x));
}
__device__ __forceinline__ float mu_to_unit(float mu_in) {
    const float mu = clampf(mu_in, -1.0f, 1.0f);
    return ((signbit(mu) ? -1.0f : 1.0f) * sqrtf(fabsf(mu)) + 1.0f) * 0.5f;
}
__device__ __forceinline__ float nu_from_unit(float u) {
    const float d = 1.0f - clampf(u, 0.0f, 1.0f);
    return 1.0f - 2.0f * d * d;
}
__device__ __forceinline__ float nu_to_unit(float nu) { return 1.0f - sqrtf((1.0f - clampf(nu, -1.0f, 1.0f)) * 0.5f); }
__device__ __forceinline__ float height_to_unit(float h, float H) { return sqrtf(clampf(h, 0.0f, H) / H); }
__device__ __forceinline__ float height_from_unit(float u, float H) {
    const float c = clampf(u, 0.0f, 1.0f);
    return H * (c * c);
}

__device__ void density_at(const BakeParams &P, float h_in, float *rho) {
    const float h = fmaxf(h_in, 0.0f);
    rho[0] = expf(-h / P.c.rayleigh_scale_height_m);
    rho[1] = expf(-h / P.c.mie_scale_height_m);
    rho[2] = fmaxf(1.0f - fabsf((h - 25000.0f) / 15000.0f), 0.0f) * P.c.ozone_du / 300.0f;
}
__device__ float distance_to_top(const BakeParams &P, float h, float mu) {
    const float r = P.c.bottom_radius_m + clampf(h, 0.0f, P.H);
    const float radial = r * mu;
    const float disc = radial * radial + (P.c.top_radius_m - r) * (P.c.top_radius_m + r);
    return fmaxf(-radial + sqrtf(fmaxf(disc, 0.0f)), 0.0f);
}
__device__ bool distance_to_ground(const BakeParams &P, float h, float mu, float &out) {
    if (mu >= 0.0f) return false;
    const float r = P.c.bottom_radius_m + clampf(h, 0.0f, P.H);
    const float radial = r * mu;
    const float d = radial * radial - (r - P.c.bottom_radius_m) * (r + P.c.bottom_radius_m);
    if (d < 0.0f) return false;
    const float s = -radial - sqrtf(d);
    if (!(s >= 0.0f)) return false;
    out = s;
    return true;
}
__device__ float distance_to_boundary(const BakeParams &P, float h, float mu) {
    float s;
    return distance_to_ground(P, h, mu, s) ? s : distance_to_top(P, h, mu);
}
__device__ float altitude_along(const BakeParams &P, float h, float mu, float s) {
    const float r = P.c.bottom_radius_m + clampf(h, 0.0f, P.H);
    return sqrtf(fmaxf(r * r + s * s + 2.0f * r * mu * s, 0.0f)) - P.c.bottom_radius_m;
}
__device__ void optical_columns(const BakeParams &P, float h, float mu, float d, float *out) {  // 64 steps
    out[0] = out[1] = out[2] = 0.0f;
    if (d <= 0.0f) return;
    const float ds = d / 64.0f;
    for (int i = 0; i < 64; i++) {
        float rho[3];
        density_at(P, altitude_along(P, h, mu, ((float)i + 0.5f) * ds), rho);
        for (int k = 0; k < 3; k++) out[k] += rho[k] * ds;
    }
}
__device__ __forceinline__ float transmittance_at(const BakeParams &P, const float *col, int w) {
    return expf(-fmaxf(P.rayleigh[w] * col[0] + P.mie_ext[w] * col[1] + P.ozone[w] * col[2], 0.0f));
}
__device__ __forceinline__ float extinction_at(const BakeParams &P, const float *rho, int w) {
    return fmaxf(P.rayleigh[w] * rho[0] + P.mie_ext[w] * rho[1] + P.ozone[w] * rho[2], 0.0f);
}
__device__ __forceinline__ float cell_length(float extinction, float ds) {
    return extinction <= 1.0e-12f ? ds : -expm1f(-extinction * ds) / extinction;
}
__device__ void transmittance_segment(const BakeParams &P, float h, float mu, float d, float *t) {
    float col[3];
    optical_columns(P, h, mu, d, col);
    for (int w = 0; w < NW; w++) t[w] = transmittance_at(P, col, w);
}

struct Geom {
    float altitude_m, mu_sun, outgoing[3], sun[3], up[3], tangent[3];
};
__device__ Geom ray_geometry(const BakeParams &P, float h, float mu_view, float mu_sun, float nu, float distance) {
    Geom g;
    const float mv = clampf(mu_view, -1.0f, 1.0f), ms = clampf(mu_sun, -1.0f, 1.0f);
    const float vx = sqrtf(fmaxf(1.0f - mv * mv, 0.0f)), sh = sqrtf(fmaxf(1.0f - ms * ms, 0.0f));
    const float requested = vx > 1.0e-6f ? (clampf(nu, -1.0f, 1.0f) - mv * ms) / vx : 0.0f;
    const float sx = clampf(requested, -sh, sh);
    const float sz = sqrtf(fmaxf(sh * sh - sx * sx, 0.0f));
    g.outgoing[0] = vx, g.outgoing[1] = mv, g.outgoing[2] = 0.0f;
    g.sun[0] = sx, g.sun[1] = ms, g.sun[2] = sz;
    const float r = P.c.bottom_radius_m + clampf(h, 0.0f, P.H);
    const float position[3] = {g.outgoing[0] * distance, r + g.outgoing[1] * distance, 0.0f};
    const float sr = fmaxf(sqrtf(dot3f(position, position)), P.c.bottom_radius_m);
    g.up[0] = position[0] / sr, g.up[1] = position[1] / sr, g.up[2] = 0.0f;
    g.tangent[0] = g.up[1], g.tangent[1] = -g.up[0], g.tangent[2] = 0.0f;
    g.altitude_m = clampf(sr - P.c.bottom_radius_m, 0.0f, P.H);
    g.mu_sun = clampf(dot3f(g.sun, g.up), -1.0f, 1.0f);
    return g;
}

// spectrum -> RGBA16F (bake.rs:1466-1474, spectral.rs:94-124)
__device__ void store_rgba(const BakeParams &P, const float *s, uint16_t *out) {
    float xyz[3] = {0.0f, 0.0f, 0.0f}, sum = 0.0f;
    for (int i = 0; i < NW; i++) {
        const float weight = (i == 0 || i + 1 == NW) ? 0.5f : 1.0f;
        for (int k = 0; k < 3; k++) xyz[k] += s[i] * kCie[i][k] * weight;
        sum += s[i];
    }
    const float raw[3] = {3.2404542f * xyz[0] + -1.5371385f * xyz[1] + -0.4985314f * xyz[2],
                          -0.969266f * xyz[0] + 1.8760108f * xyz[1] + 0.041556f * xyz[2],
                          0.0556434f * xyz[0] + -0.2040259f * xyz[1] + 1.0572252f * xyz[2]};
    for (int k = 0; k < 3; k++) out[k] = half_bits(clampf(raw[k] / P.white[k], 0.0f, 65504.0f));
    out[3] = half_bits(clampf(sum / (float)NW, 0.0f, 65504.0f));
}

__device__ __forceinline__ void entry_coordinates(const BakeParams &P, uint32_t e, float &h, float &nu, float &ms, float &mv) {
    const uint32_t nv = P.c.scattering_mu_view, ns = P.c.scattering_mu_sun, nn = P.c.scattering_nu, nh = P.c.scattering_height;
    const uint32_t vi = e % nv, si = (e / nv) % ns, ni = (e / (nv * ns)) % nn, hi = e / (nv * ns * nn);
    h = height_from_unit((float)hi / (float)(nh - 1u), P.H);
    nu = nu_from_unit((float)ni / (float)(nn - 1u));
    ms = mu_from_unit((float)si / (float)(ns - 1u));
    mv = mu_from_unit((float)vi / (float)(nv - 1u));
}

// direct sun on the ground seen along the ray (order 1's boundary term: ground_boundary_along_ray with no incident field)
__device__ void direct_ground(const BakeParams &P, float h, float mv, float ms, float nu, float *out) {
    for (int w = 0; w < NW; w++) out[w] = 0.0f;
    float length;
    if (!distance_to_ground(P, h, mv, length) || P.c.ground_albedo <= 0.0f) return;
    const Geom end = ray_geometry(P, h, mv, ms, nu, length);
    const float sun_y = dot3f(end.sun, end.up);
    if (!(sun_y > 0.0f)) return;
    const float mu_sun = clampf(sun_y, 0.0f, 1.0f);
    float ts[NW], tv[NW];
    transmittance_segment(P, 0.0f, mu_sun, distance_to_top(P, 0.0f, mu_sun), ts);
    transmittance_segment(P, h, mv, length, tv);
    for (int w = 0; w < NW; w++) out[w] = tv[w] * (0.0f * P.c.ground_albedo / kPiF + P.c.ground_albedo * mu_sun * ts[w] / kPiF);
}

__global__ void k_bake_transmittance(const BakeParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = P.c.transmittance_mu * P.c.transmittance_height;
    if (i >= n) return;
    const uint32_t mi = i % P.c.transmittance_mu, hi = i / P.c.transmittance_mu;
    const float h = P.H * (float)hi / (float)(P.c.transmittance_height - 1u);
    const float mu = -1.0f + 2.0f * (float)mi / (float)(P.c.transmittance_mu - 1u);
    float s[NW];
    transmittance_segment(P, h, mu, distance_to_boundary(P, h, mu), s);
    store_rgba(P, s, P.t_out + 4u * i);
}

__global__ void k_bake_aerial(const BakeParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = P.c.aerial_distance * P.c.aerial_mu_view * P.c.aerial_height;
    if (i >= n) return;
    const uint32_t di = i % P.c.aerial_distance, vi = (i / P.c.aerial_distance) % P.c.aerial_mu_view, hi = i / (P.c.aerial_distance * P.c.aerial_mu_view);
    const float h = P.H * (float)hi / (float)(P.c.aerial_height - 1u);
    const float mu = -1.0f + 2.0f * (float)vi / (float)(P.c.aerial_mu_view - 1u);
    const float distance = P.c.max_aerial_distance_m * (float)di / (float)(P.c.aerial_distance - 1u);
    float t[NW], sum = 0.0f;
    transmittance_segment(P, h, mu, fminf(distance_to_boundary(P, h, mu), distance), t);
    for (int w = 0; w < NW; w++) sum += t[w];
    uint16_t *o = P.aerial_out + 4u * i;
    o[0] = o[1] = o[2] = 0u;
    o[3] = half_bits(sum / (float)NW);
}

// integrate_single_scattering (bake.rs:875-915) + the direct-sun ground term: one lane per entry
__global__ void k_bake_single(const BakeParams P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.count) return;
    float h, nu, ms, mv;
    entry_coordinates(P, e, h, nu, ms, mv);
    float radiance[NW];
    for (int w = 0; w < NW; w++) radiance[w] = 0.0f;
    const float length = distance_to_boundary(P, h, mv);
    if (length > 0.0f) {
        const float ds = length / 64.0f;
        const float pr = rayleigh_phase(nu), pm = mie_phase(nu, P.c.mie_g);
        float view_columns[3] = {0.0f, 0.0f, 0.0f};
        for (int i = 0; i < 64; i++) {
            const Geom g = ray_geometry(P, h, mv, ms, nu, ((float)i + 0.5f) * ds);
            float rho[3], unused;
            density_at(P, g.altitude_m, rho);
            if (!distance_to_ground(P, g.altitude_m, g.mu_sun, unused)) {
                float sun_columns[3];
                optical_columns(P, g.altitude_m, g.mu_sun, distance_to_top(P, g.altitude_m, g.mu_sun), sun_columns);
                for (int w = 0; w < NW; w++) {
                    const float scatter = P.rayleigh[w] * rho[0] * pr + P.mie_sca[w] * rho[1] * pm;
                    radiance[w] += transmittance_at(P, view_columns, w) * transmittance_at(P, sun_columns, w) * scatter *
                                   cell_length(extinction_at(P, rho, w), ds);
                }
            }
            for (int k = 0; k < 3; k++) view_columns[k] += rho[k] * ds;
        }
    }
    float ground[NW];
    direct_ground(P, h, mv, ms, nu, ground);
    for (int w = 0; w < NW; w++) {
        P.single[(size_t)e * NW + w] = radiance[w];
        P.accumulated[(size_t)e * NW + w] = radiance[w];
        P.previous[(size_t)e * NW + w] = radiance[w] + ground[w];
    }
}

// sample_spectral_scattering (bake.rs:1225-1270): quadrilinear fetch of the previous order
__device__ void sample_previous(const BakeParams &P, float h, float mu_sun, float mu_view, float nu, float *out) {
    const float p[4] = {height_to_unit(h, P.H) * (float)(P.c.scattering_height - 1u), nu_to_unit(nu) * (float)(P.c.scattering_nu - 1u),
                        mu_to_unit(mu_sun) * (float)(P.c.scattering_mu_sun - 1u), mu_to_unit(mu_view) * (float)(P.c.scattering_mu_view - 1u)};
    const uint32_t ext[4] = {P.c.scattering_height, P.c.scattering_nu, P.c.scattering_mu_sun, P.c.scattering_mu_view};
    uint32_t lo[4], hi[4];
    float f[4];
    for (int a = 0; a < 4; a++) {
        lo[a] = (uint32_t)floorf(p[a]);
        hi[a] = lo[a] + 1u < ext[a] - 1u ? lo[a] + 1u : ext[a] - 1u;
        f[a] = p[a] - (float)lo[a];
    }
    for (int k = 0; k < NW; k++) out[k] = 0.0f;
    for (int corner = 0; corner < 16; corner++) {  // hs, ns, ss, vs nested in the reference's order
        const int sides[4] = {(corner >> 3) & 1, (corner >> 2) & 1, (corner >> 1) & 1, corner & 1};
        float w = 1.0f;
        uint32_t i[4];
        for (int a = 0; a < 4; a++) {
            i[a] = sides[a] == 0 ? lo[a] : hi[a];
            w *= sides[a] == 0 ? 1.0f - f[a] : f[a];
        }
        const float *q = P.previous + (size_t)((((i[0] * P.c.scattering_nu + i[1]) * P.c.scattering_mu_sun + i[2]) * P.c.scattering_mu_view) + i[3]) * NW;
        for (int k = 0; k < NW; k++) out[k] += w * q[k];
    }
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// integrate_scattering_order (bake.rs:1400-1463): one wave per entry, the 512 directions 8 to a lane
__global__ __launch_bounds__(64) void k_bake_order(const BakeParams P) {
    const uint32_t e = blockIdx.x, lane = threadIdx.x;
    float h, nu, ms, mv;
    entry_coordinates(P, e, h, nu, ms, mv);
    float volume[NW];
    for (int w = 0; w < NW; w++) volume[w] = 0.0f;
    const float length = distance_to_boundary(P, h, mv);
    float ground_length = 0.0f;
    const bool ground_bound = distance_to_ground(P, h, mv, ground_length);
    if (length > 0.0f) {
        float columns[3] = {0.0f, 0.0f, 0.0f};
        for (int i = 0; i < kOrderSteps; i++) {
            const float u0 = (float)i / (float)kOrderSteps, u1 = (float)(i + 1) / (float)kOrderSteps;
            const float start = ground_bound ? length * (1.0f - (1.0f - u0) * (1.0f - u0)) : length * u0 * u0;
            const float end = ground_bound ? length * (1.0f - (1.0f - u1) * (1.0f - u1)) : length * u1 * u1;
            const float ds = end - start, distance = 0.5f * (start + end);
            const Geom g = ray_geometry(P, h, mv, ms, nu, distance);
            float rho[3];
            density_at(P, g.altitude_m, rho);
            // phase_quadrature_normalization (:1286-1298)
            float n0 = 0.0f, n1 = 0.0f;
            for (int k = 0; k < NQ / 64; k++) {
                const float4 q = P.quad[lane + 64 * k];
                const float incoming[3] = {g.tangent[0] * q.x + g.up[0] * q.y, g.tangent[1] * q.x + g.up[1] * q.y, q.z};
                const float cosine = dot3f(incoming, g.outgoing);
                n0 += rayleigh_phase(cosine) * q.w;
                n1 += mie_phase(cosine, P.c.mie_g) * q.w;
            }
            n0 = fmaxf(wave_sum(n0), 1.0e-8f);
            n1 = fmaxf(wave_sum(n1), 1.0e-8f);
            float source[NW];
            for (int w = 0; w < NW; w++) source[w] = 0.0f;
            for (int k = 0; k < NQ / 64; k++) {
                const float4 q = P.quad[lane + 64 * k];
                const float incoming[3] = {g.tangent[0] * q.x + g.up[0] * q.y, g.tangent[1] * q.x + g.up[1] * q.y, q.z};
                float l[NW];
                sample_previous(P, g.altitude_m, g.mu_sun, dot3f(incoming, g.up), dot3f(incoming, g.sun), l);
                const float cosine = dot3f(incoming, g.outgoing);
                const float phr = rayleigh_phase(cosine), phm = mie_phase(cosine, P.c.mie_g);
                for (int w = 0; w < NW; w++) {
                    const float scatter = P.rayleigh[w] * rho[0] * phr / n0 + P.mie_sca[w] * rho[1] * phm / n1;
                    source[w] += scatter * l[w] * q.w;
                }
            }
            for (int w = 0; w < NW; w++) {
                const float s = wave_sum(source[w]);
                volume[w] += transmittance_at(P, columns, w) * s * cell_length(extinction_at(P, rho, w), ds);
            }
            for (int k = 0; k < 3; k++) columns[k] += rho[k] * ds;
        }
    }
    // ground_boundary_along_ray with the previous order as the incident field (:1300-1372)
    float boundary[NW];
    for (int w = 0; w < NW; w++) boundary[w] = 0.0f;
    if (ground_bound && P.c.ground_albedo > 0.0f) {
        const Geom end = ray_geometry(P, h, mv, ms, nu, ground_length);
        const float sun_local[3] = {dot3f(end.sun, end.tangent), dot3f(end.sun, end.up), end.sun[2]};
        float irradiance[NW];
        for (int w = 0; w < NW; w++) irradiance[w] = 0.0f;
        for (int k = 0; k < NQ / 64; k++) {
            const float4 q = P.quad[lane + 64 * k];
            if (q.y <= 0.0f) continue;
            const float local[3] = {q.x, q.y, q.z};
            float sample[NW];
            sample_previous(P, 0.0f, clampf(sun_local[1], -1.0f, 1.0f), q.y, dot3f(local, sun_local), sample);
            for (int w = 0; w < NW; w++) irradiance[w] += sample[w] * q.y * q.w * P.cos_norm;
        }
        float tv[NW];
        transmittance_segment(P, h, mv, ground_length, tv);
        for (int w = 0; w < NW; w++) boundary[w] = tv[w] * (wave_sum(irradiance[w]) * P.c.ground_albedo / kPiF);
    }
    if (lane == 0u) {
        for (int w = 0; w < NW; w++) {
            P.next[(size_t)e * NW + w] = volume[w] + boundary[w];
            P.accumulated[(size_t)e * NW + w] += volume[w];
        }
    }
}

// mean |field| over the table in table order (bake.rs:1547-1553, 1590-1596): one thread, the reference's order
__global__ void k_bake_delta(const BakeParams P, const float *field) {
    if (blockIdx.x != 0u || threadIdx.x != 0u) return;
    float total = 0.0f;
    for (size_t i = 0; i < (size_t)P.count * NW; i++) total += fabsf(field[i]);
    P.deltas[P.order - 1u] = total / (float)((size_t)P.count * NW);
}

__global__ void k_bake_store(const BakeParams P) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.count) return;
    float s[NW], t[NW];
    for (int w = 0; w < NW; w++) {
        s[w] = P.single[(size_t)e * NW + w];
        t[w] = P.accumulated[(size_t)e * NW + w];
    }
    store_rgba(P, s, P.single_out + 4u * e);
    store_rgba(P, t, P.acc_out + 4u * e);
}

void hip_ok(hipError_t e, const char *what) {
    if (e != hipSuccess) fail(F3D_STATUS_DEVICE, "HIP failure in %s: %s", what, hipGetErrorString(e));
}

void validate(const f3d_aether_bake_config &c) {  // AtmosphereConfig::validate + LutDimensions::validate, bake.rs:60-85,164-229
    const float scalars[9] = {c.turbidity, c.ozone_du, c.mie_g, c.bottom_radius_m, c.top_radius_m, c.rayleigh_scale_height_m,
                              c.mie_scale_height_m, c.max_aerial_distance_m, c.ground_albedo};
    for (float v : scalars)
        if (!std::isfinite(v)) fail(F3D_STATUS_VALUE, "all scalar parameters must be finite");
    if (!(c.turbidity >= 1.0f && c.turbidity <= 10.0f)) fail(F3D_STATUS_VALUE, "turbidity must be in [1, 10]");
    if (!(c.ozone_du >= 0.0f && c.ozone_du <= 600.0f)) fail(F3D_STATUS_VALUE, "ozone must be in [0, 600] DU");
    if (!(c.mie_g >= 0.0f && c.mie_g <= 0.99f)) fail(F3D_STATUS_VALUE, "mie_g must be in [0, 0.99]");
    if (c.bottom_radius_m <= 0.0f || c.top_radius_m <= c.bottom_radius_m) fail(F3D_STATUS_VALUE, "top radius must exceed a positive bottom radius");
    if (c.rayleigh_scale_height_m <= 0.0f || c.mie_scale_height_m <= 0.0f || c.max_aerial_distance_m <= 0.0f)
        fail(F3D_STATUS_VALUE, "scale heights and aerial distance must be positive");
    if (!(c.ground_albedo >= 0.0f && c.ground_albedo <= 1.0f)) fail(F3D_STATUS_VALUE, "ground albedo must be in [0, 1]");
    if (c.scattering_orders < 2u || c.scattering_orders > 8u) fail(F3D_STATUS_VALUE, "scattering_orders must be in [2, 8]");
    const uint32_t axes[9] = {c.transmittance_mu, c.transmittance_height, c.scattering_mu_view, c.scattering_mu_sun, c.scattering_height,
                              c.scattering_nu, c.aerial_distance, c.aerial_mu_view, c.aerial_height};
    for (uint32_t a : axes) {
        if (a < 2u) fail(F3D_STATUS_VALUE, "every atmosphere LUT axis must contain at least two samples");
        if (a > 256u) fail(F3D_STATUS_VALUE, "atmosphere LUT axes are capped at 256 samples");
    }
}

}  // namespace

extern "C" int f3d_aether_bake(const f3d_aether_bake_config *config, uint16_t *transmittance, uint16_t *single_scattering,
                               uint16_t *accumulated_scattering, uint16_t *aerial, float *order_deltas, double *seconds, char *err,
                               size_t errlen) {
    if (err && errlen) err[0] = 0;
    std::vector<void *> owned;
    int rc = F3D_STATUS_OK;
    try {
        if (!config || !transmittance || !single_scattering || !accumulated_scattering || !aerial || !order_deltas)
            fail(F3D_STATUS_VALUE, "null argument");
        validate(*config);
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            fail(F3D_STATUS_DEVICE, "no HIP device available: libf3dhip has no CPU fallback");
        const f3d_aether_bake_config &c = *config;
        BakeParams P{};
        P.c = c;
        P.H = c.top_radius_m - c.bottom_radius_m;
        const float wl[NW] = {380.0f, 420.0f, 460.0f, 500.0f, 540.0f, 580.0f, 620.0f, 660.0f, 700.0f, 740.0f, 780.0f};
        for (int w = 0; w < NW; w++) {  // spectral.rs:62-92, bake.rs:234-241,796-798
            const float x = 550.0f / wl[w], x2 = x * x;
            P.rayleigh[w] = (5.10e-31f * (x2 * x2)) * 2.546899e25f;
            P.mie_ext[w] = (1.0e-5f * c.turbidity) * std::pow(550.0f / wl[w], 1.0f);
            P.mie_sca[w] = P.mie_ext[w] * 0.9f;
            const float t = (wl[w] - 600.0f) / 85.0f;
            P.ozone[w] = 1.2e-6f * std::exp(-0.5f * (t * t));
        }
        {  // white point of the 11-sample basis
            const float cie[NW][3] = {{0.001368f, 0.000039f, 0.006450f}, {0.134380f, 0.004000f, 0.645600f}, {0.290800f, 0.060000f, 1.669200f},
                                      {0.004900f, 0.323000f, 0.272000f}, {0.290400f, 0.954000f, 0.020300f}, {0.916300f, 0.870000f, 0.001650f},
                                      {0.854450f, 0.381000f, 0.000190f}, {0.164900f, 0.061000f, 0.000000f}, {0.011359f, 0.004102f, 0.000000f},
                                      {0.000690f, 0.000249f, 0.000000f}, {0.000042f, 0.000015f, 0.000000f}};
            float xyz[3] = {0.0f, 0.0f, 0.0f};
            for (int i = 0; i < NW; i++)
                for (int k = 0; k < 3; k++) xyz[k] += 1.0f * cie[i][k] * ((i == 0 || i + 1 == NW) ? 0.5f : 1.0f);
            P.white[0] = 3.2404542f * xyz[0] + -1.5371385f * xyz[1] + -0.4985314f * xyz[2];
            P.white[1] = -0.969266f * xyz[0] + 1.8760108f * xyz[1] + 0.041556f * xyz[2];
            P.white[2] = 0.0556434f * xyz[0] + -0.2040259f * xyz[1] + 1.0572252f * xyz[2];
        }
        // 16-point Gauss-Legendre in mu x 32 equal azimuths (bake.rs:1137-1215)
        const float gl[8][3] = {{0.09501251f, 0.99547607f, 0.1894506f}, {0.28160354f, 0.95953083f, 0.18260342f}, {0.45801678f, 0.88894355f, 0.16915652f},
                                {0.61787623f, 0.7862754f, 0.14959599f}, {0.7554044f, 0.6552589f, 0.12462897f}, {0.8656312f, 0.5006822f, 0.09515851f},
                                {0.944575f, 0.32829565f, 0.062253524f}, {0.9894009f, 0.14520948f, 0.02715246f}};
        const float az8[8][2] = {{0.9951847f, 0.09801714f}, {0.95694035f, 0.29028466f}, {0.8819213f, 0.47139674f}, {0.77301043f, 0.6343933f},
                                 {0.6343933f, 0.77301043f}, {0.47139674f, 0.8819213f}, {0.29028466f, 0.95694035f}, {0.09801714f, 0.9951847f}};
        std::vector<float4> quad(NQ);
        float cos_sum = 0.0f;
        for (int i = 0; i < 16; i++) {
            const float *node = gl[i < 8 ? 7 - i : i - 8];
            const float mu = i < 8 ? -node[0] : node[0];
            for (int a = 0; a < 32; a++) {
                const int k = a & 7, quadrant = a >> 3;
                const float *p = az8[(quadrant & 1) ? 7 - k : k];
                const float cx = (quadrant == 1 || quadrant == 2) ? -p[0] : p[0], sy = quadrant >= 2 ? -p[1] : p[1];
                quad[i * 32 + a] = float4{node[1] * cx, mu, node[1] * sy, node[2] * (6.28318530717958647692f / 32.0f)};
            }
        }
        for (int q = 0; q < NQ; q++)
            if (quad[q].y > 0.0f) cos_sum += quad[q].y * quad[q].w;
        P.cos_norm = kPiF / cos_sum;

        auto alloc = [&](size_t bytes, const char *what) {
            void *p = nullptr;
            hip_ok(device_alloc(&p, bytes), what);
            owned.push_back(p);
            return p;
        };
        P.count = c.scattering_mu_view * c.scattering_mu_sun * c.scattering_height * c.scattering_nu;
        const size_t field = (size_t)P.count * NW * sizeof(float);
        float4 *d_quad = (float4 *)alloc(NQ * sizeof(float4), "quadrature");
        hip_ok(hipMemcpy(d_quad, quad.data(), NQ * sizeof(float4), hipMemcpyHostToDevice), "quadrature upload");
        P.quad = d_quad;
        P.previous = (float *)alloc(field, "previous order");
        P.next = (float *)alloc(field, "next order");
        P.accumulated = (float *)alloc(field, "accumulated scattering");
        P.single = (float *)alloc(field, "single scattering");
        const size_t nt = (size_t)c.transmittance_mu * c.transmittance_height, na = (size_t)c.aerial_distance * c.aerial_mu_view * c.aerial_height;
        P.t_out = (uint16_t *)alloc(nt * 8, "transmittance table");
        P.single_out = (uint16_t *)alloc((size_t)P.count * 8, "single table");
        P.acc_out = (uint16_t *)alloc((size_t)P.count * 8, "accumulated table");
        P.aerial_out = (uint16_t *)alloc(na * 8, "aerial table");
        P.deltas = (float *)alloc(8 * sizeof(float), "deltas");
        hipEvent_t e0, e1;
        hip_ok(hipEventCreate(&e0), "event");
        hip_ok(hipEventCreate(&e1), "event");
        hip_ok(hipEventRecord(e0, nullptr), "event");
        hipLaunchKernelGGL(k_bake_transmittance, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, nullptr, P);
        hipLaunchKernelGGL(k_bake_aerial, dim3((unsigned)((na + 63) / 64)), dim3(64), 0, nullptr, P);
        hipLaunchKernelGGL(k_bake_single, dim3((P.count + 63u) / 64u), dim3(64), 0, nullptr, P);
        P.order = 1u;
        hipLaunchKernelGGL(k_bake_delta, dim3(1), dim3(64), 0, nullptr, P, (const float *)P.previous);
        for (uint32_t order = 2u; order <= c.scattering_orders; order++) {
            P.order = order;
            hipLaunchKernelGGL(k_bake_order, dim3(P.count), dim3(64), 0, nullptr, P);
            hipLaunchKernelGGL(k_bake_delta, dim3(1), dim3(64), 0, nullptr, P, (const float *)P.next);
            std::swap(P.previous, P.next);
        }
        hipLaunchKernelGGL(k_bake_store, dim3((P.count + 63u) / 64u), dim3(64), 0, nullptr, P);
        hip_ok(hipGetLastError(), "bake kernels");
        hip_ok(hipEventRecord(e1, nullptr), "event");
        hip_ok(hipDeviceSynchronize(), "bake");
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (seconds) *seconds = ms * 1e-3;
        hip_ok(hipMemcpy(transmittance, P.t_out, nt * 8, hipMemcpyDeviceToHost), "readback");
        hip_ok(hipMemcpy(single_scattering, P.single_out, (size_t)P.count * 8, hipMemcpyDeviceToHost), "readback");
        hip_ok(hipMemcpy(accumulated_scattering, P.acc_out, (size_t)P.count * 8, hipMemcpyDeviceToHost), "readback");
        hip_ok(hipMemcpy(aerial, P.aerial_out, na * 8, hipMemcpyDeviceToHost), "readback");
        hip_ok(hipMemcpy(order_deltas, P.deltas, c.scattering_orders * sizeof(float), hipMemcpyDeviceToHost), "readback");
    } catch (const Failure &f) {
        rc = report(f, err, errlen);
    } catch (const std::exception &e) {
        if (err && errlen) snprintf(err, errlen, "host failure: %s", e.what());
        rc = F3D_STATUS_DEVICE;
    } catch (...) {
        rc = F3D_STATUS_DEVICE;
    }
    for (void *p : owned) (void)device_free(p);
    return rc;
}
