/* oracle/aether_ref_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked or imported by the product).
 *
 * Restates the reference's acceptance-only stochastic spectral atmosphere tracer in its own shape: one loop nest per
 * pixel over samples and wavelengths, sums in the loop (the product runs every wavelength path as its own lane and folds
 * afterwards: forge3d_amd/csrc/f3d_aether_ref.h).
 *   main_aether_spectral_reference          src/shaders/atmosphere/prometheus_spectral_reference.wgsl:423-480
 *   aether_ref_trace_wavelength             :299-421      aether_ref_boundary               :132-152
 *   aether_ref_transmittance_to_sun         :160-188      phases / samplers / basis / RR     :190-297
 *   constants, spectra, density, roots      :8-130
 *   render_aether_spectral_reference        src/path_tracing/hybrid_compute/aether_reference.rs:203-560 (uniforms,
 *                                           seed mixing, finalisation, variance, converged)
 *   validate_desc                           aether_reference.rs:83-153
 * Camera rays, terrain hits and sun visibility go through f3do_terrain_trace (oracle/f3d_oracle.c, the pinned
 * restatement of terrain_trace), RNG and tent filter as in the terrain tracer (hybrid_kernel.wgsl:78-85,
 * hybrid_terrain_traversal.wgsl:409-414).
 * PARITY PIN: the tracer is stochastic and its transcendentals are whatever the GPU's are, so the reference has no
 * golden output; it is pinned here by the reference's own tests restated as properties (tests/
 * test_atmosphere_pt_reference.py:114-205, aether_reference.rs tests) and by its acceptance gate against the LUT
 * transport whose oracle IS pinned (tests/test_atmosphere_reference.py:249-400: CIEDE2000 < 2 over the sun-elevation
 * sweep) -- tests/test_aether_ref.py.  "parity pinned by KAT properties and the acceptance gate only" (DESIGN.md).
 * Arithmetic: IEEE f32, no contraction; exp / log / sin / cos are fixed polynomials restated from f3d_math.h. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern void *f3do_terrain_open(const float *heights, uint32_t w, uint32_t h, float spacing_x, float spacing_z, float exaggeration);
extern void f3do_terrain_close(void *handle);
extern int f3do_terrain_trace(const void *handle, const float *o, float tmin, const float *d, float tmax, int32_t any_hit, float *t_out, float *n_out);

typedef struct { float x, y, z; } v3;
static v3 mk(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 scale(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static v3 cross3(v3 a, v3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static float len3(v3 a) { return sqrtf(dot3(a, a)); }
static v3 norm3(v3 a) { return scale(a, 1.0f / sqrtf(dot3(a, a))); }
static v3 along(v3 o, float t, v3 d) { return mk(fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), fmaf(t, d.z, o.z)); }
static v3 combine(float a, v3 x, float b, v3 y, float c, v3 z) {
    return mk(fmaf(c, z.x, fmaf(b, y.x, a * x.x)), fmaf(c, z.y, fmaf(b, y.y, a * x.y)), fmaf(c, z.z, fmaf(b, y.z, a * x.z)));
}
static float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
static float bits_f(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t f_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

#define PI_F 3.14159265358979323846f
#define HALF_PI_F 1.57079632679489661923f
static float exp_fixed(float x) {
    if (x > 88.0f) return INFINITY;
    if (x < -103.0f) return 0.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    const float z = r * r;
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float y = fmaf(p, z, r) + 1.0f;
    const int e = (int)n;
    if (e < -126) return (y * bits_f((uint32_t)(e + 64 + 127) << 23)) * 5.42101086242752217e-20f;
    return y * bits_f((uint32_t)(e + 127) << 23);
}
static float log_fixed(float x) {
    const uint32_t b = f_bits(x);
    int e = (int)(b >> 23) - 126;
    float m = bits_f((b & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    const float z = m * m;
    float p = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    const float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(0.693359375f, fe, m + y);
}
static void sincos_turn(float u, float *s_out, float *c_out) { /* sin, cos of 2 pi u */
    float a = 4.0f * u, k = rintf(a), x = (a - k) * HALF_PI_F, z = x * x;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = fmaf(ps * z, x, x);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
    int q = ((int)k) & 3;
    *s_out = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    *c_out = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}
static float xorshift32(uint32_t *state) { /* hybrid_kernel.wgsl:78-85 */
    uint32_t x = *state;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    *state = x;
    return (float)x / 4294967296.0f;
}
static float tent_offset(float u) { /* hybrid_terrain_traversal.wgsl:409-414 */
    if (u < 0.5f) return sqrtf(2.0f * u) - 1.0f;
    return 1.0f - sqrtf(2.0f * (1.0f - u));
}

#define BOTTOM_R 6360000.0f
#define TOP_R 6460000.0f
typedef struct {
    const void *terrain;
    v3 cam_origin, cam_right, cam_up, cam_forward;
    float half_w, half_h;
    uint32_t width, height, seed_hi, seed_lo, frame_index;
    v3 sun_direction;
    float sun_radiance, turbidity, mie_g, ozone_scale, ground_albedo;
} scene_t;

static const float WAVELENGTH_NM[11] = {380.0f, 420.0f, 460.0f, 500.0f, 540.0f, 580.0f, 620.0f, 660.0f, 700.0f, 740.0f, 780.0f};
static const float CIE_XYZ[11][3] = {{0.001368f, 0.000039f, 0.006450f}, {0.134380f, 0.004000f, 0.645600f}, {0.290800f, 0.060000f, 1.669200f},
                                     {0.004900f, 0.323000f, 0.272000f}, {0.290400f, 0.954000f, 0.020300f}, {0.916300f, 0.870000f, 0.001650f},
                                     {0.854450f, 0.381000f, 0.000190f}, {0.164900f, 0.061000f, 0.000000f}, {0.011359f, 0.004102f, 0.000000f},
                                     {0.000690f, 0.000249f, 0.000000f}, {0.000042f, 0.000015f, 0.000000f}};

static float rayleigh_beta(float nm) { float q = 550.0f / nm, q2 = q * q; return (float)(5.10e-31 * 2.546899e25) * (q2 * q2); }
static float mie_extinction(const scene_t *S, float nm) { return (1.0e-5f * S->turbidity) * (550.0f / nm); }
static float ozone_absorption(float nm) { float delta = (nm - 600.0f) / 85.0f; return 1.2e-6f * exp_fixed((-0.5f * delta) * delta); }
static v3 planet_center(void) { return mk(0.0f, -BOTTOM_R, 0.0f); }
static void density(const scene_t *S, v3 p, float *rayleigh, float *mie, float *ozone) {
    float altitude = fmaxf(len3(sub(p, planet_center())) - BOTTOM_R, 0.0f);
    *rayleigh = exp_fixed(-altitude / 8000.0f);
    *mie = exp_fixed(-altitude / 1200.0f);
    *ozone = fmaxf(1.0f - fabsf((altitude - 25000.0f) / 15000.0f), 0.0f) * S->ozone_scale;
}
static void sphere_roots(v3 o, v3 d, float radius, float *r0, float *r1) {
    v3 oc = sub(o, planet_center());
    float b = dot3(oc, d), c = dot3(oc, oc) - radius * radius, disc = b * b - c;
    if (disc < 0.0f) { *r0 = *r1 = 1e30f; return; }
    float root = sqrtf(disc);
    *r0 = -b - root; *r1 = -b + root;
}
static float positive_root(float r0, float r1) { return r0 > 1e-3f ? r0 : (r1 > 1e-3f ? r1 : 1e30f); }
static float extinction(const scene_t *S, float nm, float dr, float dm, float dz) {
    return (rayleigh_beta(nm) * dr + mie_extinction(S, nm) * dm) + ozone_absorption(nm) * dz;
}
static float transmittance_to_sun(const scene_t *S, v3 position, float nm) {
    v3 sun = S->sun_direction, o = along(position, 1e-2f, sun);
    float t0, t1, g0, g1;
    sphere_roots(o, sun, TOP_R, &t0, &t1);
    float top_t = t1 > 1e-3f ? t1 : t0;
    sphere_roots(o, sun, BOTTOM_R, &g0, &g1);
    if (positive_root(g0, g1) < top_t) return 0.0f;
    { /* intersect_shadow_ray(shadow_ray, top_t): any terrain hit (all of it lies inside the top sphere) */
        float oo[3] = {o.x, o.y, o.z}, dd[3] = {sun.x, sun.y, sun.z};
        if (f3do_terrain_trace(S->terrain, oo, 1e-3f, dd, 1e30f, 1, NULL, NULL)) return 0.0f;
    }
    float step_length = top_t / 64.0f, optical_depth = 0.0f;
    for (uint32_t step = 0; step < 64u; step++) {
        float t = ((float)step + 0.5f) * step_length, dr, dm, dz;
        density(S, along(o, t, sun), &dr, &dm, &dz);
        optical_depth = optical_depth + extinction(S, nm, dr, dm, dz) * step_length;
    }
    return exp_fixed(-fmaxf(optical_depth, 0.0f));
}
static float rayleigh_phase(float cos_theta) { float c = clampf(cos_theta, -1.0f, 1.0f); return (3.0f * (1.0f + c * c)) / (16.0f * PI_F); }
static float pow15(float v) { return v * sqrtf(v); }
static float mie_phase(float cos_theta, float g) {
    float c = clampf(cos_theta, -1.0f, 1.0f), gg = clampf(g, -0.999f, 0.999f);
    float denominator = pow15(fmaxf((1.0f + gg * gg) - (2.0f * gg) * c, 1e-6f));
    return ((3.0f * (1.0f - gg * gg)) * (1.0f + c * c)) / (((8.0f * PI_F) * (2.0f + gg * gg)) * denominator);
}
static v3 basis_direction(v3 axis, float cosine, float u) { /* phi = 2 pi u */
    v3 n = norm3(axis);
    float sign = n.z < 0.0f ? -1.0f : 1.0f, a = -1.0f / (sign + n.z), b = (n.x * n.y) * a;
    v3 tangent = mk(1.0f + ((sign * n.x) * n.x) * a, sign * b, -sign * n.x), bitangent = mk(b, sign + (n.y * n.y) * a, -n.y);
    float sine = sqrtf(fmaxf(1.0f - cosine * cosine, 0.0f)), sn, cs;
    sincos_turn(u, &sn, &cs);
    return norm3(combine(cosine, n, sine * cs, tangent, sine * sn, bitangent));
}
static int russian_roulette(float *throughput, uint32_t depth, uint32_t *state) {
    if (depth < 3u) return 1;
    float survival = clampf(*throughput, 0.1f, 0.95f);
    if (xorshift32(state) > survival) return 0;
    *throughput = *throughput / survival;
    return 1;
}

static float trace_wavelength(const scene_t *S, v3 cam_o, v3 cam_d, float nm, uint32_t *state) {
    v3 ro = cam_o, rd = cam_d;
    float throughput = 1.0f, radiance = 0.0f;
    float beta_rayleigh = rayleigh_beta(nm), beta_mie_ext = mie_extinction(S, nm), beta_mie_sca = beta_mie_ext * 0.9f;
    float beta_ozone = ozone_absorption(nm), majorant = (beta_rayleigh + beta_mie_ext) + beta_ozone * S->ozone_scale;
    for (uint32_t depth = 0; depth < 6u; depth++) {
        /* aether_ref_boundary */
        float t0, t1, g0, g1, bt, th = 0.0f, tn[3] = {0, 0, 0};
        uint32_t kind = 0u;
        sphere_roots(ro, rd, TOP_R, &t0, &t1);
        bt = t1 > 1e-3f ? t1 : t0;
        float oo[3] = {ro.x, ro.y, ro.z}, dd[3] = {rd.x, rd.y, rd.z};
        int hit = f3do_terrain_trace(S->terrain, oo, 1e-3f, dd, 1e30f, 0, &th, tn);
        sphere_roots(ro, rd, BOTTOM_R, &g0, &g1);
        float ground_t = positive_root(g0, g1);
        if (ground_t < bt) { bt = ground_t; kind = 2u; }
        if (hit && th < bt) { bt = th; kind = 1u; }
        if (!(bt > 1e-3f) || !(bt < 1e29f)) return radiance;

        float travelled = 0.0f;
        uint32_t scatter_kind = 0u, null_count = 0u;
        v3 scatter_position = mk(0, 0, 0);
        for (;;) {
            if (null_count >= 2048u) return bits_f(0x7fc00000u);
            null_count++;
            float free_flight = -log_fixed(fmaxf(1.0f - xorshift32(state), 1e-7f)) / fmaxf(majorant, 1e-12f);
            if (travelled + free_flight >= bt) break;
            travelled = travelled + free_flight;
            scatter_position = along(ro, travelled, rd);
            float dr, dm, dz;
            density(S, scatter_position, &dr, &dm, &dz);
            float sigma_rayleigh = beta_rayleigh * dr, sigma_mie_sca = beta_mie_sca * dm, sigma_mie_abs = (beta_mie_ext - beta_mie_sca) * dm;
            float sigma_ozone = beta_ozone * dz, sigma_total = ((sigma_rayleigh + sigma_mie_sca) + sigma_mie_abs) + sigma_ozone;
            if (xorshift32(state) * majorant >= sigma_total) continue;
            float event = xorshift32(state) * sigma_total;
            scatter_kind = event < sigma_rayleigh ? 1u : (event < sigma_rayleigh + sigma_mie_sca ? 2u : 3u);
            break;
        }
        if (scatter_kind == 3u) return radiance;
        if (scatter_kind != 0u) {
            float cosine_to_sun = dot3(rd, S->sun_direction);
            float phase = scatter_kind == 1u ? rayleigh_phase(cosine_to_sun) : mie_phase(cosine_to_sun, S->mie_g);
            float sun_t = transmittance_to_sun(S, scatter_position, nm);
            radiance = radiance + ((throughput * S->sun_radiance) * phase) * sun_t;
            v3 direction;
            float weight;
            if (scatter_kind == 1u) { /* aether_ref_sample_rayleigh */
                float cosine = 0.0f;
                int accepted = 0;
                for (uint32_t attempt = 0; attempt < 16u; attempt++) {
                    cosine = 2.0f * xorshift32(state) - 1.0f;
                    if (xorshift32(state) <= 0.5f * (1.0f + cosine * cosine)) { accepted = 1; break; }
                }
                weight = 1.0f;
                if (!accepted) {
                    cosine = 2.0f * xorshift32(state) - 1.0f;
                    weight = rayleigh_phase(cosine) / (0.25f / PI_F);
                }
                direction = basis_direction(rd, cosine, xorshift32(state));
            } else { /* aether_ref_sample_mie */
                float g = clampf(S->mie_g, -0.999f, 0.999f), u = xorshift32(state), cosine = 2.0f * u - 1.0f;
                if (fabsf(g) > 1e-3f) {
                    float ratio = (1.0f - g * g) / ((1.0f - g) + (2.0f * g) * u);
                    cosine = clampf(((1.0f + g * g) - ratio * ratio) / (2.0f * g), -1.0f, 1.0f);
                }
                float hg_pdf = (1.0f - g * g) / ((4.0f * PI_F) * pow15(fmaxf((1.0f + g * g) - (2.0f * g) * cosine, 1e-6f)));
                direction = basis_direction(rd, cosine, xorshift32(state));
                weight = mie_phase(cosine, g) / fmaxf(hg_pdf, 1e-12f);
            }
            throughput = throughput * weight;
            ro = along(scatter_position, 1e-2f, direction);
            rd = direction;
            if (!russian_roulette(&throughput, depth + 1u, state)) return radiance;
            continue;
        }
        if (kind == 0u) return radiance;
        v3 surface_position = along(ro, bt, rd);
        v3 normal = norm3(sub(surface_position, planet_center()));
        if (kind == 1u) normal = mk(tn[0], tn[1], tn[2]);
        v3 terrain_origin = along(surface_position, 1e-2f, normal);
        v3 planet_origin = along(planet_center(), BOTTOM_R + 2.0f, normal);
        v3 surface_origin = kind == 2u ? planet_origin : terrain_origin;
        float ndotl = fmaxf(dot3(normal, S->sun_direction), 0.0f);
        if (ndotl > 0.0f) {
            float sun_t = transmittance_to_sun(S, surface_origin, nm);
            radiance = radiance + ((((throughput * S->ground_albedo) * S->sun_radiance) * sun_t) * ndotl) / PI_F;
        }
        throughput = throughput * S->ground_albedo;
        { /* aether_ref_sample_cosine */
            float u1 = xorshift32(state), u2 = xorshift32(state);
            rd = basis_direction(normal, sqrtf(fmaxf(1.0f - u1, 0.0f)), u2);
        }
        ro = surface_origin;
        if (!russian_roulette(&throughput, depth + 1u, state)) return radiance;
    }
    return radiance;
}

typedef struct {
    uint32_t dem_width, dem_height;
    const float *heights;
    float spacing_x, spacing_z, exaggeration;
    float cam_origin[3], cam_look_at[3], cam_up[3], fov_y_deg;
    float sun_azimuth_deg, sun_elevation_deg, sun_intensity;
    float turbidity, ozone_du, mie_g, ground_albedo;
    uint32_t width, height, seed, spp;
    int32_t enabled;
    float variance_threshold;
} aref_desc;

/* Outputs: mean_xyz, linear_rgb (width*height*3), scalars[0] = variance, [1] = converged, hits = primary hits.
 * Returns 0, or 2 with the reference's message in err. */
int aether_ref_oracle_render(const aref_desc *d, float *mean_xyz, float *linear_rgb, float *scalars, uint64_t *hits, char *err, size_t errlen) {
#define INVALID(msg) do { snprintf(err, errlen, "%s", msg); return 2; } while (0)
    if (d->width == 0 || d->height == 0) INVALID("AETHER spectral reference requires non-zero width and height");
    if (d->spp == 0 || d->spp > 4096) INVALID("AETHER spectral reference spp must be in 1..=4096");
    unsigned long long paths = (unsigned long long)d->width * d->height * d->spp * 11ull;
    if (paths > 8000000ull) {
        snprintf(err, errlen, "AETHER spectral reference request has %llu wavelength paths; acceptance lane limit is 8000000", paths);
        return 2;
    }
    if (!(isfinite(d->spacing_x) && d->spacing_x > 0.0f && isfinite(d->spacing_z) && d->spacing_z > 0.0f))
        INVALID("AETHER spectral reference spacing must be finite and positive");
    if (!(isfinite(d->exaggeration) && d->exaggeration > 0.0f)) INVALID("AETHER spectral reference exaggeration must be finite and positive");
    for (int i = 0; i < 3; i++)
        if (!isfinite(d->cam_origin[i]) || !isfinite(d->cam_look_at[i]) || !isfinite(d->cam_up[i]))
            INVALID("AETHER spectral reference camera vectors must be finite");
    v3 origin = mk(d->cam_origin[0], d->cam_origin[1], d->cam_origin[2]);
    v3 fwd = sub(mk(d->cam_look_at[0], d->cam_look_at[1], d->cam_look_at[2]), origin), upv = mk(d->cam_up[0], d->cam_up[1], d->cam_up[2]);
    if (len3(fwd) < 1e-6f || len3(cross3(norm3(fwd), upv)) < 1e-6f) INVALID("AETHER spectral reference camera basis is degenerate");
    float observer_altitude = len3(sub(origin, planet_center())) - BOTTOM_R;
    if (!(observer_altitude >= 0.0f && observer_altitude < 100000.0f)) INVALID("AETHER spectral reference camera must be inside the 0..100 km atmosphere");
    if (!(isfinite(d->fov_y_deg) && d->fov_y_deg > 0.0f && d->fov_y_deg < 180.0f)) INVALID("AETHER spectral reference fov_y_deg must be in (0, 180)");
    if (!(isfinite(d->sun_azimuth_deg) && isfinite(d->sun_elevation_deg) && isfinite(d->sun_intensity) && d->sun_intensity >= 0.0f))
        INVALID("AETHER spectral reference sun inputs must be finite and intensity non-negative");
    const char *problem = NULL;
    if (!(d->turbidity >= 1.0f && d->turbidity <= 10.0f)) problem = "turbidity must be in [1, 10]";
    else if (!(d->ozone_du >= 0.0f && d->ozone_du <= 600.0f)) problem = "ozone must be in [0, 600] DU";
    else if (!(d->mie_g >= 0.0f && d->mie_g <= 0.99f)) problem = "mie_g must be in [0, 0.99]";
    else if (!(d->ground_albedo >= 0.0f && d->ground_albedo <= 1.0f)) problem = "ground_albedo must be in [0, 1]";
    if (problem) { snprintf(err, errlen, "invalid canonical AETHER settings for spectral reference: %s", problem); return 2; }
    if (!(isfinite(d->variance_threshold) && d->variance_threshold > 0.0f)) INVALID("AETHER spectral reference variance_threshold must be finite and positive");

    size_t pixels = (size_t)d->width * d->height;
    scalars[0] = 0.0f; scalars[1] = 1.0f; *hits = 0;
    if (!d->enabled) { /* :430-434 */
        memset(mean_xyz, 0, pixels * 3 * sizeof(float));
        memset(linear_rgb, 0, pixels * 3 * sizeof(float));
        return 0;
    }
    scene_t S;
    memset(&S, 0, sizeof(S));
    void *terrain = f3do_terrain_open(d->heights, d->dem_width, d->dem_height, d->spacing_x, d->spacing_z, d->exaggeration);
    if (!terrain) { snprintf(err, errlen, "heightmap must be at least 2 x 2"); return 3; }
    S.terrain = terrain;
    const float deg = 0.017453292519943295f;
    v3 forward = norm3(fwd), right = norm3(cross3(forward, upv)), up = norm3(cross3(right, forward));
    float az = d->sun_azimuth_deg * deg, el = d->sun_elevation_deg * deg;
    S.cam_origin = origin; S.cam_right = right; S.cam_up = up; S.cam_forward = forward;
    S.half_h = tanf(0.5f * (d->fov_y_deg * deg));
    S.half_w = ((float)d->width / (float)d->height) * S.half_h;
    S.width = d->width; S.height = d->height;
    S.seed_hi = d->seed;
    S.seed_lo = ((d->seed << 16) | (d->seed >> 16)) ^ 0x85EBCA6Bu;
    S.frame_index = 0u;
    S.sun_direction = norm3(mk(cosf(az) * cosf(el), sinf(el), sinf(az) * cosf(el)));
    S.sun_radiance = dot3(mk(d->sun_intensity, d->sun_intensity, d->sun_intensity), mk(0.2126f, 0.7152f, 0.0722f));
    S.turbidity = d->turbidity; S.mie_g = d->mie_g; S.ozone_scale = d->ozone_du / 300.0f; S.ground_albedo = d->ground_albedo;

    float variance = 0.0f;
    int finite = 1;
    uint64_t primary_hits = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(max : variance) reduction(+ : primary_hits) reduction(& : finite)
    for (long pixel = 0; pixel < (long)pixels; pixel++) {
        uint32_t gx = (uint32_t)(pixel % d->width), gy = (uint32_t)(pixel / d->width);
        uint32_t state = S.seed_hi ^ (gx * 1664525u) ^ (gy * 1013904223u) ^ (S.frame_index * 92837111u) ^ S.seed_lo;
        float sum[3] = {0, 0, 0}, mean_y = 0.0f, m2_y = 0.0f;
        uint32_t terrain_primary_hits = 0;
        for (uint32_t sample = 0; sample < d->spp; sample++) {
            float jitter_x = tent_offset(xorshift32(&state)) * 0.5f, jitter_y = tent_offset(xorshift32(&state)) * 0.5f;
            float ndc_x = (((float)gx + 0.5f + jitter_x) / (float)d->width) * 2.0f - 1.0f;
            float ndc_y = (1.0f - ((float)gy + 0.5f + jitter_y) / (float)d->height) * 2.0f - 1.0f;
            v3 direction = norm3(mk(ndc_x * S.half_w, ndc_y * S.half_h, -1.0f));
            direction = norm3(combine(direction.x, S.cam_right, direction.y, S.cam_up, direction.z, scale(S.cam_forward, -1.0f)));
            float oo[3] = {origin.x, origin.y, origin.z}, dd[3] = {direction.x, direction.y, direction.z};
            if (f3do_terrain_trace(terrain, oo, 1e-3f, dd, 1e30f, 0, NULL, NULL)) terrain_primary_hits++;
            float xyz[3] = {0, 0, 0};
            for (uint32_t wi = 0; wi < 11u; wi++) {
                uint32_t wavelength_state = state ^ ((wi + 1u) * 0x9e3779b9u);
                float value = trace_wavelength(&S, origin, direction, WAVELENGTH_NM[wi], &wavelength_state);
                float weight = (wi == 0u || wi + 1u == 11u) ? 0.5f : 1.0f;
                for (int c = 0; c < 3; c++) xyz[c] = xyz[c] + (value * CIE_XYZ[wi][c]) * weight;
                (void)xorshift32(&state);
            }
            for (int c = 0; c < 3; c++) sum[c] = sum[c] + xyz[c];
            float count = (float)(sample + 1u), delta = xyz[1] - mean_y;
            mean_y = mean_y + delta / count;
            m2_y = m2_y + delta * (xyz[1] - mean_y);
        }
        if (!(isfinite(sum[0]) && isfinite(sum[1]) && isfinite(sum[2]) && isfinite(mean_y) && isfinite(m2_y))) finite = 0;
        /* aether_reference.rs:515-532 */
        float inverse_count = 1.0f / (float)d->spp, mean[3] = {sum[0] * inverse_count, sum[1] * inverse_count, sum[2] * inverse_count};
        float rgb[3] = {(3.2404542f * mean[0] - 1.5371385f * mean[1] - 0.4985314f * mean[2]) / 3.2613921f,
                        (-0.9692660f * mean[0] + 1.8760108f * mean[1] + 0.0415560f * mean[2]) / 2.5069624f,
                        (0.0556434f * mean[0] - 0.2040259f * mean[1] + 1.0572252f * mean[2]) / 2.3679786f};
        for (int c = 0; c < 3; c++) { mean_xyz[3 * pixel + c] = mean[c]; linear_rgb[3 * pixel + c] = fmaxf(rgb[c], 0.0f); }
        primary_hits += terrain_primary_hits;
        if (d->spp > 1u) {
            float v = m2_y / ((float)d->spp * (float)(d->spp - 1u));
            if (v > variance) variance = v;
        }
    }
    f3do_terrain_close(terrain);
    if (!finite) { snprintf(err, errlen, "AETHER spectral reference produced non-finite transport output"); return 2; }
    scalars[0] = variance;
    scalars[1] = (d->spp > 1u && variance <= d->variance_threshold) ? 1.0f : 0.0f;
    *hits = primary_hits;
    return 0;
#undef INVALID
}
