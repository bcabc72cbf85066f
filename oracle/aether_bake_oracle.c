/* oracle/aether_bake_oracle.c -- CPU restatement of forge3d's AETHER atmosphere LUT baker (the
 * offline half: the tables the aerial-perspective post reads).  TEST INFRASTRUCTURE ONLY: tests/ may use it as the
 * checker of the HIP baker (csrc/f3d_aether_bake.hip); the product never does.
 *
 * Reference: src/core/atmosphere/bake.rs (bake_atmosphere_luts :1481-1666 and the functions it calls, each cited below)
 * and src/core/atmosphere/spectral.rs (11-wavelength basis, Rayleigh / Mie / ozone coefficients, phase functions,
 * CIE 1931 -> linear sRGB).  Sequential f32 arithmetic in the reference's operation order, glibc expf / expm1f / powf
 * where the reference calls Rust's (which are the platform libm's on Linux), no contraction (-ffp-contract=off).
 *
 * PARITY PIN: the reference's shipped anchors src/core/atmosphere/precomputed/turbidity-{1,2,4,8,10}.bin -- three of
 * them committed as data fixtures under tests/golden/atmosphere/ -- hold exactly these tables (transmittance, single
 * scattering, accumulated scattering, aerial, order deltas: precomputed.rs:14-25) for the default configuration at
 * that turbidity.  tests/test_aether_bake.py compares table by table.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NW 11
static const float WL[NW] = {380.0f, 420.0f, 460.0f, 500.0f, 540.0f, 580.0f, 620.0f, 660.0f, 700.0f, 740.0f, 780.0f};
static const float CIE[NW][3] = {{0.001368f, 0.000039f, 0.006450f}, {0.134380f, 0.004000f, 0.645600f}, {0.290800f, 0.060000f, 1.669200f},
                                 {0.004900f, 0.323000f, 0.272000f}, {0.290400f, 0.954000f, 0.020300f}, {0.916300f, 0.870000f, 0.001650f},
                                 {0.854450f, 0.381000f, 0.000190f}, {0.164900f, 0.061000f, 0.000000f}, {0.011359f, 0.004102f, 0.000000f},
                                 {0.000690f, 0.000249f, 0.000000f}, {0.000042f, 0.000015f, 0.000000f}};
static const float XYZ2RGB[3][3] = {{3.2404542f, -1.5371385f, -0.4985314f}, {-0.969266f, 1.8760108f, 0.041556f}, {0.0556434f, -0.2040259f, 1.0572252f}};
#define PI_F 3.14159265358979323846f
#define TAU_F 6.28318530717958647692f

typedef struct { /* AtmosphereConfig + LutDimensions, bake.rs:32-42,132-144 */
    float turbidity, ozone_du, mie_g, bottom_radius_m, top_radius_m, rayleigh_scale_height_m, mie_scale_height_m,
        max_aerial_distance_m, ground_albedo;
    uint32_t scattering_orders;
    uint32_t transmittance_mu, transmittance_height, scattering_mu_view, scattering_mu_sun, scattering_height, scattering_nu,
        aerial_distance, aerial_mu_view, aerial_height;
} abo_config;

typedef struct { float v[NW]; } spec;

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float atmosphere_height(const abo_config *c) { return c->top_radius_m - c->bottom_radius_m; }
static inline float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; } /* :776 */

/* spectral.rs:62-92 (mie(): bake.rs:234-241) */
static inline float mie_extinction(const abo_config *c, float w) { return (1.0e-5f * c->turbidity) * powf(550.0f / w, 1.0f); }
static inline float mie_scattering(const abo_config *c, float w) { return mie_extinction(c, w) * 0.9f; }
static inline float rayleigh_coefficient(float w) {
    const float x = 550.0f / w, x2 = x * x;
    return (5.10e-31f * (x2 * x2)) * 2.546899e25f;
}
static inline float rayleigh_phase(float ct) {
    const float c = clampf(ct, -1.0f, 1.0f);
    return 3.0f * (1.0f + c * c) / (16.0f * PI_F);
}
static inline float mie_phase(float ct, float g_in) {
    const float c = clampf(ct, -1.0f, 1.0f), g = clampf(g_in, -0.999f, 0.999f);
    const float den = powf(fmaxf(1.0f + g * g - 2.0f * g * c, 1.0e-6f), 1.5f);
    return 3.0f * (1.0f - g * g) * (1.0f + c * c) / (8.0f * PI_F * (2.0f + g * g) * den);
}
static void xyz_to_rgb(const float *xyz, float *rgb) {
    for (int r = 0; r < 3; r++) rgb[r] = XYZ2RGB[r][0] * xyz[0] + XYZ2RGB[r][1] * xyz[1] + XYZ2RGB[r][2] * xyz[2];
}
static void integrate_xyz(const float *s, float *xyz) { /* :94-107 */
    xyz[0] = xyz[1] = xyz[2] = 0.0f;
    for (int i = 0; i < NW; i++) {
        const float weight = (i == 0 || i + 1 == NW) ? 0.5f : 1.0f;
        for (int k = 0; k < 3; k++) xyz[k] += s[i] * CIE[i][k] * weight;
    }
}
static void spectral_to_linear_rgb(const float *s, float *rgb) { /* :119-124 */
    float xyz[3], raw[3], ones[NW], white[3];
    integrate_xyz(s, xyz);
    xyz_to_rgb(xyz, raw);
    for (int i = 0; i < NW; i++) ones[i] = 1.0f;
    integrate_xyz(ones, xyz);
    xyz_to_rgb(xyz, white);
    for (int k = 0; k < 3; k++) rgb[k] = raw[k] / white[k];
}
static uint16_t f16_bits(float f) { /* half::f16::from_f32: round to nearest even */
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, mag = x & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (mag > 0x7F800000u ? 0x0200u : 0u));
    if (mag >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);
    if (mag <= 0x33000000u) return (uint16_t)sign;
    const int32_t e = (int32_t)(mag >> 23) - 127;
    const uint32_t m = (mag & 0x007FFFFFu) | 0x00800000u;
    const uint32_t drop = (e < -14) ? (uint32_t)(13 + (-14 - e)) : 13u;
    uint32_t q = m >> drop;
    const uint32_t rem = m & ((1u << drop) - 1u), half = 1u << (drop - 1u);
    if (rem > half || (rem == half && (q & 1u))) q++;
    if (e < -14) return (uint16_t)(sign | q);
    return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (q - 0x400u)));
}
static void rgba_from_spectral(const float *s, float alpha, uint16_t *out) { /* bake.rs:1466-1474 */
    float rgb[3];
    spectral_to_linear_rgb(s, rgb);
    for (int k = 0; k < 3; k++) out[k] = f16_bits(clampf(rgb[k], 0.0f, 65504.0f));
    out[3] = f16_bits(clampf(alpha, 0.0f, 65504.0f));
}
static float spec_mean(const float *s) {
    float t = 0.0f;
    for (int i = 0; i < NW; i++) t += s[i];
    return t / (float)NW;
}

/* nonlinear LUT coordinates, bake.rs:291-319 */
static inline float signumf(float x) { return x < 0.0f ? -1.0f : 1.0f; } /* f32::signum: +1 for +0, -1 for -0 (never -0 here) */
static inline float mu_from_unit(float u) { const float x = 2.0f * clampf(u, 0.0f, 1.0f) - 1.0f; return (signbit(x) ? -1.0f : 1.0f) * (fabsf(x) * fabsf(x)); }
static inline float mu_to_unit(float mu_in) { const float mu = clampf(mu_in, -1.0f, 1.0f); return ((signbit(mu) ? -1.0f : 1.0f) * sqrtf(fabsf(mu)) + 1.0f) * 0.5f; }
static inline float nu_from_unit(float u) { const float d = 1.0f - clampf(u, 0.0f, 1.0f); return 1.0f - 2.0f * d * d; }
static inline float nu_to_unit(float nu) { return 1.0f - sqrtf((1.0f - clampf(nu, -1.0f, 1.0f)) * 0.5f); }
static inline float height_to_unit(float h, float H) { return sqrtf(clampf(h, 0.0f, H) / H); }
static inline float height_from_unit(float u, float H) { const float c = clampf(u, 0.0f, 1.0f); return H * (c * c); }

/* medium, bake.rs:788-873 */
static void density_at(const abo_config *c, float h_in, float *rho) {
    const float h = fmaxf(h_in, 0.0f);
    rho[0] = expf(-h / c->rayleigh_scale_height_m);
    rho[1] = expf(-h / c->mie_scale_height_m);
    rho[2] = fmaxf(1.0f - fabsf((h - 25000.0f) / 15000.0f), 0.0f) * c->ozone_du / 300.0f;
}
static inline float ozone_absorption(float w) { const float t = (w - 600.0f) / 85.0f; return 1.2e-6f * expf(-0.5f * (t * t)); }
static float distance_to_top(const abo_config *c, float h, float mu) {
    const float r = c->bottom_radius_m + clampf(h, 0.0f, atmosphere_height(c));
    const float radial = r * mu;
    const float disc = radial * radial + (c->top_radius_m - r) * (c->top_radius_m + r);
    return fmaxf(-radial + sqrtf(fmaxf(disc, 0.0f)), 0.0f);
}
static int distance_to_ground(const abo_config *c, float h, float mu, float *out) {
    if (mu >= 0.0f) return 0;
    const float r = c->bottom_radius_m + clampf(h, 0.0f, atmosphere_height(c));
    const float radial = r * mu;
    const float d = radial * radial - (r - c->bottom_radius_m) * (r + c->bottom_radius_m);
    if (d < 0.0f) return 0;
    const float s = -radial - sqrtf(d);
    if (!(s >= 0.0f)) return 0;
    *out = s;
    return 1;
}
static float distance_to_boundary(const abo_config *c, float h, float mu) {
    float s;
    return distance_to_ground(c, h, mu, &s) ? s : distance_to_top(c, h, mu);
}
static float altitude_along(const abo_config *c, float h, float mu, float s) {
    const float r = c->bottom_radius_m + clampf(h, 0.0f, atmosphere_height(c));
    return sqrtf(fmaxf(r * r + s * s + 2.0f * r * mu * s, 0.0f)) - c->bottom_radius_m;
}
static void optical_columns(const abo_config *c, float h, float mu, float d, int steps, float *out) {
    out[0] = out[1] = out[2] = 0.0f;
    if (d <= 0.0f) return;
    const float ds = d / (float)steps;
    for (int i = 0; i < steps; i++) {
        float rho[3];
        density_at(c, altitude_along(c, h, mu, ((float)i + 0.5f) * ds), rho);
        for (int k = 0; k < 3; k++) out[k] += rho[k] * ds;
    }
}
static void transmittance_from_columns(const abo_config *c, const float *col, float *t) {
    for (int i = 0; i < NW; i++) {
        const float w = WL[i];
        t[i] = expf(-fmaxf(rayleigh_coefficient(w) * col[0] + mie_extinction(c, w) * col[1] + ozone_absorption(w) * col[2], 0.0f));
    }
}
static inline float extinction_at_density(const abo_config *c, const float *rho, float w) {
    return fmaxf(rayleigh_coefficient(w) * rho[0] + mie_extinction(c, w) * rho[1] + ozone_absorption(w) * rho[2], 0.0f);
}
static inline float attenuated_cell_length(float extinction, float ds) {
    return extinction <= 1.0e-12f ? ds : -expm1f(-extinction * ds) / extinction;
}
static void transmittance_segment(const abo_config *c, float h, float mu, float d, float *t) {
    float col[3];
    optical_columns(c, h, mu, d, 64, col);
    transmittance_from_columns(c, col, t);
}

/* ray_sample_vectors / _altitude_and_sun_cosine / _geometry, bake.rs:1071-1135 */
typedef struct { float altitude_m, mu_sun, outgoing[3], sun[3], up[3], tangent[3]; } geom;
static geom ray_sample_geometry(const abo_config *c, float h, float mu_view, float mu_sun, float nu, float distance) {
    geom g;
    const float mv = clampf(mu_view, -1.0f, 1.0f), ms = clampf(mu_sun, -1.0f, 1.0f);
    const float vx = sqrtf(fmaxf(1.0f - mv * mv, 0.0f)), sh = sqrtf(fmaxf(1.0f - ms * ms, 0.0f));
    const float requested = vx > 1.0e-6f ? (clampf(nu, -1.0f, 1.0f) - mv * ms) / vx : 0.0f;
    const float sx = clampf(requested, -sh, sh);
    const float sz = sqrtf(fmaxf(sh * sh - sx * sx, 0.0f));
    g.outgoing[0] = vx; g.outgoing[1] = mv; g.outgoing[2] = 0.0f;
    g.sun[0] = sx; g.sun[1] = ms; g.sun[2] = sz;
    const float r = c->bottom_radius_m + clampf(h, 0.0f, atmosphere_height(c));
    const float position[3] = {g.outgoing[0] * distance, r + g.outgoing[1] * distance, 0.0f};
    const float sr = fmaxf(sqrtf(dot3(position, position)), c->bottom_radius_m);
    g.up[0] = position[0] / sr; g.up[1] = position[1] / sr; g.up[2] = 0.0f;
    g.tangent[0] = g.up[1]; g.tangent[1] = -g.up[0]; g.tangent[2] = 0.0f;
    g.altitude_m = clampf(sr - c->bottom_radius_m, 0.0f, atmosphere_height(c));
    g.mu_sun = clampf(dot3(g.sun, g.up), -1.0f, 1.0f);
    return g;
}

/* integrate_single_scattering, bake.rs:875-915 (radiance only) */
static void integrate_single_scattering(const abo_config *c, float h, float mu_view, float mu_sun, float nu, float distance_limit,
                                        int steps, float *radiance) {
    for (int w = 0; w < NW; w++) radiance[w] = 0.0f;
    const float length = fminf(distance_to_boundary(c, h, mu_view), distance_limit);
    if (length <= 0.0f) return;
    const float ds = length / (float)steps;
    float view_columns[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 0; i < steps; i++) {
        const float s = ((float)i + 0.5f) * ds;
        const geom g = ray_sample_geometry(c, h, mu_view, mu_sun, nu, s);
        float rho[3], view_start[NW], unused;
        density_at(c, g.altitude_m, rho);
        transmittance_from_columns(c, view_columns, view_start);
        if (!distance_to_ground(c, g.altitude_m, g.mu_sun, &unused)) {
            float sun_columns[3], sun_t[NW];
            optical_columns(c, g.altitude_m, g.mu_sun, distance_to_top(c, g.altitude_m, g.mu_sun), 64, sun_columns);
            transmittance_from_columns(c, sun_columns, sun_t);
            for (int w = 0; w < NW; w++) {
                const float wl = WL[w];
                const float scatter = rayleigh_coefficient(wl) * rho[0] * rayleigh_phase(nu) + mie_scattering(c, wl) * rho[1] * mie_phase(nu, c->mie_g);
                const float cell = attenuated_cell_length(extinction_at_density(c, rho, wl), ds);
                radiance[w] += view_start[w] * sun_t[w] * scatter * cell;
            }
        }
        for (int k = 0; k < 3; k++) view_columns[k] += rho[k] * ds;
    }
}

/* angular quadrature, bake.rs:1137-1215 */
static const float GL[16][3] = {{-0.9894009f, 0.14520948f, 0.02715246f}, {-0.944575f, 0.32829565f, 0.062253524f}, {-0.8656312f, 0.5006822f, 0.09515851f},
                                {-0.7554044f, 0.6552589f, 0.12462897f}, {-0.61787623f, 0.7862754f, 0.14959599f}, {-0.45801678f, 0.88894355f, 0.16915652f},
                                {-0.28160354f, 0.95953083f, 0.18260342f}, {-0.09501251f, 0.99547607f, 0.1894506f}, {0.09501251f, 0.99547607f, 0.1894506f},
                                {0.28160354f, 0.95953083f, 0.18260342f}, {0.45801678f, 0.88894355f, 0.16915652f}, {0.61787623f, 0.7862754f, 0.14959599f},
                                {0.7554044f, 0.6552589f, 0.12462897f}, {0.8656312f, 0.5006822f, 0.09515851f}, {0.944575f, 0.32829565f, 0.062253524f},
                                {0.9894009f, 0.14520948f, 0.02715246f}};
static const float AZ_POS[8][2] = {{0.9951847f, 0.09801714f}, {0.95694035f, 0.29028466f}, {0.8819213f, 0.47139674f}, {0.77301043f, 0.6343933f},
                                   {0.6343933f, 0.77301043f}, {0.47139674f, 0.8819213f}, {0.29028466f, 0.95694035f}, {0.09801714f, 0.9951847f}};
#define NQ 512
static float QDIR[NQ][3], QW[NQ], Q_COS_NORM;
static void build_quadrature(void) {
    float az[32][2];
    for (int k = 0; k < 8; k++) {       /* the four quadrants of the reference's 32-entry table */
        az[k][0] = AZ_POS[k][0];           az[k][1] = AZ_POS[k][1];
        az[8 + k][0] = -AZ_POS[7 - k][0];  az[8 + k][1] = AZ_POS[7 - k][1];
        az[16 + k][0] = -AZ_POS[k][0];     az[16 + k][1] = -AZ_POS[k][1];
        az[24 + k][0] = AZ_POS[7 - k][0];  az[24 + k][1] = -AZ_POS[7 - k][1];
    }
    const float azimuth_weight = TAU_F / 32.0f;
    int q = 0;
    for (int i = 0; i < 16; i++)
        for (int a = 0; a < 32; a++, q++) {
            QDIR[q][0] = GL[i][1] * az[a][0];
            QDIR[q][1] = GL[i][0];
            QDIR[q][2] = GL[i][1] * az[a][1];
            QW[q] = GL[i][2] * azimuth_weight;
        }
    float sum = 0.0f;                   /* ground_boundary_source's cosine_weight_sum, :1312-1317 */
    for (q = 0; q < NQ; q++)
        if (QDIR[q][1] > 0.0f) sum += QDIR[q][1] * QW[q];
    Q_COS_NORM = PI_F / sum;
}

static inline size_t scattering_index(const abo_config *c, size_t h, size_t n, size_t s, size_t v) { /* :1218-1222 */
    return (((h * c->scattering_nu + n) * c->scattering_mu_sun + s) * c->scattering_mu_view) + v;
}
/* sample_spectral_scattering, :1225-1270 */
static void sample_scattering(const abo_config *c, const spec *values, float h, float mu_sun, float mu_view, float nu, float *out) {
    const float p[4] = {height_to_unit(h, atmosphere_height(c)) * (float)(c->scattering_height - 1u), nu_to_unit(nu) * (float)(c->scattering_nu - 1u),
                        mu_to_unit(mu_sun) * (float)(c->scattering_mu_sun - 1u), mu_to_unit(mu_view) * (float)(c->scattering_mu_view - 1u)};
    const size_t ext[4] = {c->scattering_height, c->scattering_nu, c->scattering_mu_sun, c->scattering_mu_view};
    size_t lo[4], hi[4];
    float f[4];
    for (int a = 0; a < 4; a++) {
        lo[a] = (size_t)floorf(p[a]);
        hi[a] = lo[a] + 1 < ext[a] - 1 ? lo[a] + 1 : ext[a] - 1;
        f[a] = p[a] - (float)lo[a];
    }
    for (int k = 0; k < NW; k++) out[k] = 0.0f;
    for (int hs = 0; hs < 2; hs++)
        for (int ns = 0; ns < 2; ns++)
            for (int ss = 0; ss < 2; ss++)
                for (int vs = 0; vs < 2; vs++) {
                    const int sides[4] = {hs, ns, ss, vs};
                    float w = 1.0f;
                    size_t i[4];
                    for (int a = 0; a < 4; a++) {
                        i[a] = sides[a] == 0 ? lo[a] : hi[a];
                        w *= sides[a] == 0 ? 1.0f - f[a] : f[a];
                    }
                    const spec *q = &values[scattering_index(c, i[0], i[1], i[2], i[3])];
                    for (int k = 0; k < NW; k++) out[k] += w * q->v[k];
                }
}
static inline void quadrature_direction(const geom *g, const float *local, float *out) { /* :1273-1279 */
    out[0] = g->tangent[0] * local[0] + g->up[0] * local[1];
    out[1] = g->tangent[1] * local[0] + g->up[1] * local[1];
    out[2] = local[2];
}
static void phase_normalization(const geom *g, float mie_g, float *n) { /* :1286-1298 */
    n[0] = n[1] = 0.0f;
    for (int q = 0; q < NQ; q++) {
        float incoming[3];
        quadrature_direction(g, QDIR[q], incoming);
        const float cosine = dot3(incoming, g->outgoing);
        n[0] += rayleigh_phase(cosine) * QW[q];
        n[1] += mie_phase(cosine, mie_g) * QW[q];
    }
    n[0] = fmaxf(n[0], 1.0e-8f);
    n[1] = fmaxf(n[1], 1.0e-8f);
}
/* ground_boundary_source, :1300-1350 */
static void ground_boundary_source(const abo_config *c, const spec *incident, const float *sun, int include_direct_sun, float *radiance) {
    for (int w = 0; w < NW; w++) radiance[w] = 0.0f;
    if (c->ground_albedo <= 0.0f) return;
    float irradiance[NW];
    for (int w = 0; w < NW; w++) irradiance[w] = 0.0f;
    if (incident) {
        for (int q = 0; q < NQ; q++) {
            if (QDIR[q][1] <= 0.0f) continue;
            float sample[NW];
            sample_scattering(c, incident, 0.0f, clampf(sun[1], -1.0f, 1.0f), QDIR[q][1], dot3(QDIR[q], sun), sample);
            for (int w = 0; w < NW; w++) irradiance[w] += sample[w] * QDIR[q][1] * QW[q] * Q_COS_NORM;
        }
    }
    for (int w = 0; w < NW; w++) radiance[w] = irradiance[w] * c->ground_albedo / PI_F;
    if (include_direct_sun && sun[1] > 0.0f) {
        const float mu_sun = clampf(sun[1], 0.0f, 1.0f);
        float t[NW];
        transmittance_segment(c, 0.0f, mu_sun, distance_to_top(c, 0.0f, mu_sun), t);
        for (int w = 0; w < NW; w++) radiance[w] += c->ground_albedo * mu_sun * t[w] / PI_F;
    }
}
/* ground_boundary_along_ray, :1354-1372 */
static void ground_boundary_along_ray(const abo_config *c, const spec *incident, float h, float mu_view, float mu_sun, float nu,
                                      int include_direct_sun, float *out) {
    float length;
    for (int w = 0; w < NW; w++) out[w] = 0.0f;
    if (!distance_to_ground(c, h, mu_view, &length)) return;
    const geom end = ray_sample_geometry(c, h, mu_view, mu_sun, nu, length);
    const float sun_local[3] = {dot3(end.sun, end.tangent), dot3(end.sun, end.up), end.sun[2]};
    float boundary[NW], t[NW];
    ground_boundary_source(c, incident, sun_local, include_direct_sun, boundary);
    transmittance_segment(c, h, mu_view, length, t);
    for (int w = 0; w < NW; w++) out[w] = t[w] * boundary[w];
}
#define ORDER_STEPS 16
static inline float cell_edge(float length, int index, int ground_bound) { /* :1387-1397 */
    const float unit = (float)index / (float)ORDER_STEPS;
    return ground_bound ? length * (1.0f - (1.0f - unit) * (1.0f - unit)) : length * unit * unit;
}
/* integrate_scattering_order, :1400-1463 */
static void integrate_scattering_order(const abo_config *c, const spec *previous, float h, float mu_view, float mu_sun, float nu,
                                       float *volume, float *transport) {
    const float length = distance_to_boundary(c, h, mu_view);
    for (int w = 0; w < NW; w++) volume[w] = 0.0f;
    if (length > 0.0f) {
        float unused;
        const int ground_bound = distance_to_ground(c, h, mu_view, &unused);
        float columns[3] = {0.0f, 0.0f, 0.0f};
        for (int i = 0; i < ORDER_STEPS; i++) {
            const float start = cell_edge(length, i, ground_bound), end = cell_edge(length, i + 1, ground_bound);
            const float ds = end - start, distance = 0.5f * (start + end);
            const geom g = ray_sample_geometry(c, h, mu_view, mu_sun, nu, distance);
            float rho[3], view_start[NW], norm[2], source[NW];
            density_at(c, g.altitude_m, rho);
            transmittance_from_columns(c, columns, view_start);
            phase_normalization(&g, c->mie_g, norm);
            for (int w = 0; w < NW; w++) source[w] = 0.0f;
            for (int q = 0; q < NQ; q++) {
                float incoming[3], l[NW];
                quadrature_direction(&g, QDIR[q], incoming);
                sample_scattering(c, previous, g.altitude_m, g.mu_sun, dot3(incoming, g.up), dot3(incoming, g.sun), l);
                const float cosine = dot3(incoming, g.outgoing);
                for (int w = 0; w < NW; w++) {
                    const float wl = WL[w];
                    const float scatter = rayleigh_coefficient(wl) * rho[0] * rayleigh_phase(cosine) / norm[0] +
                                          mie_scattering(c, wl) * rho[1] * mie_phase(cosine, c->mie_g) / norm[1];
                    source[w] += scatter * l[w] * QW[q];
                }
            }
            for (int w = 0; w < NW; w++) volume[w] += view_start[w] * source[w] * attenuated_cell_length(extinction_at_density(c, rho, WL[w]), ds);
            for (int k = 0; k < 3; k++) columns[k] += rho[k] * ds;
        }
    }
    float boundary[NW];
    ground_boundary_along_ray(c, previous, h, mu_view, mu_sun, nu, 0, boundary);
    for (int w = 0; w < NW; w++) transport[w] = volume[w] + boundary[w];
}

/* bake_atmosphere_luts, bake.rs:1481-1666.  Outputs: RGBA16F bit patterns (x fastest as in LutData), order deltas.
 * single / accumulated: [height][nu][mu_sun][mu_view]; transmittance [height][mu]; aerial [height][mu_view][distance]. */
int abo_bake(const abo_config *c, uint16_t *transmittance, uint16_t *single_rgba, uint16_t *accumulated_rgba, uint16_t *aerial, float *deltas) {
    if (!c || c->scattering_orders < 2u || c->scattering_orders > 8u) return 1;
    build_quadrature();
    const float H = atmosphere_height(c);
    for (uint32_t hi = 0; hi < c->transmittance_height; hi++) {
        const float h = H * (float)hi / (float)(c->transmittance_height - 1u);
        for (uint32_t mi = 0; mi < c->transmittance_mu; mi++) {
            const float mu = -1.0f + 2.0f * (float)mi / (float)(c->transmittance_mu - 1u);
            float s[NW];
            transmittance_segment(c, h, mu, distance_to_boundary(c, h, mu), s);
            rgba_from_spectral(s, spec_mean(s), transmittance + 4u * ((size_t)hi * c->transmittance_mu + mi));
        }
    }
    const size_t count = (size_t)c->scattering_mu_view * c->scattering_mu_sun * c->scattering_height * c->scattering_nu;
    spec *single = (spec *)malloc(count * sizeof(spec)), *previous = (spec *)malloc(count * sizeof(spec)),
         *accumulated = (spec *)malloc(count * sizeof(spec)), *next = (spec *)malloc(count * sizeof(spec));
    if (!single || !previous || !accumulated || !next) return 2;
    const int64_t nh = c->scattering_height, nn = c->scattering_nu, ns = c->scattering_mu_sun, nv = c->scattering_mu_view;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t e = 0; e < (int64_t)count; e++) {
        const int64_t vi = e % nv, si = (e / nv) % ns, ni = (e / (nv * ns)) % nn, hi = e / (nv * ns * nn);
        const float h = height_from_unit((float)hi / (float)(nh - 1), H), nu = nu_from_unit((float)ni / (float)(nn - 1));
        const float ms = mu_from_unit((float)si / (float)(ns - 1)), mv = mu_from_unit((float)vi / (float)(nv - 1));
        float volume[NW], direct_ground[NW];
        integrate_single_scattering(c, h, mv, ms, nu, 3.402823466e38f, 64, volume);
        ground_boundary_along_ray(c, NULL, h, mv, ms, nu, 1, direct_ground);
        for (int w = 0; w < NW; w++) {
            single[e].v[w] = volume[w];
            previous[e].v[w] = volume[w] + direct_ground[w];
            accumulated[e].v[w] = volume[w];
        }
    }
    float first = 0.0f;
    for (size_t e = 0; e < count; e++)
        for (int w = 0; w < NW; w++) first += fabsf(previous[e].v[w]);
    deltas[0] = first / (float)(count * NW);
    for (uint32_t order = 2; order <= c->scattering_orders; order++) {
#pragma omp parallel for schedule(dynamic, 8)
        for (int64_t e = 0; e < (int64_t)count; e++) {
            const int64_t vi = e % nv, si = (e / nv) % ns, ni = (e / (nv * ns)) % nn, hi = e / (nv * ns * nn);
            const float h = height_from_unit((float)hi / (float)(nh - 1), H), nu = nu_from_unit((float)ni / (float)(nn - 1));
            const float ms = mu_from_unit((float)si / (float)(ns - 1)), mv = mu_from_unit((float)vi / (float)(nv - 1));
            float volume[NW], transport[NW];
            integrate_scattering_order(c, previous, h, mv, ms, nu, volume, transport);
            for (int w = 0; w < NW; w++) {
                next[e].v[w] = transport[w];
                accumulated[e].v[w] += volume[w];
            }
        }
        float delta = 0.0f;
        for (size_t e = 0; e < count; e++)
            for (int w = 0; w < NW; w++) delta += fabsf(next[e].v[w]);
        deltas[order - 1u] = delta / (float)(count * NW);
        spec *t = previous;
        previous = next;
        next = t;
    }
    for (size_t e = 0; e < count; e++) {
        rgba_from_spectral(single[e].v, spec_mean(single[e].v), single_rgba + 4u * e);
        rgba_from_spectral(accumulated[e].v, spec_mean(accumulated[e].v), accumulated_rgba + 4u * e);
    }
    for (uint32_t hi = 0; hi < c->aerial_height; hi++) {
        const float h = H * (float)hi / (float)(c->aerial_height - 1u);
        for (uint32_t vi = 0; vi < c->aerial_mu_view; vi++) {
            const float mu = -1.0f + 2.0f * (float)vi / (float)(c->aerial_mu_view - 1u);
            for (uint32_t di = 0; di < c->aerial_distance; di++) {
                const float distance = c->max_aerial_distance_m * (float)di / (float)(c->aerial_distance - 1u);
                float t[NW];
                transmittance_segment(c, h, mu, fminf(distance_to_boundary(c, h, mu), distance), t);
                uint16_t *o = aerial + 4u * (((size_t)hi * c->aerial_mu_view + vi) * c->aerial_distance + di);
                o[0] = o[1] = o[2] = 0u;
                o[3] = f16_bits(spec_mean(t));
            }
        }
    }
    free(single); free(previous); free(accumulated); free(next);
    return 0;
}
