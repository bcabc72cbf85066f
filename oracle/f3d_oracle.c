/* oracle/f3d_oracle.c
 *
 * TEST INFRASTRUCTURE ONLY -- see f3d_oracle.h.  A plain-C, CPU restatement of the
 * reference's PROMETHEUS terrain path tracer.  Every function cites the reference
 * file:line it follows (paths relative to the reference checkout).  The structure is
 * deliberately literal (one function per WGSL function, same control flow, same
 * 64-entry stack, same 80-byte reservoirs, three separate passes per frame) so it can
 * be audited against the WGSL line by line; it is NOT how the HIP product is built.
 *
 * Floating-point conventions (the WGSL spec leaves these to the driver; the reference
 * accepts any conforming choice because its gate is SSIM/mean-abs):
 *   - IEEE f32, round-to-nearest-even, denormals kept, compiled with -ffp-contract=off.
 *   - FMA is used ONLY where this file spells fmaf(): dot products, mix(), the
 *     "origin + t * direction" family and polynomial evaluation.  The HIP kernels
 *     spell the same fmaf() at the same places, which makes the two bit-comparable.
 *   - normalize(v) = v * (1/sqrt(dot(v,v))) (glam's and most drivers' lowering).
 *   - x / spacing is evaluated as x * (1/spacing) with the reciprocal rounded once.
 *   - sin/cos/atan2/acos are evaluated by the fixed polynomials below (error < 2 ulp,
 *     far inside WGSL's accuracy allowance) instead of libm so that every platform
 *     gets the same bits; tan(fov/2) is a per-render constant taken from libm tanf.
 */
#include "f3d_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* small vector helpers                                                       */
/* ------------------------------------------------------------------------- */
typedef struct {
    float x, y, z;
} v3;

static inline v3 v3_make(float x, float y, float z) {
    v3 r = {x, y, z};
    return r;
}
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline float dot2(float ax, float az, float bx, float bz) { return fmaf(az, bz, ax * bx); }
static inline v3 cross3(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 normalize3(v3 a) {
    float inv = 1.0f / sqrtf(dot3(a, a));
    return v3_scale(a, inv);
}
/* o + t*d, component-wise fma */
static inline v3 v3_madd(v3 o, float t, v3 d) {
    return v3_make(fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), fmaf(t, d.z, o.z));
}
/* WGSL mix(a,b,t) = a*(1-t) + b*t */
static inline float mixf(float a, float b, float t) { return fmaf(b, t, a * (1.0f - t)); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
/* WGSL u32(f): saturating */
static inline uint32_t sat_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

/* ------------------------------------------------------------------------- */
/* deterministic transcendental kernels (cephes single-precision coefficients) */
/* ------------------------------------------------------------------------- */
static const float F3D_PI = 3.14159265358979323846f;
static const float F3D_HALF_PI = 1.57079632679489661923f;

static inline float poly_sin(float x) { /* |x| <= pi/4 */
    float z = x * x;
    float p = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    return fmaf(p * z, x, x);
}
static inline float poly_cos(float x) { /* |x| <= pi/4 */
    float z = x * x;
    float p = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                   4.166664568298827e-2f);
    return fmaf(p, z * z, fmaf(-0.5f, z, 1.0f));
}
/* sin and cos of phi = 2*pi*u, u in [0,1]: quadrant from 4u, remainder * pi/2. */
#ifdef F3DO_EXPERIMENT
extern int f3do_experiment_libm; /* 1: libm sinf / cosf of 2*pi*u instead of the fixed polynomials */
#endif
void f3do_sincos_2pi(float u, float *s_out, float *c_out) {
#ifdef F3DO_EXPERIMENT
    if (f3do_experiment_libm) { float phi = 6.283185307179586f * u; *s_out = sinf(phi); *c_out = cosf(phi); return; }
#endif
    float a = 4.0f * u;
    float k = rintf(a);
    float r = a - k;
    float x = r * F3D_HALF_PI;
    float s = poly_sin(x), c = poly_cos(x);
    int q = ((int)k) & 3;
    float so, co;
    if (q == 0) { so = s; co = c; }
    else if (q == 1) { so = c; co = -s; }
    else if (q == 2) { so = -s; co = -c; }
    else { so = -c; co = s; }
    *s_out = so;
    *c_out = co;
}
static inline float det_atanf(float xx) {
    float sign = 1.0f, x = xx;
    if (xx < 0.0f) { sign = -1.0f; x = -xx; }
    float y;
    if (x > 2.414213562373095f) { y = F3D_HALF_PI; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
    else { y = 0.0f; }
    float z = x * x;
    float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z,
                   -3.33329491539e-1f);
    y = y + fmaf(p * z, x, x);
    return sign * y;
}
static inline float det_atan2f(float y, float x) {
    if (x > 0.0f) return det_atanf(y / x);
    if (x < 0.0f) {
        float a = det_atanf(y / x);
        return (y >= 0.0f) ? a + F3D_PI : a - F3D_PI;
    }
    if (y > 0.0f) return F3D_HALF_PI;
    if (y < 0.0f) return -F3D_HALF_PI;
    return 0.0f;
}
static inline float det_asinf_core(float x) { /* |x| <= 0.5 */
    float z = x * x;
    float p = fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z,
                        7.4953002686e-2f), z, 1.6666752422e-1f);
    return fmaf(x * z, p, x);
}
static inline float det_acosf(float x) { /* x in [-1,1] */
    if (x < -0.5f) return F3D_PI - 2.0f * det_asinf_core(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * det_asinf_core(sqrtf(0.5f * (1.0f - x)));
    return F3D_HALF_PI - det_asinf_core(x);
}

/* ------------------------------------------------------------------------- */
/* f16 (IEEE binary16) round trip: RGBA16F storage of out_tex and the AOVs     */
/* (render_terrain.rs:1358-1366, :438-447)                                     */
/* ------------------------------------------------------------------------- */
static inline uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7FFFFFFFu;
    if (absx >= 0x7F800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | ((absx > 0x7F800000u) ? 0x0200u : 0u));
    }
    if (absx >= 0x477FF000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (absx < 0x33000001u) { /* < 2^-25 (or exactly) -> 0 (ties-to-even at 2^-25 -> 0) */
        return (uint16_t)sign;
    }
    int32_t exp = (int32_t)(absx >> 23) - 127;
    uint32_t mant = (absx & 0x007FFFFFu) | 0x00800000u;
    uint32_t half;
    if (exp < -14) { /* subnormal half */
        int shift = (-14 - exp) + 13; /* bits to drop from the 24-bit mantissa */
        uint32_t q = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (q & 1u))) q++;
        half = q;
    } else {
        uint32_t q = ((uint32_t)(exp + 15) << 10) | ((mant >> 13) & 0x3FFu);
        uint32_t rem = mant & 0x1FFFu;
        if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;
        half = q;
    }
    return (uint16_t)(sign | half);
}
static inline float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t mant = h & 0x3FFu;
    uint32_t x;
    if (exp == 0) {
        if (mant == 0) { x = sign; }
        else {
            int e = -1;
            do { e++; mant <<= 1; } while ((mant & 0x400u) == 0);
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        x = sign | 0x7F800000u | (mant << 13);
    } else {
        x = sign | ((exp + 112u) << 23) | (mant << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}
float f3do_f16_round(float v) { return f16_bits_to_f32(f32_to_f16_bits(v)); }

/* ------------------------------------------------------------------------- */
/* geo::refraction (src/geo/refraction.rs)                                    */
/* ------------------------------------------------------------------------- */
static const double WGS84_A_M = 6378137.0;
static const double WGS84_E2 = 6.6943799901413165e-3;
static const double DEG2RAD = 0.017453292519943295;

int f3do_effective_radius_m(int32_t earth_model, double latitude_deg, double sphere_radius_m,
                            int32_t refraction_model, double pressure_mbar,
                            double temperature_c, double k_in, double azimuth_deg,
                            double *radius_out, char *err, size_t errlen) {
    /* effective_radius_m, refraction.rs:137-148 */
    if (earth_model == 0 && refraction_model != 0) {
        snprintf(err, errlen, "flat earth only supports refraction_model='none'");
        return 1;
    }
    /* directional_radius_m, refraction.rs:57-77 */
    if (!isfinite(azimuth_deg)) {
        snprintf(err, errlen, "azimuth must be finite");
        return 1;
    }
    double radius;
    if (earth_model == 0) {
        radius = INFINITY;
    } else if (earth_model == 1) {
        if (!(isfinite(sphere_radius_m) && sphere_radius_m > 0.0)) {
            snprintf(err, errlen, "sphere radius must be finite and positive");
            return 1;
        }
        radius = sphere_radius_m;
    } else if (earth_model == 2) {
        if (!(isfinite(latitude_deg) && latitude_deg >= -90.0 && latitude_deg <= 90.0)) {
            snprintf(err, errlen, "latitude must be finite and in [-90, 90]");
            return 1;
        }
        /* principal_radii_m, refraction.rs:6-13 */
        double phi = latitude_deg * DEG2RAD;
        double sp = sin(phi);
        double w = sqrt(1.0 - WGS84_E2 * (sp * sp));
        double meridional = WGS84_A_M * (1.0 - WGS84_E2) / (w * w * w);
        double prime_vertical = WGS84_A_M / w;
        double az = azimuth_deg * DEG2RAD;
        double ca = cos(az), sa = sin(az);
        radius = 1.0 / ((ca * ca) / meridional + (sa * sa) / prime_vertical);
    } else {
        snprintf(err, errlen, "unsupported earth_model");
        return 1;
    }
    /* RefractionModel::k, refraction.rs:100-122 and standard_k :160-165 */
    double k;
    if (refraction_model == 0) k = 0.0;
    else if (refraction_model == 3) k = k_in;
    else if (refraction_model == 1 || refraction_model == 2) {
        double base = (refraction_model == 1) ? 0.13 : (1.0 / 7.0);
        if (!isfinite(pressure_mbar) || pressure_mbar <= 0.0 || temperature_c <= -273.15) {
            snprintf(err, errlen, "pressure must be positive and temperature above absolute zero");
            return 1;
        }
        k = base * (pressure_mbar / 1013.25) * (288.15 / (273.15 + temperature_c));
    } else {
        snprintf(err, errlen, "unsupported refraction_model");
        return 1;
    }
    if (!(isfinite(k) && k < 1.0)) {
        snprintf(err, errlen, "refraction k must be finite and less than 1");
        return 1;
    }
    *radius_out = radius / (1.0 - k);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* build_minmax_mips (terrain_heightfield.rs:132-202)                          */
/* ------------------------------------------------------------------------- */
static uint32_t next_pow2_u32(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

#define F3DO_MAX_LEVELS 16

typedef struct {
    float *levels[F3DO_MAX_LEVELS]; /* (ph, pw, 2) */
    uint32_t pw[F3DO_MAX_LEVELS], ph[F3DO_MAX_LEVELS];
    uint32_t count;
    uint32_t cell_w, cell_h;
    uint64_t bytes; /* DEM texture + all mip levels, as TerrainMinMaxPyramid::byte_size */
} mips_t;

static void mips_free(mips_t *m) {
    for (uint32_t i = 0; i < m->count; i++) free(m->levels[i]);
    m->count = 0;
}

static int mips_build(const float *heights, uint32_t w, uint32_t h, mips_t *m) {
    memset(m, 0, sizeof(*m));
    uint32_t cw = w - 1, ch = h - 1;
    uint32_t pw = next_pow2_u32(cw), ph = next_pow2_u32(ch);
    float *l0 = (float *)malloc((size_t)pw * ph * 2 * sizeof(float));
    if (!l0) return -1;
    for (size_t i = 0; i < (size_t)pw * ph; i++) { l0[2 * i] = INFINITY; l0[2 * i + 1] = -INFINITY; }
    for (uint32_t y = 0; y < ch; y++) {
        for (uint32_t x = 0; x < cw; x++) {
            size_t i00 = (size_t)y * w + x;
            float a = heights[i00], b = heights[i00 + 1], c = heights[i00 + w], d = heights[i00 + w + 1];
            l0[2 * ((size_t)y * pw + x)] = fminf(fminf(fminf(a, b), c), d);
            l0[2 * ((size_t)y * pw + x) + 1] = fmaxf(fmaxf(fmaxf(a, b), c), d);
        }
    }
    m->levels[0] = l0; m->pw[0] = pw; m->ph[0] = ph; m->count = 1;
    m->cell_w = cw; m->cell_h = ch;
    m->bytes = (uint64_t)w * h * 4 + (uint64_t)pw * ph * 8;
    while (m->pw[m->count - 1] > 1 || m->ph[m->count - 1] > 1) {
        if (m->count >= F3DO_MAX_LEVELS) return -2;
        uint32_t lw = m->pw[m->count - 1], lh = m->ph[m->count - 1];
        uint32_t nw = lw / 2 > 1 ? lw / 2 : 1, nh = lh / 2 > 1 ? lh / 2 : 1;
        const float *prev = m->levels[m->count - 1];
        float *next = (float *)malloc((size_t)nw * nh * 2 * sizeof(float));
        if (!next) return -1;
        for (uint32_t y = 0; y < nh; y++) {
            for (uint32_t x = 0; x < nw; x++) {
                float mn = INFINITY, mx = -INFINITY;
                for (uint32_t dy = 0; dy < 2; dy++) {
                    for (uint32_t dx = 0; dx < 2; dx++) {
                        uint32_t sx = min_u32(2 * x + dx, lw - 1);
                        uint32_t sy = min_u32(2 * y + dy, lh - 1);
                        const float *v = &prev[2 * ((size_t)sy * lw + sx)];
                        mn = fminf(mn, v[0]);
                        mx = fmaxf(mx, v[1]);
                    }
                }
                next[2 * ((size_t)y * nw + x)] = mn;
                next[2 * ((size_t)y * nw + x) + 1] = mx;
            }
        }
        m->levels[m->count] = next; m->pw[m->count] = nw; m->ph[m->count] = nh;
        m->bytes += (uint64_t)nw * nh * 8;
        m->count++;
    }
    return 0;
}

int f3do_build_minmax_mips(const float *heights, uint32_t w, uint32_t h, float *levels_out,
                           uint32_t *dims_out, uint32_t max_levels, uint64_t *total_floats) {
    if (w < 2 || h < 2) return -10; /* "at least 2x2 texels" */
    for (size_t i = 0; i < (size_t)w * h; i++)
        if (!isfinite(heights[i])) return -11; /* "non-finite samples" */
    mips_t m;
    int rc = mips_build(heights, w, h, &m);
    if (rc != 0) { mips_free(&m); return rc; }
    uint64_t tot = 0;
    for (uint32_t l = 0; l < m.count; l++) {
        size_t n = (size_t)m.pw[l] * m.ph[l] * 2;
        if (levels_out) memcpy(levels_out + tot, m.levels[l], n * sizeof(float));
        if (dims_out && l < max_levels) { dims_out[2 * l] = m.pw[l]; dims_out[2 * l + 1] = m.ph[l]; }
        tot += n;
    }
    if (total_floats) *total_floats = tot;
    int count = (int)m.count;
    mips_free(&m);
    return count;
}

/* ------------------------------------------------------------------------- */
/* uniforms                                                                   */
/* ------------------------------------------------------------------------- */
typedef struct { /* WGSL Ray, hybrid_traversal.wgsl:29-34 */
    v3 origin;
    float tmin;
    v3 direction;
    float tmax;
} ray_t;

typedef struct { /* WGSL HybridHitResult, hybrid_traversal.wgsl:19-27 */
    float t;
    v3 point;
    v3 normal;
    uint32_t material_id, hit_type, hit;
} hit_t;

typedef struct {
    /* TerrainPtUniforms (terrain_heightfield.rs:29-38, :348-369) */
    float origin_x, origin_z, spacing_x, spacing_z;
    float inv_spacing_x, inv_spacing_z;
    float exaggeration, env_intensity;
    v3 albedo;
    uint32_t dem_w, dem_h, cell_w, cell_h;
    uint32_t mip_count, enabled, env_w, env_h;
    uint32_t spp, welford_window;
    /* EarthCurvatureUniforms (terrain_heightfield.rs:42-84) */
    float inv_two_r_prime;
    uint32_t curvature_enabled;
    /* data */
    const float *heights;
    const mips_t *mips;
    const float *env; /* rgb triples */
    /* HybridUniforms (hybrid_traversal.wgsl:9-17) */
    uint32_t traversal_mode;
    const float *mesh_vertices;
    uint32_t mesh_vertex_count;
    const uint32_t *mesh_indices;
    uint32_t mesh_index_count;
} scene_t;

typedef struct { /* Uniforms + LightingUniforms, hybrid_kernel.wgsl:8-38 */
    uint32_t width, height, frame_index, aov_flags;
    v3 cam_origin, cam_right, cam_up, cam_forward;
    float half_h, half_w; /* tan(0.5*fov_y), aspect*half_h */
    float cam_exposure;
    uint32_t seed_hi, seed_lo;
    v3 light_dir, light_color;
    uint32_t shadows_enabled;
} uniforms_t;

typedef struct {
    uint64_t n_node, n_leaf, n_hit, n_rays;
} counters_t;

/* ------------------------------------------------------------------------- */
/* hybrid_terrain_traversal.wgsl                                               */
/* ------------------------------------------------------------------------- */
#define TERRAIN_STACK_SIZE 64u
#define TERRAIN_RESTIR_M_CAP 512u

/* terrain_safe_inv, :88-91 */
static inline float terrain_safe_inv(float d) {
    float ad = fmaxf(fabsf(d), 1e-12f);
    return d < 0.0f ? -1.0f / ad : 1.0f / ad;
}

/* terrain_curved_height, :95-103.  c2 = dot(d.xz,d.xz) * inv_two_r_prime is the
 * per-ray constant of the curvature parabola (0 when the policy is off). */
static inline float curved_height(const ray_t *ray, float t, float c2) {
    return fmaf(t * t, c2, fmaf(t, ray->direction.y, ray->origin.y));
}

/* terrain_curved_height_range, :108-127 */
static inline void curved_height_range(const ray_t *ray, float t0, float t1, float c2, int curved,
                                       float *rmin, float *rmax) {
    float y0 = curved_height(ray, t0, c2);
    float y1 = curved_height(ray, t1, c2);
    float minimum = fminf(y0, y1);
    if (curved) {
        float a = c2;
        if (a > 0.0f) {
            float vertex = -ray->direction.y / (2.0f * a);
            if (vertex >= t0 && vertex <= t1) minimum = fminf(minimum, curved_height(ray, vertex, c2));
        }
    }
    *rmin = minimum;
    *rmax = fmaxf(y0, y1);
}

/* terrain_slab_xz, :131-141 (inv_x / inv_z hoisted: they only depend on the ray) */
static inline void slab_xz(const ray_t *ray, float inv_x, float inv_z, float x0, float x1, float z0,
                           float z1, float *t_enter, float *t_exit) {
    float tx0 = (x0 - ray->origin.x) * inv_x;
    float tx1 = (x1 - ray->origin.x) * inv_x;
    if (tx0 > tx1) { float tmp = tx0; tx0 = tx1; tx1 = tmp; }
    float tz0 = (z0 - ray->origin.z) * inv_z;
    float tz1 = (z1 - ray->origin.z) * inv_z;
    if (tz0 > tz1) { float tmp = tz0; tz0 = tz1; tz1 = tmp; }
    *t_enter = fmaxf(tx0, tz0);
    *t_exit = fminf(tx1, tz1);
}

/* terrain_pack_node, :144-146 */
static inline uint32_t pack_node(uint32_t level, uint32_t x, uint32_t y) {
    return (level << 26) | (y << 13) | x;
}

/* terrain_cell_heights, :149-156 */
static inline void cell_heights(const scene_t *sc, uint32_t cx, uint32_t cz, float h[4]) {
    float ex = sc->exaggeration;
    size_t i00 = (size_t)cz * sc->dem_w + cx;
    h[0] = sc->heights[i00] * ex;
    h[1] = sc->heights[i00 + 1] * ex;
    h[2] = sc->heights[i00 + sc->dem_w] * ex;
    h[3] = sc->heights[i00 + sc->dem_w + 1] * ex;
}

/* plane coordinate: origin + f32(cell) * spacing */
static inline float plane(float origin, uint32_t c, float spacing) {
    return fmaf((float)c, spacing, origin);
}

static float span_root(float d0, float dm, float d1, int any_hit);

/* terrain_leaf_intersect, :167-235 */
static inline int leaf_intersect(const scene_t *sc, const ray_t *ray, uint32_t cx, uint32_t cz,
                                 float t0, float t1, float c2, int any_hit, float *t_out) {
    float h[4];
    cell_heights(sc, cx, cz, h);
    float tm = 0.5f * (t0 + t1);
    float d3[3];
    for (int i = 0; i < 3; i++) {
        float t = (i == 0) ? t0 : ((i == 1) ? tm : t1);
        float px = fmaf(t, ray->direction.x, ray->origin.x);
        float pz = fmaf(t, ray->direction.z, ray->origin.z);
        float u = clampf(fmaf(px - sc->origin_x, sc->inv_spacing_x, -(float)cx), 0.0f, 1.0f);
        float v = clampf(fmaf(pz - sc->origin_z, sc->inv_spacing_z, -(float)cz), 0.0f, 1.0f);
        float hh = mixf(mixf(h[0], h[1], u), mixf(h[2], h[3], u), v);
        d3[i] = curved_height(ray, t, c2) - hh;
    }
    float s_hit = span_root(d3[0], d3[1], d3[2], any_hit);
    if (s_hit <= 1.0f) {
        float t = fmaf(s_hit, t1 - t0, t0);
        if (t > ray->tmin && t < ray->tmax) {
            *t_out = t;
            return 1;
        }
    }
    return 0;
}

/* The quadratic part of terrain_leaf_intersect (:197-226): smallest root s in [0,1] of the
 * parabola through d(0)=d0, d(1/2)=dm, d(1)=d1, or 1e30 when there is none. */
static float span_root(float d0, float dm, float d1, int any_hit) {
    float c = d0;
    float a = 2.0f * d1 + 2.0f * d0 - 4.0f * dm;
    float b = d1 - d0 - a;
    float s_hit = 1e30f;
    if (any_hit && c <= 0.0f) {
        s_hit = 0.0f;
    } else if (fabsf(a) < 1e-12f) {
        if (fabsf(b) > 1e-12f) {
            float s = -c / b;
            if (s >= 0.0f && s <= 1.0f) s_hit = s;
        }
    } else {
        float four_ac = 4.0f * a * c;
        float disc = fmaf(b, b, -four_ac);
        if (disc >= 0.0f) {
            float sq = sqrtf(disc);
            float q = -0.5f * (b + (b >= 0.0f ? sq : -sq));
            float r0 = q / a;
            float r1 = (fabsf(q) < 1e-30f) ? 1e30f : c / q;
            if (r0 > r1) { float tmp = r0; r0 = r1; r1 = tmp; }
            if (r0 >= 0.0f && r0 <= 1.0f) s_hit = r0;
            else if (r1 >= 0.0f && r1 <= 1.0f) s_hit = r1;
        }
    }
    return s_hit;
}
int f3do_deviation_span_hit(float d0, float dm, float d1, int32_t any_hit) {
    return span_root(d0, dm, d1, any_hit != 0) <= 1.0f;
}

/* terrain_normal_at, :239-248 */
static inline v3 terrain_normal_at(const scene_t *sc, v3 p, uint32_t cx, uint32_t cz) {
    float h[4];
    cell_heights(sc, cx, cz, h);
    float u = clampf(fmaf(p.x - sc->origin_x, sc->inv_spacing_x, -(float)cx), 0.0f, 1.0f);
    float v = clampf(fmaf(p.z - sc->origin_z, sc->inv_spacing_z, -(float)cz), 0.0f, 1.0f);
    float dh_du = mixf(h[1] - h[0], h[3] - h[2], v);
    float dh_dv = mixf(h[2] - h[0], h[3] - h[1], u);
    return normalize3(v3_make(-dh_du * sc->inv_spacing_x, 1.0f, -dh_dv * sc->inv_spacing_z));
}

/* terrain_trace, :254-372 */
static hit_t terrain_trace(const scene_t *sc, const ray_t *ray, int any_hit, int apply_curvature,
                           counters_t *cnt) {
    hit_t res;
    memset(&res, 0, sizeof(res));
    res.hit = 0u;
    res.t = ray->tmax;
    res.hit_type = 3u;
    if (!sc->enabled) return res;
    cnt->n_rays++;

    const uint32_t cell_w = sc->cell_w, cell_h = sc->cell_h;
    const float ox = sc->origin_x, oz = sc->origin_z, sx = sc->spacing_x, sz = sc->spacing_z;
    const float inv_x = terrain_safe_inv(ray->direction.x);
    const float inv_z = terrain_safe_inv(ray->direction.z);
    const int curved = apply_curvature && sc->curvature_enabled != 0u;
    const float hd2 = dot2(ray->direction.x, ray->direction.z, ray->direction.x, ray->direction.z);
    const float c2 = curved ? hd2 * sc->inv_two_r_prime : 0.0f;

    uint32_t stack[TERRAIN_STACK_SIZE];
    uint32_t sp = 0u;
    stack[sp++] = pack_node(sc->mip_count - 1u, 0u, 0u);

    for (;;) {
        if (sp == 0u) break;
        sp--;
        uint32_t node = stack[sp];
        uint32_t level = node >> 26, ny = (node >> 13) & 0x1FFFu, nx = node & 0x1FFFu;

        uint32_t cx0 = nx << level, cz0 = ny << level;
        if (cx0 >= cell_w || cz0 >= cell_h) continue;
        uint32_t cx1 = min_u32((nx + 1u) << level, cell_w);
        uint32_t cz1 = min_u32((ny + 1u) << level, cell_h);

        float s_en, s_ex;
        slab_xz(ray, inv_x, inv_z, plane(ox, cx0, sx), plane(ox, cx1, sx), plane(oz, cz0, sz),
                plane(oz, cz1, sz), &s_en, &s_ex);
        float t_lo = fmaxf(s_en, ray->tmin);
        float t_hi = fminf(s_ex, fminf(ray->tmax, res.t));
        if (t_lo > t_hi) continue;

        const float *mmp = &sc->mips->levels[level][2 * ((size_t)ny * sc->mips->pw[level] + nx)];
        cnt->n_node++;
        float mm_min = mmp[0] * sc->exaggeration, mm_max = mmp[1] * sc->exaggeration;
        float rh_min, rh_max;
        curved_height_range(ray, t_lo, t_hi, c2, curved, &rh_min, &rh_max);
        if (rh_min > mm_max || rh_max < mm_min) continue;

        if (level == 0u) {
            float lt;
            cnt->n_leaf++;
            if (leaf_intersect(sc, ray, cx0, cz0, t_lo, t_hi, c2, any_hit, &lt) && lt < res.t) {
                cnt->n_hit++;
                res.hit = 1u;
                res.t = lt;
                res.point = v3_madd(ray->origin, lt, ray->direction);
                res.normal = terrain_normal_at(sc, res.point, cx0, cz0);
                res.material_id = 0u;
                res.hit_type = 3u;
                if (any_hit) return res;
            }
            continue;
        }

        uint32_t child_level = level - 1u;
        float child_t[4];
        uint32_t child_id[4];
        uint32_t child_count = 0u;
        for (uint32_t cy = 0u; cy < 2u; cy++) {
            for (uint32_t cxi = 0u; cxi < 2u; cxi++) {
                uint32_t ccx = nx * 2u + cxi, ccy = ny * 2u + cy;
                uint32_t gx0 = ccx << child_level, gz0 = ccy << child_level;
                if (gx0 >= cell_w || gz0 >= cell_h) continue;
                uint32_t gx1 = min_u32((ccx + 1u) << child_level, cell_w);
                uint32_t gz1 = min_u32((ccy + 1u) << child_level, cell_h);
                float c_en, c_ex;
                slab_xz(ray, inv_x, inv_z, plane(ox, gx0, sx), plane(ox, gx1, sx),
                        plane(oz, gz0, sz), plane(oz, gz1, sz), &c_en, &c_ex);
                float ct_lo = fmaxf(c_en, t_lo);
                float ct_hi = fminf(c_ex, t_hi);
                if (ct_lo > ct_hi) continue;
                child_t[child_count] = ct_lo;
                child_id[child_count] = pack_node(child_level, ccx, ccy);
                child_count++;
            }
        }
        for (uint32_t i = 1u; i < child_count; i++) {
            float kt = child_t[i];
            uint32_t kid = child_id[i];
            uint32_t j = i;
            for (;;) {
                if (j == 0u || child_t[j - 1u] >= kt) break;
                child_t[j] = child_t[j - 1u];
                child_id[j] = child_id[j - 1u];
                j--;
            }
            child_t[j] = kt;
            child_id[j] = kid;
        }
        for (uint32_t i = 0u; i < child_count; i++) {
            if (sp < TERRAIN_STACK_SIZE) stack[sp++] = child_id[i];
        }
    }
    return res;
}

/* ------------------------------------------------------------------------- */
/* hybrid_traversal.wgsl                                                      */
/* ------------------------------------------------------------------------- */
/* ray_triangle_intersect, :86-132 */
static inline hit_t ray_triangle_intersect(const ray_t *ray, v3 v0, v3 v1, v3 v2) {
    hit_t result;
    memset(&result, 0, sizeof(result));
    result.t = ray->tmax;
    v3 edge1 = v3_sub(v1, v0), edge2 = v3_sub(v2, v0);
    v3 h = cross3(ray->direction, edge2);
    float a = dot3(edge1, h);
    if (fabsf(a) < 1e-7f) return result;
    float f = 1.0f / a;
    v3 s = v3_sub(ray->origin, v0);
    float u = f * dot3(s, h);
    if (u < 0.0f || u > 1.0f) return result;
    v3 q = cross3(s, edge1);
    float v = f * dot3(ray->direction, q);
    if (v < 0.0f || u + v > 1.0f) return result;
    float t = f * dot3(edge2, q);
    if (t > ray->tmin && t < ray->tmax) {
        result.hit = 1u;
        result.t = t;
        result.point = v3_madd(ray->origin, t, ray->direction);
        result.normal = normalize3(cross3(edge1, edge2));
        result.material_id = 0u;
        result.hit_type = 0u;
    }
    return result;
}

/* intersect_mesh, :137-172 (brute-force sweep; the BVH buffer is bound but unread) */
static hit_t intersect_mesh(const scene_t *sc, const ray_t *ray) {
    hit_t result;
    memset(&result, 0, sizeof(result));
    result.t = ray->tmax;
    uint32_t index_count = sc->mesh_index_count;
    if (index_count < 3u) return result;
    for (uint32_t tri = 0u; tri + 2u < index_count; tri += 3u) {
        uint32_t i0 = sc->mesh_indices[tri], i1 = sc->mesh_indices[tri + 1u], i2 = sc->mesh_indices[tri + 2u];
        if (i0 >= sc->mesh_vertex_count || i1 >= sc->mesh_vertex_count || i2 >= sc->mesh_vertex_count)
            continue;
        const float *p0 = &sc->mesh_vertices[3 * (size_t)i0];
        const float *p1 = &sc->mesh_vertices[3 * (size_t)i1];
        const float *p2 = &sc->mesh_vertices[3 * (size_t)i2];
        hit_t th = ray_triangle_intersect(ray, v3_make(p0[0], p0[1], p0[2]), v3_make(p1[0], p1[1], p1[2]),
                                          v3_make(p2[0], p2[1], p2[2]));
        if (th.hit != 0u && th.t < result.t) result = th;
    }
    return result;
}

/* intersect_hybrid, :175-201 */
static hit_t intersect_hybrid(const scene_t *sc, const ray_t *ray, counters_t *cnt) {
    hit_t best;
    memset(&best, 0, sizeof(best));
    best.t = ray->tmax;
    if (sc->traversal_mode == 0u || sc->traversal_mode == 2u) {
        hit_t mh = intersect_mesh(sc, ray);
        if (mh.hit != 0u && mh.t < best.t) best = mh;
    }
    if ((sc->traversal_mode == 0u || sc->traversal_mode == 3u) && sc->enabled) {
        ray_t tray = *ray;
        tray.tmax = best.t;
        hit_t th = terrain_trace(sc, &tray, 0, 0, cnt);
        if (th.hit != 0u && th.t < best.t) best = th;
    }
    return best;
}

/* intersect_hybrid_optimized, :204-235 */
static hit_t intersect_hybrid_optimized(const scene_t *sc, const ray_t *ray, float early_exit_distance,
                                        int apply_terrain_curvature, counters_t *cnt) {
    hit_t best;
    memset(&best, 0, sizeof(best));
    best.t = ray->tmax;
    if (sc->traversal_mode == 0u || sc->traversal_mode == 2u) {
        hit_t mh = intersect_mesh(sc, ray);
        if (mh.hit != 0u && mh.t < early_exit_distance) return mh;
        if (mh.hit != 0u && mh.t < best.t) best = mh;
    }
    if ((sc->traversal_mode == 0u || sc->traversal_mode == 3u) && sc->enabled) {
        ray_t tray = *ray;
        tray.tmax = best.t;
        hit_t th = terrain_trace(sc, &tray, 1, apply_terrain_curvature, cnt);
        if (th.hit != 0u && th.t < best.t) best = th;
    }
    return best;
}

/* get_surface_properties, :238-245 */
static inline v3 get_surface_properties(const scene_t *sc, const hit_t *hit) {
    if (hit->hit_type == 3u) return sc->albedo;
    return v3_make(0.7f, 0.7f, 0.8f);
}
/* intersect_shadow_ray, :248-251 */
static inline int intersect_shadow_ray(const scene_t *sc, const ray_t *ray, float max_distance, counters_t *cnt) {
    hit_t h = intersect_hybrid_optimized(sc, ray, 0.01f, 1, cnt);
    return h.hit != 0u && h.t < max_distance;
}
/* intersect_ibl_occlusion_ray, :256-259 */
static inline int intersect_ibl_occlusion_ray(const scene_t *sc, const ray_t *ray, float max_distance, counters_t *cnt) {
    hit_t h = intersect_hybrid_optimized(sc, ray, 0.01f, 0, cnt);
    return h.hit != 0u && h.t < max_distance;
}

/* ------------------------------------------------------------------------- */
/* shading helpers (hybrid_terrain_traversal.wgsl:392-431, hybrid_kernel.wgsl)  */
/* ------------------------------------------------------------------------- */
/* xorshift32, hybrid_kernel.wgsl:78-85 */
static inline float xorshift32(uint32_t *state) {
    uint32_t x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *state = x;
    return (float)x / 4294967296.0f;
}
/* terrain_env_radiance, :392-405 */
static inline v3 terrain_env_radiance(const scene_t *sc, v3 dir) {
    float intensity = sc->env_intensity;
    uint32_t ew = sc->env_w, eh = sc->env_h;
    if (ew == 0u || eh == 0u) return v3_make(intensity, intensity, intensity);
    v3 d = normalize3(dir);
    float uu = det_atan2f(d.z, d.x) / (2.0f * F3D_PI) + 0.5f;
    float vv = det_acosf(clampf(d.y, -1.0f, 1.0f)) / F3D_PI;
    uint32_t px = min_u32(sat_u32(uu * (float)ew), ew - 1u);
    uint32_t py = min_u32(sat_u32(vv * (float)eh), eh - 1u);
    const float *t = &sc->env[3 * ((size_t)py * ew + px)];
    return v3_make(t[0] * intensity, t[1] * intensity, t[2] * intensity);
}
/* terrain_tent_offset, :409-414 */
static inline float terrain_tent_offset(float u) {
    if (u < 0.5f) return sqrtf(2.0f * u) - 1.0f;
    return 1.0f - sqrtf(2.0f * (1.0f - u));
}
/* terrain_luminance, :416-418 */
static inline float terrain_luminance(v3 c) { return dot3(c, v3_make(0.2126f, 0.7152f, 0.0722f)); }
/* terrain_cosine_dir, :421-431 */
static inline v3 terrain_cosine_dir(v3 n, float u1, float u2) {
    float sign = n.z < 0.0f ? -1.0f : 1.0f;
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    v3 t = v3_make(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    v3 bt = v3_make(b, sign + n.y * n.y * a, -n.y);
    float r = sqrtf(u1);
    float sn, cs;
    f3do_sincos_2pi(u2, &sn, &cs);
    float lx = r * cs, ly = r * sn, lz = sqrtf(fmaxf(0.0f, 1.0f - u1));
    v3 o = v3_make(fmaf(lz, n.x, fmaf(ly, bt.x, lx * t.x)), fmaf(lz, n.y, fmaf(ly, bt.y, lx * t.y)),
                   fmaf(lz, n.z, fmaf(ly, bt.z, lx * t.z)));
    return normalize3(o);
}
/* reinhard_tonemap, hybrid_kernel.wgsl:109-112 */
static inline v3 reinhard_tonemap(v3 color, float exposure) {
    v3 e = v3_scale(color, exposure);
    return v3_make(e.x / (1.0f + e.x), e.y / (1.0f + e.y), e.z / (1.0f + e.z));
}
/* camera ray for pixel (gx,gy) with sub-pixel offset (jx,jy); main_terrain :481-485 */
static inline ray_t camera_ray(const uniforms_t *u, uint32_t gx, uint32_t gy, float jx, float jy) {
    float ndc_x = (((float)gx + 0.5f + jx) / (float)u->width) * 2.0f - 1.0f;
    float ndc_y = (1.0f - ((float)gy + 0.5f + jy) / (float)u->height) * 2.0f - 1.0f;
    v3 rd = normalize3(v3_make(ndc_x * u->half_w, ndc_y * u->half_h, -1.0f));
    v3 nf = v3_make(-u->cam_forward.x, -u->cam_forward.y, -u->cam_forward.z);
    v3 w = v3_make(fmaf(rd.z, nf.x, fmaf(rd.y, u->cam_up.x, rd.x * u->cam_right.x)),
                   fmaf(rd.z, nf.y, fmaf(rd.y, u->cam_up.y, rd.x * u->cam_right.y)),
                   fmaf(rd.z, nf.z, fmaf(rd.y, u->cam_up.z, rd.x * u->cam_right.z)));
    ray_t r;
    r.origin = u->cam_origin;
    r.tmin = 1e-3f;
    r.direction = normalize3(w);
    r.tmax = 1e30f;
    return r;
}

/* ------------------------------------------------------------------------- */
/* render state                                                               */
/* ------------------------------------------------------------------------- */
typedef struct {
    float *accum;   /* vec4 per pixel */
    float *welford; /* vec2 per pixel */
    f3do_reservoir *res_curr, *res_out, *res_prev;
    float *gbuffer_nr, *gbuffer_pos; /* vec4 per pixel */
    uint16_t *out_tex;               /* RGBA16F */
    uint16_t *aov_albedo, *aov_normal; /* RGBA16F */
    float *aov_depth;                /* R32F */
} state_t;

/* Every division of the ReSTIR reservoir arithmetic (candidate / merged / reused weights, the M-clamp scale, the
 * selection probability) is evaluated as a * (1 / b) with the reciprocal rounded once -- the lowering GPU compilers
 * give f32 division (WGSL allows it: 2.5 ULP) and the one pt_restir_spatial.wgsl:100 spells itself.  It is not a
 * detail: with spp = 1 the temporal pass of frame 1 compares two weights that are both EXACTLY 1 in real arithmetic
 * (pt_restir_temporal.wgsl:88, `rp.weight > rc.weight`: tp/tp against (sum of k times tp/tp)/k), and which side wins
 * fixes the reuse weight's start value, 0.96 or 1.6 on the golden scene, for a relaxation with a time constant of
 * 513 frames.  IEEE division resolves every such tie to `curr`; a * (1/b) sends 12.6 % of them to `prev`, and the
 * reference's golden shows 13 % (tools/golden_offset.py, DESIGN.md section 8.1: mean-abs against the golden 1.364 ->
 * 0.278, the one-sided +2.4 / 255 of rounds 1-5 gone). */
#ifdef F3DO_EXPERIMENT /* tools/golden_offset.py builds a second library with this; the shipped oracle has no such code */
static float experiment_div(float a, float b);
static int experiment_tie(size_t idx, float rp_weight, float rc_weight, int choose_prev);
#endif
static inline float restir_div(float a, float b) {
#ifdef F3DO_EXPERIMENT
    return experiment_div(a, b);
#else
    return a * (1.0f / b);
#endif
}
static inline float reservoir_weight(float w_sum, uint32_t m, float target_pdf) {
    return restir_div(w_sum, (float)m * target_pdf); /* :79-81 */
}

/* main_terrain, hybrid_terrain_traversal.wgsl:445-610 */
static void main_terrain_pixel(const scene_t *sc, const uniforms_t *un, state_t *st, uint32_t gx,
                               uint32_t gy, counters_t *cnt) {
    const uint32_t W = un->width, H = un->height;
    (void)H;
    const size_t pix = (size_t)gy * W + gx;

    f3do_reservoir prev_r = st->res_prev[pix];
    if (prev_r.m > TERRAIN_RESTIR_M_CAP) {
        float scale = restir_div((float)TERRAIN_RESTIR_M_CAP, (float)prev_r.m);
        prev_r.w_sum = prev_r.w_sum * scale;
        prev_r.m = TERRAIN_RESTIR_M_CAP;
        if (prev_r.target_pdf > 0.0f)
            prev_r.weight = reservoir_weight(prev_r.w_sum, prev_r.m, prev_r.target_pdf);
        st->res_prev[pix] = prev_r;
    }
    const int prev_valid = un->frame_index > 0u && prev_r.m > 0u && prev_r.weight > 0.0f &&
                           prev_r.target_pdf > 0.0f && prev_r.light_type == 1u;

    uint32_t rng = un->seed_hi ^ (gx * 1664525u) ^ (gy * 1013904223u) ^
                   (un->frame_index * 92837111u) ^ un->seed_lo;
    const uint32_t spp = sc->spp > 1u ? sc->spp : 1u;

    v3 frame_radiance = v3_make(0.0f, 0.0f, 0.0f);
    f3do_reservoir cand;
    memset(&cand, 0, sizeof(cand));

    for (uint32_t s = 0u; s < spp; s++) {
        float jx = terrain_tent_offset(xorshift32(&rng)) * 0.5f;
        float jy = terrain_tent_offset(xorshift32(&rng)) * 0.5f;
        ray_t ray = camera_ray(un, gx, gy, jx, jy);
        cnt->n_rays += 0;
        hit_t hit = intersect_hybrid(sc, &ray, cnt);
        if (hit.hit == 0u) {
            frame_radiance = v3_add(frame_radiance, terrain_env_radiance(sc, ray.direction));
            continue;
        }
        v3 n = hit.normal;
        v3 albedo = get_surface_properties(sc, &hit);

        v3 wi = normalize3(un->light_dir);
        float ndotl = fmaxf(dot3(n, wi), 0.0f);
        float target_pdf = terrain_luminance(v3_scale(v3_mul(albedo, un->light_color), ndotl));
        if (target_pdf > 0.0f) {
            cand.position[0] = hit.point.x; cand.position[1] = hit.point.y; cand.position[2] = hit.point.z;
            cand.light_index = 0u;
            cand.direction[0] = wi.x; cand.direction[1] = wi.y; cand.direction[2] = wi.z;
            cand.intensity = terrain_luminance(un->light_color);
            cand.light_type = 1u;
            cand.w_sum = cand.w_sum + target_pdf;
            cand.m = cand.m + 1u;
            cand.target_pdf = target_pdf;
        }

        v3 sun_dir = wi;
        float reuse_w = 1.0f;
        if (prev_valid) {
            sun_dir = normalize3(v3_make(prev_r.direction[0], prev_r.direction[1], prev_r.direction[2]));
            reuse_w = clampf(prev_r.weight, 0.0f, 4.0f);
        }
        v3 sun = v3_make(0.0f, 0.0f, 0.0f);
        float nd = fmaxf(dot3(n, sun_dir), 0.0f);
        v3 sorigin = v3_madd(hit.point, 1e-3f, n);
        if (nd > 0.0f) {
            ray_t sray = {sorigin, 1e-3f, sun_dir, 1e30f};
            float vis = 1.0f;
            if (un->shadows_enabled != 0u && intersect_shadow_ray(sc, &sray, 1e30f, cnt)) vis = 0.0f;
            sun = v3_scale(v3_scale(v3_scale(v3_mul(albedo, un->light_color), nd), vis), reuse_w);
        }

        float u1 = xorshift32(&rng);
        float u2 = xorshift32(&rng);
        v3 ei = terrain_cosine_dir(n, u1, u2);
        ray_t eray = {sorigin, 1e-3f, ei, 1e30f};
        float env_vis = 1.0f;
        if (intersect_ibl_occlusion_ray(sc, &eray, 1e30f, cnt)) env_vis = 0.0f;
        v3 ibl = v3_scale(v3_mul(albedo, terrain_env_radiance(sc, ei)), env_vis);

        frame_radiance = v3_add(v3_add(frame_radiance, sun), ibl);
    }
    frame_radiance = v3_make(frame_radiance.x / (float)spp, frame_radiance.y / (float)spp,
                             frame_radiance.z / (float)spp);

    if (cand.m > 0u && cand.w_sum > 0.0f && cand.target_pdf > 0.0f)
        cand.weight = reservoir_weight(cand.w_sum, cand.m, cand.target_pdf);
    st->res_curr[pix] = cand;

    float *acc = &st->accum[4 * pix];
    acc[0] += frame_radiance.x;
    acc[1] += frame_radiance.y;
    acc[2] += frame_radiance.z;
    acc[3] += 1.0f;

    const uint32_t window = sc->welford_window > 2u ? sc->welford_window : 2u;
    float wf_mean = st->welford[2 * pix], wf_m2 = st->welford[2 * pix + 1];
    if (un->frame_index % window == 0u) { wf_mean = 0.0f; wf_m2 = 0.0f; }
    v3 mean_rgb = v3_make(acc[0] / acc[3], acc[1] / acc[3], acc[2] / acc[3]);
    float mean_lum = terrain_luminance(mean_rgb);
    float k = (float)(un->frame_index % window) + 1.0f;
    float delta = mean_lum - wf_mean;
    float mean = wf_mean + delta / k;
    float m2 = fmaf(delta, mean_lum - mean, wf_m2);
    st->welford[2 * pix] = mean;
    st->welford[2 * pix + 1] = m2;

    v3 ldr = reinhard_tonemap(mean_rgb, un->cam_exposure);
    st->out_tex[4 * pix + 0] = f32_to_f16_bits(ldr.x);
    st->out_tex[4 * pix + 1] = f32_to_f16_bits(ldr.y);
    st->out_tex[4 * pix + 2] = f32_to_f16_bits(ldr.z);
    st->out_tex[4 * pix + 3] = f32_to_f16_bits(1.0f);

    if (un->aov_flags != 0u) { /* :583-609, centre ray */
        ray_t cray = camera_ray(un, gx, gy, 0.0f, 0.0f);
        hit_t chit = intersect_hybrid(sc, &cray, cnt);
        int is_hit = chit.hit != 0u;
        v3 calbedo = get_surface_properties(sc, &chit);
        if (chit.hit_type == 3u) calbedo = sc->albedo;
        v3 zero = v3_make(0.0f, 0.0f, 0.0f);
        v3 a = is_hit ? calbedo : zero;
        v3 nn = is_hit ? chit.normal : zero;
        st->aov_albedo[4 * pix + 0] = f32_to_f16_bits(a.x);
        st->aov_albedo[4 * pix + 1] = f32_to_f16_bits(a.y);
        st->aov_albedo[4 * pix + 2] = f32_to_f16_bits(a.z);
        st->aov_albedo[4 * pix + 3] = f32_to_f16_bits(1.0f);
        st->aov_normal[4 * pix + 0] = f32_to_f16_bits(nn.x);
        st->aov_normal[4 * pix + 1] = f32_to_f16_bits(nn.y);
        st->aov_normal[4 * pix + 2] = f32_to_f16_bits(nn.z);
        st->aov_normal[4 * pix + 3] = f32_to_f16_bits(1.0f);
        if (is_hit) st->aov_depth[pix] = chit.t;
        else { uint32_t q = 0x7fc00000u; memcpy(&st->aov_depth[pix], &q, 4); }
    }
}

/* main_terrain_gbuffer, :619-644 */
static void main_terrain_gbuffer_pixel(const scene_t *sc, const uniforms_t *un, state_t *st, uint32_t gx,
                                       uint32_t gy, counters_t *cnt) {
    const size_t pix = (size_t)gy * un->width + gx;
    ray_t ray = camera_ray(un, gx, gy, 0.0f, 0.0f);
    hit_t hit = intersect_hybrid(sc, &ray, cnt);
    float *nr = &st->gbuffer_nr[4 * pix], *pos = &st->gbuffer_pos[4 * pix];
    if (hit.hit != 0u) {
        nr[0] = hit.normal.x; nr[1] = hit.normal.y; nr[2] = hit.normal.z; nr[3] = 1.0f;
        pos[0] = hit.point.x; pos[1] = hit.point.y; pos[2] = hit.point.z; pos[3] = 1.0f;
    } else {
        nr[0] = 0.0f; nr[1] = 0.0f; nr[2] = 1.0f; nr[3] = 1.0f;
        pos[0] = pos[1] = pos[2] = pos[3] = 0.0f;
    }
}

/* pt_restir_temporal.wgsl:54-109 */
static void restir_temporal_pixel(state_t *st, size_t idx) {
    const f3do_reservoir rp = st->res_prev[idx];
    const f3do_reservoir rc = st->res_curr[idx];
    f3do_reservoir ro;
    memset(&ro, 0, sizeof(ro));
    int prev_valid = (rp.m > 0u) && (rp.weight > 0.0f) && (rp.target_pdf > 0.0f);
    int curr_valid = (rc.m > 0u) && (rc.weight > 0.0f) && (rc.target_pdf > 0.0f);
    if (!prev_valid && !curr_valid) { st->res_out[idx] = rc; return; }
    if (!prev_valid) { st->res_out[idx] = rc; return; }
    if (!curr_valid) { st->res_out[idx] = rp; return; }
    int choose_prev = rp.weight > rc.weight;
#ifdef F3DO_EXPERIMENT
    choose_prev = experiment_tie(idx, rp.weight, rc.weight, choose_prev);
#endif
    ro = choose_prev ? rp : rc; /* sample + target_pdf */
    ro.m = rp.m + rc.m;
    ro.w_sum = rp.w_sum + rc.w_sum;
    if (ro.w_sum > 0.0f && ro.target_pdf > 0.0f) ro.weight = restir_div(ro.w_sum, (float)ro.m * ro.target_pdf);
    else ro.weight = 0.0f;
    st->res_out[idx] = ro;
}

/* pt_restir_spatial.wgsl:45-111 consider_candidate, specialised to the scene the
 * driver binds: one directional light with importance 1 (render_terrain.rs:756-771),
 * one zeroed area light (:772-781). */
typedef struct {
    float wsum;
    f3do_reservoir chosen; /* only the sample fields are meaningful */
    float chosen_pdf;
    uint32_t seed;
} spatial_acc_t;

static void consider_candidate(const f3do_reservoir *r, const float *nr, spatial_acc_t *acc) {
    if (r->m == 0u) return;
    v3 N = normalize3(v3_make(nr[0], nr[1], nr[2]));
    float p_curr = 0.0f;
    if (r->light_type == 1u) {
        const uint32_t dir_count = 1u;
        const float sum_imp_dir = 0.0f + fmaxf(1.0f, 0.0f);
        float imp = fmaxf(1.0f, 0.0f);
        float p_sel = sum_imp_dir > 0.0f ? imp / fmaxf(sum_imp_dir, 1e-8f) : 1.0f / (float)dir_count;
        v3 wi = normalize3(v3_make(r->direction[0], r->direction[1], r->direction[2]));
        float cosTheta = fmaxf(dot3(N, wi), 0.0f);
        if (cosTheta <= 0.0f) return;
        p_curr = p_sel;
    } else if (r->light_type == 2u) {
        /* area light: never produced by main_terrain (light_type is 0 or 1) */
        return;
    } else {
        return;
    }
    if (p_curr <= 0.0f || r->target_pdf <= 0.0f) return;
    float w = r->w_sum * restir_div(p_curr, fmaxf(r->target_pdf, 1e-6f));
    if (w <= 0.0f) return;
    acc->wsum = acc->wsum + w;
    float u = xorshift32(&acc->seed);
    if (u < restir_div(w, acc->wsum)) {
        acc->chosen = *r;
        acc->chosen_pdf = p_curr;
    }
}

/* pt_restir_spatial.wgsl:158-222 */
static void restir_spatial_pixel(const uniforms_t *un, state_t *st, size_t idx) {
    const uint32_t W = un->width, H = un->height;
    const uint32_t x = (uint32_t)(idx % W), y = (uint32_t)(idx / W);
    const uint32_t K = 8u, R = 3u;
    spatial_acc_t acc;
    acc.seed = (un->seed_hi ^ un->frame_index) + (uint32_t)idx * 1664525u + 1013904223u;
    const f3do_reservoir r_self = st->res_out[idx];
    acc.chosen = r_self;
    acc.chosen_pdf = r_self.target_pdf;
    acc.wsum = 0.0f;
    uint32_t m_total = 0u;
    const float *nr = &st->gbuffer_nr[4 * idx];

    consider_candidate(&r_self, nr, &acc);
    m_total += r_self.m;
    for (uint32_t i = 0u; i < K; i++) {
        int rx = (int)floorf(xorshift32(&acc.seed) * (float)(2u * R + 1u)) - (int)R;
        int ry = (int)floorf(xorshift32(&acc.seed) * (float)(2u * R + 1u)) - (int)R;
        if (rx == 0 && ry == 0) continue;
        int nx = (int)x + rx, ny = (int)y + ry;
        if (nx < 0) nx = 0; if (nx > (int)W - 1) nx = (int)W - 1;
        if (ny < 0) ny = 0; if (ny > (int)H - 1) ny = (int)H - 1;
        size_t ni = (size_t)ny * W + (size_t)nx;
        const f3do_reservoir rn = st->res_out[ni];
        consider_candidate(&rn, nr, &acc);
        m_total += rn.m;
    }
    f3do_reservoir out_r = acc.chosen; /* sample fields */
    out_r.target_pdf = acc.chosen_pdf;
    out_r.w_sum = acc.wsum;
    out_r.m = m_total;
    if (out_r.w_sum > 0.0f && out_r.target_pdf > 0.0f)
        out_r.weight = restir_div(out_r.w_sum, (float)out_r.m * out_r.target_pdf);
    else
        out_r.weight = 0.0f;
    st->res_prev[idx] = out_r;
}

/* ------------------------------------------------------------------------- */
/* driver: HybridPathTracer::render_terrain_reference (render_terrain.rs:563-1434) */
/* ------------------------------------------------------------------------- */
static int finite3(const float *v) { return isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]); }
static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
#define FAIL(code, ...)                        \
    do {                                       \
        snprintf(err, errlen, __VA_ARGS__);    \
        rc = (code);                           \
        goto done;                             \
    } while (0)

static void fmt_rust_exp(char *buf, size_t n, double v, int prec) {
    /* Rust {:.Ne}: mantissa with N decimals, 'e', exponent without padding/plus */
    if (isinf(v)) { snprintf(buf, n, v < 0 ? "-inf" : "inf"); return; }
    if (isnan(v)) { snprintf(buf, n, "NaN"); return; }
    char tmp[64];
    snprintf(tmp, sizeof(tmp), "%.*e", prec, v);
    char *e = strchr(tmp, 'e');
    int ex = atoi(e + 1);
    *e = 0;
    snprintf(buf, n, "%se%d", tmp, ex);
}


/* ------------------------------------------------------------------------- */
/* AETHER aerial-perspective post (prometheus_aerial.wgsl + evaluation_core.wgsl) */
/* ------------------------------------------------------------------------- */
/* e^x with every operation spelled (the reference's det_exp is exp2(x*log2e) on the GPU driver; this fixed
 * polynomial is shared -- restated, not included -- with the HIP kernel so that both produce the same bits) */
static float ae_exp(float x) {
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
static float ae_clamp_scale(float v) { return fminf(fmaxf(v, 0.0f), 65504.0f); } /* evaluation_core.wgsl:29-33 */
static v3 ae_clamp_hdr(v3 c) { /* :35-37 */
    return v3_make(fminf(fmaxf(c.x, 0.0f), 65504.0f), fminf(fmaxf(c.y, 0.0f), 65504.0f), fminf(fmaxf(c.z, 0.0f), 65504.0f));
}
static float ae_mu_to_unit(float mu) { /* :82-90 */
    float b = clampf(mu, -1.0f, 1.0f), m = sqrtf(fabsf(b));
    return 0.5f * ((b >= 0.0f ? m : -m) + 1.0f);
}
static float ae_nu_to_unit(float nu) { return 1.0f - sqrtf(fmaxf(0.5f * (1.0f - clampf(nu, -1.0f, 1.0f)), 0.0f)); } /* :92-94 */
static int ae_round_index(float unit, uint32_t n) { return (int)rintf(unit * (float)((n > 1u ? n : 1u) - 1u)); }

/* aether_eval_sample_accumulated_scattering, :119-177 (table = rgba f32, x = view fastest, then sun, then h*nu+n) */
static v3 ae_scattering(const f3do_aether *A, float height_unit, float mu_sun, float mu_view, float nu) {
    const uint32_t sv = A->dims[2], ss = A->dims[3], sh = A->dims[4], sn = A->dims[5];
    int hc = (int)sh > 2 ? (int)sh : 2, nc = (int)sn > 2 ? (int)sn : 2;
    float c[4] = {ae_mu_to_unit(mu_view) * (float)(sv - 1u), ae_mu_to_unit(mu_sun) * (float)(ss - 1u),
                  sqrtf(clampf(height_unit, 0.0f, 1.0f)) * (float)(hc - 1), ae_nu_to_unit(nu) * (float)(nc - 1)};
    int lim[4] = {(int)sv - 1, (int)ss - 1, hc - 1, nc - 1}, lo[4], hi[4];
    float fr[4];
    for (int k = 0; k < 4; k++) {
        float fl = floorf(c[k]);
        lo[k] = (int)fl;
        hi[k] = lo[k] + 1 < lim[k] ? lo[k] + 1 : lim[k];
        fr[k] = c[k] - fl;
    }
    int depth = (int)(sh * sn);
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (int hs = 0; hs < 2; hs++)
        for (int ns = 0; ns < 2; ns++)
            for (int s2 = 0; s2 < 2; s2++)
                for (int vs = 0; vs < 2; vs++) {
                    int vi = vs ? hi[0] : lo[0], si = s2 ? hi[1] : lo[1], hh = hs ? hi[2] : lo[2], ni = ns ? hi[3] : lo[3];
                    float w = (vs ? fr[0] : 1.0f - fr[0]) * (s2 ? fr[1] : 1.0f - fr[1]) * (hs ? fr[2] : 1.0f - fr[2]) *
                              (ns ? fr[3] : 1.0f - fr[3]);
                    int x = vi, y = si, z = hh * nc + ni;
                    x = x < 0 ? 0 : (x > (int)sv - 1 ? (int)sv - 1 : x);
                    y = y < 0 ? 0 : (y > (int)ss - 1 ? (int)ss - 1 : y);
                    z = z < 0 ? 0 : (z > depth - 1 ? depth - 1 : z);
                    const float *t = A->scattering + 4 * (((size_t)z * ss + (size_t)y) * sv + (size_t)x);
                    ax = ax + w * t[0];
                    ay = ay + w * t[1];
                    az = az + w * t[2];
                }
    return v3_make(fmaxf(ax, 0.0f), fmaxf(ay, 0.0f), fmaxf(az, 0.0f));
}
static float ae_radius(float cam_h, float mu, float dist, float bottom) { /* :179-192 */
    float r = fmaxf(bottom, 1.0f) + clampf(cam_h, 0.0f, 100000.0f), d = clampf(dist, 0.0f, 20000000.0f);
    return sqrtf(fmaxf(r * r + d * d + 2.0f * r * d * clampf(mu, -1.0f, 1.0f), 0.0f));
}
static float ae_altitude(float cam_h, float mu, float dist, float bottom) { /* :194-203 */
    return clampf(ae_radius(cam_h, mu, dist, bottom) - fmaxf(bottom, 1.0f), 0.0f, 100000.0f);
}
static const float AE_WL[11] = {380.0f, 420.0f, 460.0f, 500.0f, 540.0f, 580.0f, 620.0f, 660.0f, 700.0f, 740.0f, 780.0f};
static const float AE_CIE[11][3] = {{0.001368f, 0.000039f, 0.006450f}, {0.134380f, 0.004000f, 0.645600f}, {0.290800f, 0.060000f, 1.669200f},
                                    {0.004900f, 0.323000f, 0.272000f}, {0.290400f, 0.954000f, 0.020300f}, {0.916300f, 0.870000f, 0.001650f},
                                    {0.854450f, 0.381000f, 0.000190f}, {0.164900f, 0.061000f, 0.000000f}, {0.011359f, 0.004102f, 0.000000f},
                                    {0.000690f, 0.000249f, 0.000000f}, {0.000042f, 0.000015f, 0.000000f}};
/* aether_eval_segment_transmittance, :238-344 */
static v3 ae_segment_transmittance(float dist, float cam_h, float mu, float bottom, float density_scale, float turbidity, float ozone_du) {
    float d = clampf(dist, 0.0f, 20000000.0f), ch = clampf(cam_h, 0.0f, 100000.0f), hs[16];
    for (int i = 0; i < 16; i++) hs[i] = ae_altitude(ch, mu, d * ((float)(2 * i + 1) * 0.03125f), bottom);
    float ray = ae_exp(-hs[0] / 8000.0f), mie = ae_exp(-hs[0] / 1200.0f), ozo = fmaxf(1.0f - fabsf((hs[0] - 25000.0f) / 15000.0f), 0.0f);
    for (int i = 1; i < 16; i++) ray = ray + ae_exp(-hs[i] / 8000.0f);
    for (int i = 1; i < 16; i++) mie = mie + ae_exp(-hs[i] / 1200.0f);
    for (int i = 1; i < 16; i++) ozo = ozo + fmaxf(1.0f - fabsf((hs[i] - 25000.0f) / 15000.0f), 0.0f);
    float per = d * density_scale * 0.0625f;
    float ray_col = per * ray, mie_col = per * mie, ozo_col = per * ozo * ozone_du / 300.0f;
    float X = 0.0f, Y = 0.0f, Z = 0.0f;
    for (int w = 0; w < 11; w++) { /* aether_eval_spectral_xyz, :50-80 */
        float ratio = 550.0f / AE_WL[w], r2 = ratio * ratio;
        float ray_beta = 1.2989e-5f * r2 * r2, mie_beta = 1.0e-5f * turbidity * ratio;
        float od = (AE_WL[w] - 600.0f) / 85.0f;
        float ozo_beta = 1.2e-6f * ae_exp(-0.5f * od * od);
        float t = ae_exp(-fmaxf(ray_beta * ray_col + mie_beta * mie_col + ozo_beta * ozo_col, 0.0f));
        float ew = (w == 0 || w == 10) ? 0.5f : 1.0f;
        float cx = AE_CIE[w][0] * t * ew, cy = AE_CIE[w][1] * t * ew, cz = AE_CIE[w][2] * t * ew;
        X = w == 0 ? cx : X + cx;
        Y = w == 0 ? cy : Y + cy;
        Z = w == 0 ? cz : Z + cz;
    }
    v3 xyz = v3_make(X, Y, Z); /* aether_eval_xyz_to_rgb, :42-48 */
    v3 rgb = v3_make(dot3(v3_make(3.2404542f, -1.5371385f, -0.4985314f), xyz) / 3.2613921f,
                     dot3(v3_make(-0.9692660f, 1.8760108f, 0.0415560f), xyz) / 2.5069624f,
                     dot3(v3_make(0.0556434f, -0.2040259f, 1.0572252f), xyz) / 2.3679786f);
    return v3_make(clampf(rgb.x, 0.0f, 1.0f), clampf(rgb.y, 0.0f, 1.0f), clampf(rgb.z, 0.0f, 1.0f));
}
/* the terrain-hit branch of prometheus_aerial.wgsl main (:160-226): surface radiance carried over `depth` along `ray` */
static v3 ae_transport(const f3do_aether *A, v3 surface, float cam_h, v3 ray, v3 sun, float depth, float sun_i) {
    float atm_h = fmaxf(A->top_radius_m - A->bottom_radius_m, 1.0f), cam_unit = clampf(cam_h / atm_h, 0.0f, 1.0f);
    float nu = dot3(ray, sun);
    float end_h = ae_altitude(cam_h, ray.y, depth, A->bottom_radius_m);
    float r = fmaxf(A->bottom_radius_m, 1.0f) + clampf(cam_h, 0.0f, 100000.0f), bd = clampf(depth, 0.0f, 20000000.0f);
    float end_r = fmaxf(ae_radius(cam_h, ray.y, bd, A->bottom_radius_m), 1.0f);
    float end_view_mu = clampf((r * clampf(ray.y, -1.0f, 1.0f) + bd) / end_r, -1.0f, 1.0f);
    float end_sun_mu = clampf((r * clampf(sun.y, -1.0f, 1.0f) + bd * clampf(nu, -1.0f, 1.0f)) / end_r, -1.0f, 1.0f);
    v3 seg = ae_segment_transmittance(depth, cam_h, ray.y, A->bottom_radius_m, 1.0f, A->turbidity, A->ozone_du);
    int bx = ae_round_index(0.5f * (clampf(ray.y, -1.0f, 1.0f) + 1.0f), A->dims[0]), by = ae_round_index(clampf(cam_unit, 0.0f, 1.0f), A->dims[1]);
    const float *bt = A->transmittance + 4 * ((size_t)by * A->dims[0] + (size_t)bx);
    v3 boundary = v3_make(clampf(bt[0], 0.0f, 1.0f), clampf(bt[1], 0.0f, 1.0f), clampf(bt[2], 0.0f, 1.0f));
    v3 cs = ae_scattering(A, cam_unit, sun.y, ray.y, nu);
    cs = v3_make(cs.x * sun_i, cs.y * sun_i, cs.z * sun_i);
    float end_unit = clampf(end_h / atm_h, 0.0f, 1.0f);
    v3 es = ae_scattering(A, end_unit, end_sun_mu, end_view_mu, nu);
    es = v3_make(es.x * sun_i, es.y * sun_i, es.z * sun_i);
    float dist_unit = depth / fmaxf(A->max_aerial_distance_m, 1.0f);
    int ax = ae_round_index(clampf(dist_unit, 0.0f, 1.0f), A->dims[6]), ay = ae_round_index(0.5f * (clampf(ray.y, -1.0f, 1.0f) + 1.0f), A->dims[7]),
        az = ae_round_index(clampf(cam_unit, 0.0f, 1.0f), A->dims[8]);
    float aerial_t = clampf(A->aerial[4 * (((size_t)az * A->dims[7] + (size_t)ay) * A->dims[6] + (size_t)ax) + 3], 0.0f, 1.0f);
    float mean_t = dot3(seg, v3_make(0.2126f, 0.7152f, 0.0722f));
    float k = aerial_t / fmaxf(mean_t, 1.0e-6f);
    v3 tr = v3_make(fmaxf(clampf(seg.x * k, 0.0f, 1.0f), boundary.x), fmaxf(clampf(seg.y * k, 0.0f, 1.0f), boundary.y),
                    fmaxf(clampf(seg.z * k, 0.0f, 1.0f), boundary.z));
    v3 ins = v3_make(fmaxf(cs.x - tr.x * es.x, 0.0f), fmaxf(cs.y - tr.y * es.y, 0.0f), fmaxf(cs.z - tr.z * es.z, 0.0f));
    return ae_clamp_hdr(v3_make(surface.x * tr.x + ins.x, surface.y * tr.y + ins.y, surface.z * tr.z + ins.z));
}
/* prometheus_aerial.wgsl main, :99-231: returns the Reinhard-mapped colour stored to the RGBA16F output */
static v3 ae_post_pixel(const f3do_aether *A, const uniforms_t *un, uint32_t gx, uint32_t gy, const float *acc, float depth,
                        int visible, float sun_intensity_in) {
    float den = fmaxf(acc[3], 1.0f);
    v3 surface = ae_clamp_hdr(v3_make(acc[0] / den, acc[1] / den, acc[2] / den));
    float ndc_x = (((float)gx + 0.5f) / (float)un->width) * 2.0f - 1.0f;
    float ndc_y = (1.0f - ((float)gy + 0.5f) / (float)un->height) * 2.0f - 1.0f;
    float aspect = (float)un->width / (float)un->height;
    float sx = ndc_x * un->half_h * aspect, sy = ndc_y * un->half_h;
    v3 ray = normalize3(v3_make(un->cam_right.x * sx + un->cam_up.x * sy + un->cam_forward.x,
                                un->cam_right.y * sx + un->cam_up.y * sy + un->cam_forward.y,
                                un->cam_right.z * sx + un->cam_up.z * sy + un->cam_forward.z));
    v3 sun = normalize3(un->light_dir);
    float sun_i = ae_clamp_scale(sun_intensity_in), exposure = ae_clamp_scale(un->cam_exposure);
    float atm_h = fmaxf(A->top_radius_m - A->bottom_radius_m, 1.0f);
    float cam_h = fmaxf(un->cam_origin.y, 0.0f), cam_unit = clampf(cam_h / atm_h, 0.0f, 1.0f);
    v3 hdr;
    if (!visible) {
        v3 s = ae_scattering(A, cam_unit, sun.y, ray.y, dot3(ray, sun));
        hdr = ae_clamp_hdr(v3_make(s.x * sun_i, s.y * sun_i, s.z * sun_i));
    } else {
        hdr = ae_transport(A, surface, cam_h, ray, sun, depth, sun_i);
    }
    v3 e = v3_make(hdr.x * exposure, hdr.y * exposure, hdr.z * exposure);
    return v3_make(e.x / (1.0f + e.x), e.y / (1.0f + e.y), e.z / (1.0f + e.z));
}

/* Test hooks: the two transport terms of the post for one view ray, without an image around them -- what
 * tests/test_aether.py compares with the vectors of the reference's own independent spectral oracle
 * (tests/golden/make_aether_independent_vectors.py).  f3do_aether_segment_transmittance = ae_segment_transmittance
 * (evaluation_core.wgsl:238-344); f3do_aether_sky = the sky branch of prometheus_aerial.wgsl (:150-158) for unit sun
 * intensity, before exposure / Reinhard. */
void f3do_aether_segment_transmittance(float dist, float cam_h, float mu, float bottom_radius_m, float turbidity, float ozone_du,
                                       float *rgb_out) {
    v3 t = ae_segment_transmittance(dist, cam_h, mu, bottom_radius_m, 1.0f, turbidity, ozone_du);
    rgb_out[0] = t.x; rgb_out[1] = t.y; rgb_out[2] = t.z;
}
void f3do_aether_sky(const f3do_aether *A, float cam_h, const float *ray, const float *sun, float *rgb_out) {
    v3 r = normalize3(v3_make(ray[0], ray[1], ray[2])), s = normalize3(v3_make(sun[0], sun[1], sun[2]));
    float atm_h = fmaxf(A->top_radius_m - A->bottom_radius_m, 1.0f);
    float cam_unit = clampf(fmaxf(cam_h, 0.0f) / atm_h, 0.0f, 1.0f);
    v3 c = ae_clamp_hdr(ae_scattering(A, cam_unit, s.y, r.y, dot3(r, s)));
    rgb_out[0] = c.x; rgb_out[1] = c.y; rgb_out[2] = c.z;
}

/* ... and the whole terrain-hit transport (surface * T + inscatter) for one ray, before exposure / Reinhard */
void f3do_aether_aerial(const f3do_aether *A, const float *surface, float cam_h, float depth, const float *ray, const float *sun, float sun_intensity,
                        float *rgb_out) {
    v3 c = ae_transport(A, v3_make(surface[0], surface[1], surface[2]), fmaxf(cam_h, 0.0f), normalize3(v3_make(ray[0], ray[1], ray[2])),
                        normalize3(v3_make(sun[0], sun[1], sun[2])), depth, ae_clamp_scale(sun_intensity));
    rgb_out[0] = c.x; rgb_out[1] = c.y; rgb_out[2] = c.z;
}

int f3do_render(const f3do_desc *d, f3do_out *out, char *err, size_t errlen) {
    int rc = 0;
    mips_t mips;
    memset(&mips, 0, sizeof(mips));
    state_t st;
    memset(&st, 0, sizeof(st));
    float *heights_copy = NULL;
    if (errlen) err[0] = 0;

    /* validate_desc, render_terrain.rs:474-557 (status 2 = RenderError::Render) */
    if (d->width == 0 || d->height == 0 || d->max_frames == 0)
        FAIL(2, "terrain reference requires non-zero width/height/max_frames");
    if (d->min_frames > d->max_frames)
        FAIL(2, "min_frames (%u) must be <= max_frames (%u)", d->min_frames, d->max_frames);
    if (d->spp == 0 || d->spp > 64) FAIL(2, "spp must be in 1..=64, got %u", d->spp);
    if (!(isfinite(d->exaggeration) && d->exaggeration > 0.0f))
        FAIL(2, "terrain exaggeration must be finite and > 0");
    if (!(finite3(d->cam_origin) && finite3(d->cam_look_at) && finite3(d->cam_up)))
        FAIL(2, "camera origin/look_at/up must be finite");
    v3 origin = v3_make(d->cam_origin[0], d->cam_origin[1], d->cam_origin[2]);
    v3 fwd_raw = v3_sub(v3_make(d->cam_look_at[0], d->cam_look_at[1], d->cam_look_at[2]), origin);
    if (sqrtf(dot3(fwd_raw, fwd_raw)) < 1e-6f) FAIL(2, "camera look_at must differ from origin");
    {
        v3 c = cross3(normalize3(fwd_raw), v3_make(d->cam_up[0], d->cam_up[1], d->cam_up[2]));
        if (sqrtf(dot3(c, c)) < 1e-6f)
            FAIL(2, "camera up vector must not be parallel to the view direction");
    }
    if (!(isfinite(d->fov_y_deg) && d->fov_y_deg > 0.0f && d->fov_y_deg < 180.0f))
        FAIL(2, "fov_y must be finite and in (0, 180) degrees, got %g", (double)d->fov_y_deg);
    if (!(isfinite(d->exposure) && d->exposure > 0.0f)) FAIL(2, "exposure must be finite and > 0");
    if (!(isfinite(d->sun_azimuth_deg) && isfinite(d->sun_elevation_deg)))
        FAIL(2, "sun azimuth/elevation must be finite");
    if (!(isfinite(d->sun_intensity) && d->sun_intensity >= 0.0f))
        FAIL(2, "sun intensity must be finite and >= 0");
    if (!finite3(d->sun_color) || d->sun_color[0] < 0.0f || d->sun_color[1] < 0.0f || d->sun_color[2] < 0.0f)
        FAIL(2, "sun color must have three finite non-negative components");
    if (!(isfinite(d->env_intensity) && d->env_intensity >= 0.0f))
        FAIL(2, "env intensity must be finite and >= 0");
    if (!(isfinite(d->variance_threshold) && d->variance_threshold > 0.0f))
        FAIL(2, "variance threshold must be finite and > 0");
    if (!(isfinite(d->spacing_x) && d->spacing_x > 0.0f && isfinite(d->spacing_z) && d->spacing_z > 0.0f))
        FAIL(2, "terrain spacing must be finite and > 0, got (%g, %g)", (double)d->spacing_x, (double)d->spacing_z);
    if (d->mesh_vertices || d->mesh_indices) {
        if (!d->mesh_vertices || d->mesh_vertex_count == 0)
            FAIL(2, "mesh vertices must be a non-empty flat [x,y,z] list");
        if (!d->mesh_indices || d->mesh_index_count == 0 || d->mesh_index_count % 3 != 0)
            FAIL(2, "mesh indices must be a non-empty multiple of 3");
        for (size_t i = 0; i < (size_t)d->mesh_vertex_count * 3; i++)
            if (!isfinite(d->mesh_vertices[i])) FAIL(2, "mesh vertices contain non-finite values");
        for (uint32_t i = 0; i < d->mesh_index_count; i++)
            if (d->mesh_indices[i] >= d->mesh_vertex_count)
                FAIL(2, "mesh indices reference out-of-bounds vertices");
    }
    /* clamps, render_terrain.rs:571-576 (AETHER_RADIOMETRIC_SCALE_MAX = 65504, src/core/atmosphere/mod.rs:11) */
    const float SCALE_MAX = 65504.0f;
    const float exposure = clampf(d->exposure, 0.0f, SCALE_MAX);
    const float sun_intensity = clampf(d->sun_intensity, 0.0f, SCALE_MAX);
    const float sun_color[3] = {clampf(d->sun_color[0], 0.0f, SCALE_MAX), clampf(d->sun_color[1], 0.0f, SCALE_MAX),
                                clampf(d->sun_color[2], 0.0f, SCALE_MAX)};
    const float env_intensity = clampf(d->env_intensity, 0.0f, SCALE_MAX);

    /* TerrainPtScene::new, terrain_heightfield.rs:390-494 (status 3 = Upload) */
    if (d->albedo[0] < 0.0f || d->albedo[1] < 0.0f || d->albedo[2] < 0.0f || !finite3(d->albedo))
        FAIL(3, "terrain albedo must be finite and >= 0");
    if (d->dem_w < 2 || d->dem_h < 2)
        FAIL(3, "terrain heightfield must be at least 2x2 texels, got %ux%u", d->dem_w, d->dem_h);
    for (size_t i = 0; i < (size_t)d->dem_w * d->dem_h; i++)
        if (!isfinite(d->heights[i])) FAIL(3, "terrain heightfield contains non-finite samples");
    if (d->env_map) {
        if (d->env_w == 0 || d->env_h == 0) FAIL(3, "env map dims do not match data length");
        for (size_t i = 0; i < (size_t)d->env_w * d->env_h * 3; i++)
            if (!isfinite(d->env_map[i])) FAIL(3, "env map contains non-finite samples");
    }
    if (mips_build(d->heights, d->dem_w, d->dem_h, &mips) != 0) FAIL(3, "min-max pyramid build failed");

    scene_t sc;
    memset(&sc, 0, sizeof(sc));
    sc.origin_x = -0.5f * ((float)d->dem_w - 1.0f) * d->spacing_x;
    sc.origin_z = -0.5f * ((float)d->dem_h - 1.0f) * d->spacing_z;
    sc.spacing_x = d->spacing_x; sc.spacing_z = d->spacing_z;
    sc.inv_spacing_x = 1.0f / d->spacing_x; sc.inv_spacing_z = 1.0f / d->spacing_z;
    sc.exaggeration = d->exaggeration; sc.env_intensity = env_intensity;
    sc.albedo = v3_make(d->albedo[0], d->albedo[1], d->albedo[2]);
    sc.dem_w = d->dem_w; sc.dem_h = d->dem_h; sc.cell_w = mips.cell_w; sc.cell_h = mips.cell_h;
    sc.mip_count = mips.count; sc.enabled = 1u;
    sc.env_w = d->env_map ? d->env_w : 0u; sc.env_h = d->env_map ? d->env_h : 0u;
    sc.spp = d->spp > 1u ? d->spp : 1u; sc.welford_window = 32u;
    sc.heights = d->heights; sc.mips = &mips; sc.env = d->env_map;
    sc.traversal_mode = d->mesh_vertices ? 0u : 3u;
    sc.mesh_vertices = d->mesh_vertices; sc.mesh_vertex_count = d->mesh_vertices ? d->mesh_vertex_count : 0u;
    sc.mesh_indices = d->mesh_indices; sc.mesh_index_count = d->mesh_vertices ? d->mesh_index_count : 0u;

    /* EarthCurvatureUniforms::new, terrain_heightfield.rs:52-84 */
    if (!isfinite(d->observer_lat_deg) || d->observer_lat_deg < -90.0 || d->observer_lat_deg > 90.0 ||
        !isfinite(d->observer_lon_deg) || d->observer_lon_deg < -180.0 || d->observer_lon_deg > 180.0)
        FAIL(2, "ray-origin latitude/longitude must be finite and in [-90,90]/[-180,180]");
    {
        double radius;
        char e2[160];
        if (f3do_effective_radius_m(d->earth_model, d->observer_lat_deg, d->sphere_radius_m,
                                    d->refraction_model, d->pressure_mbar, d->temperature_c,
                                    d->refraction_k, (double)d->sun_azimuth_deg, &radius, e2, sizeof(e2)))
            FAIL(2, "%s", e2);
        int enabled = isfinite(radius);
        sc.inv_two_r_prime = enabled ? (float)(0.5 / radius) : 0.0f;
        sc.curvature_enabled = enabled ? 1u : 0u;
    }

    /* camera + lighting, render_terrain.rs:635-717 */
    uniforms_t un;
    memset(&un, 0, sizeof(un));
    const float DEG = 0.017453292519943295f;
    v3 forward = normalize3(fwd_raw);
    v3 right = normalize3(cross3(forward, v3_make(d->cam_up[0], d->cam_up[1], d->cam_up[2])));
    v3 up = normalize3(cross3(right, forward));
    float az = d->sun_azimuth_deg * DEG, el = d->sun_elevation_deg * DEG;
    un.light_dir = v3_make(cosf(az) * cosf(el), sinf(el), sinf(az) * cosf(el));
    un.light_color = v3_make(sun_intensity * sun_color[0], sun_intensity * sun_color[1], sun_intensity * sun_color[2]);
    un.width = d->width; un.height = d->height;
    un.cam_origin = origin; un.cam_right = right; un.cam_up = up; un.cam_forward = forward;
    {
        float fov = d->fov_y_deg * DEG;
        float aspect = (float)d->width / (float)d->height;
        un.half_h = tanf(0.5f * fov);
        un.half_w = aspect * un.half_h;
    }
    un.cam_exposure = exposure;
    un.seed_hi = d->seed; un.seed_lo = d->seed ^ 0x85EBCA6Bu;
    un.shadows_enabled = 1u;

    const size_t P = (size_t)d->width * d->height;
    st.accum = (float *)calloc(P * 4, sizeof(float));
    st.welford = (float *)calloc(P * 2, sizeof(float));
    st.res_curr = (f3do_reservoir *)calloc(P, sizeof(f3do_reservoir));
    st.res_out = (f3do_reservoir *)calloc(P, sizeof(f3do_reservoir));
    st.res_prev = (f3do_reservoir *)calloc(P, sizeof(f3do_reservoir));
    st.gbuffer_nr = (float *)calloc(P * 4, sizeof(float));
    st.gbuffer_pos = (float *)calloc(P * 4, sizeof(float));
    st.out_tex = (uint16_t *)calloc(P * 4, sizeof(uint16_t));
    st.aov_albedo = (uint16_t *)calloc(P * 4, sizeof(uint16_t));
    st.aov_normal = (uint16_t *)calloc(P * 4, sizeof(uint16_t));
    st.aov_depth = (float *)calloc(P, sizeof(float));
    if (!st.accum || !st.welford || !st.res_curr || !st.res_out || !st.res_prev || !st.gbuffer_nr ||
        !st.gbuffer_pos || !st.out_tex || !st.aov_albedo || !st.aov_normal || !st.aov_depth)
        FAIL(2, "oracle: out of memory");

    counters_t total = {0, 0, 0, 0};
    const int W = (int)d->width, H = (int)d->height;

    /* one-shot G-buffer pass, render_terrain.rs:1091-1121 */
    {
        uint64_t a = 0, b = 0, c = 0, r = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : a, b, c, r)
        for (int y = 0; y < H; y++) {
            counters_t cn = {0, 0, 0, 0};
            for (int x = 0; x < W; x++) main_terrain_gbuffer_pixel(&sc, &un, &st, (uint32_t)x, (uint32_t)y, &cn);
            a += cn.n_node; b += cn.n_leaf; c += cn.n_hit; r += cn.n_rays;
        }
        /* g-buffer rays are not part of the per-sample accounting */
        (void)a; (void)b; (void)c; (void)r;
    }

    /* accumulation loop, render_terrain.rs:1123-1244 */
    uint32_t frames = 0u;
    float variance = INFINITY;
    int converged = 0;
    double t0 = now_seconds();
    while (frames < d->max_frames) {
        un.frame_index = frames;
        un.aov_flags = frames == 0u ? 0xFFu : 0u;
        uint64_t a = 0, b = 0, c = 0, r = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : a, b, c, r)
        for (int y = 0; y < H; y++) {
            counters_t cn = {0, 0, 0, 0};
            for (int x = 0; x < W; x++) main_terrain_pixel(&sc, &un, &st, (uint32_t)x, (uint32_t)y, &cn);
            a += cn.n_node; b += cn.n_leaf; c += cn.n_hit; r += cn.n_rays;
        }
        total.n_node += a; total.n_leaf += b; total.n_hit += c; total.n_rays += r;
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)P; i++) restir_temporal_pixel(&st, (size_t)i);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)P; i++) restir_spatial_pixel(&un, &st, (size_t)i);
        frames++;

        int window_full = (frames % 32u) == 0u;
        if (window_full || frames == d->max_frames) {
            uint32_t n_window = ((frames - 1u) % 32u) + 1u;
            if (n_window >= 2u) {
                float n = (float)n_window;
                float vmax = 0.0f;
                for (size_t i = 0; i < P; i++) {
                    float m2 = st.welford[2 * i + 1];
                    if (!isfinite(m2))
                        FAIL(2, "terrain PT produced non-finite variance (NaN in accumulation)");
                    vmax = fmaxf(vmax, m2 / (n - 1.0f));
                }
                variance = vmax;
                if (frames >= d->min_frames && variance < d->variance_threshold) {
                    converged = 1;
                    break;
                }
            }
        }
    }
    out->loop_seconds = now_seconds() - t0;
    out->frames = frames;
    out->variance = variance;
    out->converged = converged;
    out->n_node = total.n_node; out->n_leaf = total.n_leaf; out->n_hit = total.n_hit; out->n_rays = total.n_rays;
    out->n_samples = (uint64_t)P * sc.spp * frames;
    /* frame 0 also traced the centre AOV rays; they are included in n_rays/n_node. */
    out->minmax_pyramid_bytes = mips.bytes;
    if (!converged) {
        char vb[32], tb[32];
        fmt_rust_exp(vb, sizeof(vb), (double)variance, 3);
        fmt_rust_exp(tb, sizeof(tb), (double)d->variance_threshold, 1);
        FAIL(2,
             "terrain PT did not converge: per-pixel luminance variance %s over the last 32-frame window "
             "after %u frames (threshold %s); raise max_frames or simplify the scene \xe2\x80\x94 refusing to "
             "return a fake reference",
             vb, frames, tb);
    }

    /* reservoir validity, render_terrain.rs:1313-1337 */
    {
        int any_valid = 0;
        for (size_t i = 0; i < P; i++) {
            const f3do_reservoir *r = &st.res_prev[i];
            if (!(isfinite(r->w_sum) && isfinite(r->weight) && isfinite(r->target_pdf)))
                FAIL(2, "terrain PT reservoir bookkeeping produced non-finite values");
            if (r->m > 0u && r->weight > 0.0f && r->target_pdf > 0.0f) any_valid = 1;
        }
        int require = d->sun_elevation_deg > 0.0f && sun_intensity > 0.0f &&
                      (sun_color[0] > 0.0f || sun_color[1] > 0.0f || sun_color[2] > 0.0f);
        if (require && !any_valid)
            FAIL(2,
                 "terrain PT ReSTIR reuse chain produced no valid reservoirs for a sun-lit scene \xe2\x80\x94 "
                 "temporal/spatial reuse is broken");
    }

    /* composition hook: an external accumulation (f3d_oracle.h accum_override) takes the place of this render's own;
     * the RGBA16F target is rewritten exactly as main_terrain writes it (:557-579: Reinhard of the mean times exposure) */
    if (d->accum_override) {
        memcpy(st.accum, d->accum_override, P * 4 * sizeof(float));
        for (size_t i = 0; i < P; i++) {
            const float *acc = &st.accum[4 * i];
            v3 mean_rgb = v3_make(acc[0] / acc[3], acc[1] / acc[3], acc[2] / acc[3]);
            v3 ldr = reinhard_tonemap(mean_rgb, un.cam_exposure);
            st.out_tex[4 * i + 0] = f32_to_f16_bits(ldr.x);
            st.out_tex[4 * i + 1] = f32_to_f16_bits(ldr.y);
            st.out_tex[4 * i + 2] = f32_to_f16_bits(ldr.z);
        }
    }

    /* AETHER aerial-perspective post over the converged accumulation (render_terrain.rs:1249-1310) */
    if (d->atmosphere) {
        uniforms_t pu = un;
        for (size_t i = 0; i < P; i++) {
            int visible = !isnan(st.aov_depth[i]);
            v3 ldr = ae_post_pixel(d->atmosphere, &pu, (uint32_t)(i % d->width), (uint32_t)(i / d->width), &st.accum[4 * i],
                                   st.aov_depth[i], visible, sun_intensity);
            st.out_tex[4 * i + 0] = f32_to_f16_bits(ldr.x);
            st.out_tex[4 * i + 1] = f32_to_f16_bits(ldr.y);
            st.out_tex[4 * i + 2] = f32_to_f16_bits(ldr.z);
        }
    }

    /* readbacks + quantisation, render_terrain.rs:1340-1393 */
    for (size_t i = 0; i < P; i++) {
        for (int c = 0; c < 3; c++) {
            float v = f16_bits_to_f32(st.out_tex[4 * i + c]);
            out->rgba[4 * i + c] = (uint8_t)(clampf(v, 0.0f, 1.0f) * 255.0f + 0.5f);
            out->albedo[3 * i + c] = f16_bits_to_f32(st.aov_albedo[4 * i + c]);
            out->normal[3 * i + c] = f16_bits_to_f32(st.aov_normal[4 * i + c]);
        }
        out->rgba[4 * i + 3] = 255;
        out->depth[i] = st.aov_depth[i];
    }
    if (out->accum) memcpy(out->accum, st.accum, P * 4 * sizeof(float));
    if (out->welford) memcpy(out->welford, st.welford, P * 2 * sizeof(float));
    if (out->reservoir_prev) memcpy(out->reservoir_prev, st.res_prev, P * sizeof(f3do_reservoir));

done:
    free(st.accum); free(st.welford); free(st.res_curr); free(st.res_out); free(st.res_prev);
    free(st.gbuffer_nr); free(st.gbuffer_pos); free(st.out_tex); free(st.aov_albedo); free(st.aov_normal);
    free(st.aov_depth); free(heights_copy);
    mips_free(&mips);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* ray-batch hook (terrain_heightfield.rs:1646-1671, :1710-1745)               */
/* ------------------------------------------------------------------------- */
int f3do_terrain_trace_batch(const float *heights, uint32_t w, uint32_t h, float origin_x, float origin_z,
                             float spacing_x, float spacing_z, float exaggeration, float inv_two_r_prime,
                             uint32_t curvature_enabled, const float *rays, uint32_t n, int32_t any_hit,
                             int32_t apply_curvature, uint32_t *out_hit, float *out_t, float *out_normal,
                             uint64_t *counters3) {
    if (w < 2 || h < 2) return -10;
    mips_t mips;
    if (mips_build(heights, w, h, &mips) != 0) { mips_free(&mips); return -1; }
    scene_t sc;
    memset(&sc, 0, sizeof(sc));
    sc.origin_x = origin_x; sc.origin_z = origin_z; sc.spacing_x = spacing_x; sc.spacing_z = spacing_z;
    sc.inv_spacing_x = 1.0f / spacing_x; sc.inv_spacing_z = 1.0f / spacing_z;
    sc.exaggeration = exaggeration;
    sc.dem_w = w; sc.dem_h = h; sc.cell_w = mips.cell_w; sc.cell_h = mips.cell_h;
    sc.mip_count = mips.count; sc.enabled = 1u;
    sc.inv_two_r_prime = inv_two_r_prime; sc.curvature_enabled = curvature_enabled;
    sc.heights = heights; sc.mips = &mips; sc.traversal_mode = 3u;
    uint64_t a = 0, b = 0, c = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : a, b, c)
    for (long i = 0; i < (long)n; i++) {
        const float *r = &rays[8 * (size_t)i];
        ray_t ray = {v3_make(r[0], r[1], r[2]), r[3], v3_make(r[4], r[5], r[6]), r[7]};
        counters_t cn = {0, 0, 0, 0};
        hit_t hit = terrain_trace(&sc, &ray, any_hit != 0, apply_curvature != 0, &cn);
        out_hit[i] = hit.hit;
        if (out_t) out_t[i] = hit.t;
        if (out_normal) {
            out_normal[3 * i] = hit.hit ? hit.normal.x : 0.0f;
            out_normal[3 * i + 1] = hit.hit ? hit.normal.y : 0.0f;
            out_normal[3 * i + 2] = hit.hit ? hit.normal.z : 0.0f;
        }
        a += cn.n_node; b += cn.n_leaf; c += cn.n_hit;
    }
    if (counters3) { counters3[0] = a; counters3[1] = b; counters3[2] = c; }
    mips_free(&mips);
    return 0;
}

/* A heightfield that other oracles trace (oracle/wavefront_oracle.c: the PBR tracer's terrain primitive): placement as
 * f3do_render (centred on the world origin, y up, row = +z), terrain_trace with the curvature policy off. */
typedef struct {
    mips_t mips;
    scene_t sc;
    float *heights;
} f3do_terrain;
void *f3do_terrain_open(const float *heights, uint32_t w, uint32_t h, float spacing_x, float spacing_z, float exaggeration) {
    if (!heights || w < 2 || h < 2) return NULL;
    f3do_terrain *t = (f3do_terrain *)calloc(1, sizeof(f3do_terrain));
    if (!t) return NULL;
    t->heights = (float *)malloc((size_t)w * h * sizeof(float));
    if (!t->heights) { free(t); return NULL; }
    memcpy(t->heights, heights, (size_t)w * h * sizeof(float));
    if (mips_build(t->heights, w, h, &t->mips) != 0) { mips_free(&t->mips); free(t->heights); free(t); return NULL; }
    scene_t *sc = &t->sc;
    sc->origin_x = -0.5f * ((float)w - 1.0f) * spacing_x; /* terrain_heightfield.rs:359-362 */
    sc->origin_z = -0.5f * ((float)h - 1.0f) * spacing_z;
    sc->spacing_x = spacing_x; sc->spacing_z = spacing_z;
    sc->inv_spacing_x = 1.0f / spacing_x; sc->inv_spacing_z = 1.0f / spacing_z;
    sc->exaggeration = exaggeration;
    sc->dem_w = w; sc->dem_h = h; sc->cell_w = t->mips.cell_w; sc->cell_h = t->mips.cell_h;
    sc->mip_count = t->mips.count; sc->enabled = 1u;
    sc->inv_two_r_prime = 0.0f; sc->curvature_enabled = 0u;
    sc->heights = t->heights; sc->mips = &t->mips; sc->traversal_mode = 3u;
    return t;
}
void f3do_terrain_close(void *handle) {
    f3do_terrain *t = (f3do_terrain *)handle;
    if (!t) return;
    mips_free(&t->mips);
    free(t->heights);
    free(t);
}
/* closest hit in (tmin, tmax): returns 1 and t / normal; any_hit != 0: the first hit found decides */
int f3do_terrain_trace(const void *handle, const float *o, float tmin, const float *d, float tmax, int32_t any_hit, float *t_out, float *n_out) {
    const f3do_terrain *t = (const f3do_terrain *)handle;
    ray_t ray = {v3_make(o[0], o[1], o[2]), tmin, v3_make(d[0], d[1], d[2]), tmax};
    counters_t cn = {0, 0, 0, 0};
    hit_t hit = terrain_trace(&t->sc, &ray, any_hit != 0, 0, &cn);
    if (!hit.hit) return 0;
    if (t_out) *t_out = hit.t;
    if (n_out) { n_out[0] = hit.normal.x; n_out[1] = hit.normal.y; n_out[2] = hit.normal.z; }
    return 1;
}

int f3do_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* torch.distributed.run exports OMP_NUM_THREADS=1 to its workers; the benchmark's CPU leg sets its own team. */
void f3do_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------- */
/* tools/golden_offset.py only (-DF3DO_EXPERIMENT, a library of its own)       */
/* ------------------------------------------------------------------------- */
#ifdef F3DO_EXPERIMENT
/* How a conforming WGSL implementation may evaluate a / b in the reservoir arithmetic:
 *   0  a * (1/b), reciprocal correctly rounded (the shipped oracle)      1  IEEE a / b (the oracle of rounds 1-5)
 *   2  a * (reciprocal one ulp high)          3  a * (reciprocal one ulp low)
 *   4  a * (rcpss + one Newton step in plain f32 arithmetic)            5  the same step with fma
 * f3do_experiment_tie_prob >= 0: a pixel whose two weights agree to 4e-7 relative takes `prev` with this probability
 * (hash of the pixel), whatever the arithmetic said.  The counters see every such near-tie. */
int f3do_experiment_div_model = 0;
int f3do_experiment_libm = 0;
double f3do_experiment_tie_prob = -1.0;
unsigned f3do_experiment_ties = 0, f3do_experiment_ties_prev = 0;

static float experiment_div(float a, float b) {
    float r = 1.0f / b;
    uint32_t u;
    switch (f3do_experiment_div_model) {
    case 1: return a / b;
    case 2: memcpy(&u, &r, 4); u += 1u; memcpy(&r, &u, 4); return a * r;
    case 3: memcpy(&u, &r, 4); u -= 1u; memcpy(&r, &u, 4); return a * r;
    case 4: __asm__("rcpss %1, %0" : "=x"(r) : "x"(b)); r = r * (2.0f - b * r); return a * r;
    case 5: __asm__("rcpss %1, %0" : "=x"(r) : "x"(b)); r = fmaf(r, fmaf(-b, r, 1.0f), r); return a * r;
    default: return a * r;
    }
}

static int experiment_tie(size_t idx, float rp_weight, float rc_weight, int choose_prev) {
    if (fabsf(rp_weight - rc_weight) > 4.0e-7f * rc_weight) return choose_prev;
    if (f3do_experiment_tie_prob >= 0.0) {
        uint32_t h = (uint32_t)idx * 2654435761u ^ 0x9E3779B9u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        choose_prev = ((double)h / 4294967296.0) < f3do_experiment_tie_prob;
    }
    __atomic_fetch_add(&f3do_experiment_ties, 1u, __ATOMIC_RELAXED);
    if (choose_prev) __atomic_fetch_add(&f3do_experiment_ties_prev, 1u, __ATOMIC_RELAXED);
    return choose_prev;
}
#endif
