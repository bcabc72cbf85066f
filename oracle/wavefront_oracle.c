/* oracle/wavefront_oracle.c -- CPU restatement of forge3d's wavefront multi-bounce PBR path tracer
 * (SURVEY.md 8f row 3).  TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it as the checker; the product (forge3d_amd/) never does.
 *
 * Reference (Rust + wgpu + WGSL, not buildable here):
 *   src/path_tracing/adjudication.rs:76-364        render_pt_reference: per-frame seeds, frame loop, mean
 *   src/path_tracing/wavefront/render.rs:87-208    raygen, then <= 16 x { intersect, shade, shadow, scatter }
 *   src/shaders/pt_raygen.wgsl:75-226              camera ray, VDC/Halton + Cranley-Patterson + tent jitter
 *   src/shaders/pt_intersect.wgsl:99-218,339-558   spheres, instanced mesh (watertight triangle test)
 *   src/shaders/pt_shade.wgsl:43-862               NEE (environment mixture with MIS, directional, area discs),
 *                                                  Lambert / GGX metal / dielectric continuation, roulette
 *   src/shaders/pt_shadow.wgsl:151-294             any-hit visibility (Moller-Trumbore), accumulation
 *   src/shaders/pt_scatter.wgsl:76-133             miss -> background
 *   src/core/tonemap.rs:11-30                      Reinhard + sRGB resolve
 *
 * What is restated and what is not.  The reference moves rays through queues with atomics; a pixel's rays never
 * interact with another pixel's, so the queues are a schedule, not part of the result: here every pixel's path is
 * followed from the camera to its end.  Order of the sums: the reference executes `accum[p] = accum[p] + c` once per
 * contribution, from several queue items of one pixel in one dispatch -- a data race there whose order
 * (and, when updates collide, outcome) is not defined; the sum is what it means.  This oracle fixes ONE order: the
 * contributions of a frame's path are summed, starting from zero, in the order the stages would add them (emission,
 * environment, directional, area at every vertex; the miss of the last ray last), and the frame totals are added to
 * the pixel in frame order.  Frame totals do not depend on the running sum, which is what lets the device compute
 * them in any order and fold them afterwards.  ReSTIR guiding is switched off in render_pt_reference and is not restated;
 * the fog medium (pt_shade.wgsl:328-338, :500-520) and hair segments (pt_intersect.wgsl:21-57, :499-544; pt_shade.wgsl:
 * 708-729) are off / empty there too and are restated here behind scene fields that default to off (no golden of the
 * reference exercises them: pinned by properties only, tests/test_wavefront.py).  Mesh hits: the reference walks a BVH and keeps the first of
 * equal-t hits in ITS visit order; here all triangles of the BLAS are swept in index order (lowest index wins a
 * tie), and the BVH's box test is treated as conservative.
 *
 * Numerics contract (WGSL leaves these to the driver; this file and the HIP kernel fix the same choices):
 * IEEE f32, no contraction (-ffp-contract=off), dot() is the fma chain z,y,x of f3d_math.h, everything else is
 * spelled unfused, normalize(v) = v * (1 / sqrt(dot(v, v))), sin/cos(2 pi u), atan, exp, log by fixed polynomials,
 * pow(x, y) = exp(y log x) (integer powers by multiplication), tan(fov / 2) once on the host.
 *
 * PARITY PIN: the reference's golden tests/golden/adjudication/pt_reference.png (512 x 512, 4096 frames) with the
 * reference's own drift gate (SSIM >= 0.995, mean |d| <= 2.0): tests/test_wavefront.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

/* Sphere, pt_shade.wgsl:235-245 (the material table is the sphere table) */
typedef struct {
    float center[3], radius, albedo[3], metallic, roughness, ior, emissive[3], ax, ay;
} wfo_sphere;
typedef struct { float direction[3], intensity, color[3], importance; } wfo_dir_light;       /* pt_shade.wgsl:227-232 */
typedef struct { float position[3], radius, normal[3], intensity, color[3], importance; } wfo_area_light; /* :217-224 */
typedef struct {                                                                              /* pt_intersect.wgsl:73-83 */
    float object_to_world[16], world_to_object[16]; /* column-major mat4x4 */
    uint32_t blas_index, material_id;
} wfo_instance;
typedef struct { const float *vertices; uint32_t vertex_count; const uint32_t *indices; uint32_t triangle_count; } wfo_mesh;
struct wfo_hair { float p0[3], r0, p1[3], r1; uint32_t material_id, pad[3]; }; /* HairSegment, pt_intersect.wgsl:60-69 */
typedef struct {
    const wfo_sphere *spheres; uint32_t sphere_count;
    const wfo_mesh *meshes; uint32_t mesh_count;
    const wfo_instance *instances; uint32_t instance_count;
    const wfo_dir_light *dir_lights; uint32_t dir_light_count;
    const wfo_area_light *area_lights; uint32_t area_light_count;
    const float *object_importance; uint32_t importance_count;
    float env_ground[4], env_sky[4], miss_ground[4], miss_sky[4]; /* ReferenceEnvironment, pt_shade.wgsl:302-308 */
    float cam_origin[3], cam_right[3], cam_up[3], cam_forward[3]; /* WavefrontUniforms, adjudication.rs:22-38 */
    float cam_fov_y, cam_exposure;
    uint32_t seed_hi, seed_lo;
    /* Heightfield primitive (NOT in the reference's wavefront tracer; BASELINE.json configs[2] "GI" over a DEM): a handle of
     * f3do_terrain_open (oracle/f3d_oracle.c, linked into this library) = the terrain tracer's own terrain_trace
     * (hybrid_terrain_traversal.wgsl:254-372), curvature off; NULL = none.  Its hits use material slot terrain_material. */
    const void *terrain;
    uint32_t terrain_material;
    const struct wfo_hair *hair; uint32_t hair_count;  /* HairSegment buffer, pt_intersect.wgsl:330 */
    float medium_g, medium_sigma_t, medium_density, medium_enabled; /* MediumParams, pt_shade.wgsl:34-39 */
} wfo_scene;
extern int f3do_terrain_trace(const void *handle, const float *o, float tmin, const float *d, float tmax, int32_t any_hit, float *t_out, float *n_out);

/* ---- helpers --------------------------------------------------------------------------------- */
static inline v3 mk(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld(const float *p) { return mk(p[0], p[1], p[2]); }
static inline v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return mk(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 normalize3(v3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return scale(a, inv); }
static inline float mixf(float a, float b, float t) { return fmaf(b, t, a * (1.0f - t)); }
static inline v3 mix3(v3 a, v3 b, float t) { return mk(mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float saturate(float x) { return clampf(x, 0.0f, 1.0f); }
static inline float comp(v3 a, uint32_t k) { return k == 0u ? a.x : (k == 1u ? a.y : a.z); }
static inline v3 reflect3(v3 i, v3 n) { return sub(i, scale(n, 2.0f * dot3(n, i))); } /* WGSL reflect */

#define WF_PI 3.14159265358979323846f
#define WF_HALF_PI 1.57079632679489661923f
#define WF_QUARTER_PI 0.78539816339744830962f

static inline float poly_sin(float x) {
    float z = x * x;
    float p = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    return fmaf(p * z, x, x);
}
static inline float poly_cos(float x) {
    float z = x * x;
    float p = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    return fmaf(p, z * z, fmaf(-0.5f, z, 1.0f));
}
/* sin, cos of 2 pi u, u in [0, 1] */
static inline void sincos_turn(float u, float *s_out, float *c_out) {
    float a = 4.0f * u, k = rintf(a), x = (a - k) * WF_HALF_PI;
    float s = poly_sin(x), c = poly_cos(x);
    int q = ((int)k) & 3;
    *s_out = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    *c_out = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}
/* sin, cos of an angle in radians (|a| small multiples of pi): reduce to turns */
static inline void sincos_rad(float a, float *s_out, float *c_out) {
    float u = a * 0.15915494309189533577f;
    u = u - floorf(u);
    sincos_turn(u, s_out, c_out);
}
static inline float det_atan(float v) {
    float sign = 1.0f, x = v, y;
    if (v < 0.0f) { sign = -1.0f; x = -v; }
    if (x > 2.414213562373095f) { y = WF_HALF_PI; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = WF_QUARTER_PI; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    y = y + fmaf(p * z, x, x);
    return sign * y;
}
static inline uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float det_exp(float x) {
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
    if (e < -126) return (y * from_bits((uint32_t)(e + 64 + 127) << 23)) * 5.42101086242752217e-20f;
    return y * from_bits((uint32_t)(e + 127) << 23);
}
/* natural log of a positive normal float (cephes logf scheme) */
static inline float det_log(float x) {
    uint32_t b = bits_of(x);
    int e = (int)(b >> 23) - 126;                             /* x = m 2^e, m in [0.5, 1) */
    float m = from_bits((b & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else m = m - 1.0f;
    float z = m * m;
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
    float r = m + y;
    return fmaf(0.693359375f, fe, r);
}
/* pow(x, y) for x >= 0, y > 0 */
static inline float det_pow(float x, float y) {
    if (!(x > 0.0f)) return 0.0f;
    if (x < 1.17549435e-38f) return 0.0f;
    return det_exp(y * det_log(x));
}
static inline float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
static inline float pow16(float x) { float a = x * x; a = a * a; a = a * a; return a * a; }

/* xorshift32, pt_shade.wgsl:342-349 (pt_raygen.wgsl:75-82) */
static inline float xorshift32(uint32_t *state) {
    uint32_t x = *state;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    *state = x;
    return (float)x / 4294967296.0f;
}
/* render_pt_reference's per-frame seeds, adjudication.rs:222-228 */
static inline uint32_t splitmix32(uint32_t x) {
    x += 0x9E3779B9u;
    uint32_t z = x;
    z = (z ^ (z >> 16)) * 0x21F0AAADu;
    z = (z ^ (z >> 15)) * 0x735A2D97u;
    return z ^ (z >> 15);
}

/* ---- raygen, pt_raygen.wgsl ------------------------------------------------------------------- */
static inline float tent_filter(float u) {                     /* :88-93 */
    if (u < 0.5f) return sqrtf(2.0f * u) - 1.0f;
    return 1.0f - sqrtf(2.0f * (1.0f - u));
}
static inline float radical_inverse_vdc(uint32_t n) {          /* :98-107 */
    n = (n << 16) | (n >> 16);
    n = ((n & 0x55555555u) << 1) | ((n & 0xAAAAAAAAu) >> 1);
    n = ((n & 0x33333333u) << 2) | ((n & 0xCCCCCCCCu) >> 2);
    n = ((n & 0x0F0F0F0Fu) << 4) | ((n & 0xF0F0F0F0u) >> 4);
    n = ((n & 0x00FF00FFu) << 8) | ((n & 0xFF00FF00u) >> 8);
    return (float)n * 2.3283064365386963e-10f;
}
static inline float halton_base3(uint32_t i) {                 /* :109-122 */
    float f = 1.0f, r = 0.0f;
    uint32_t n = i;
    while (n != 0u) {
        f = f / 3.0f;
        r = r + (float)(n % 3u) * f;
        n = n / 3u;
    }
    return r;
}
static inline float cp_rotate(float u, float r) { float x = u + r; return x - floorf(x); } /* :156-160 */

typedef struct {
    v3 o, d, throughput;
    float tmin, tmax, pdf;
    uint32_t pixel, depth, rng_hi, rng_lo;
} ray_t;

typedef struct {
    uint32_t width, height, frame_index, seed_hi, seed_lo;
    v3 origin, right, up, forward;
    float half_h, half_w;
} frame_t;

static ray_t raygen(const frame_t *u, uint32_t pixel_idx) {    /* main, :162-226, spp = 1, qmc_mode = 0 */
    const uint32_t px = pixel_idx % u->width, py = pixel_idx / u->width;
    const uint32_t sample = 0u, sidx = sample + u->frame_index * 1u;
    const float u1 = radical_inverse_vdc(sidx), u2 = halton_base3(sidx);
    uint32_t rr_state = u->seed_lo ^ (px * 9781u) ^ (py * 6271u) ^ (u->seed_hi * 13007u);
    const float r1 = xorshift32(&rr_state), r2 = xorshift32(&rr_state);
    const float jx = tent_filter(cp_rotate(u1, r1)) * 0.5f, jy = tent_filter(cp_rotate(u2, r2)) * 0.5f;
    const float ndc_x = ((((float)px + 0.5f) + jx) / (float)u->width) * 2.0f - 1.0f;
    const float ndc_y = (1.0f - (((float)py + 0.5f) + jy) / (float)u->height) * 2.0f - 1.0f;
    v3 rd = normalize3(mk(ndc_x * u->half_w, ndc_y * u->half_h, -1.0f));
    const v3 nf = neg(u->forward);
    rd = normalize3(mk((rd.x * u->right.x + rd.y * u->up.x) + rd.z * nf.x, (rd.x * u->right.y + rd.y * u->up.y) + rd.z * nf.y,
                       (rd.x * u->right.z + rd.y * u->up.z) + rd.z * nf.z));
    ray_t r;
    r.o = u->origin; r.tmin = 1e-4f; r.d = rd; r.tmax = 1e30f;
    r.throughput = mk(1.0f, 1.0f, 1.0f); r.pdf = 1.0f; r.pixel = pixel_idx; r.depth = 0u;
    r.rng_hi = u->seed_hi ^ (pixel_idx * 9781u) ^ (u->frame_index * 6271u);
    r.rng_lo = u->seed_lo ^ sample;
    return r;
}

/* ---- intersect, pt_intersect.wgsl -------------------------------------------------------------- */
static inline float ray_sphere_t(v3 ro, v3 rd, v3 c, float r) { /* :416-429 */
    const v3 oc = sub(ro, c);
    const float b = dot3(oc, rd);
    const float cterm = dot3(oc, oc) - r * r;
    const float disc = b * b - cterm;
    if (disc <= 0.0f) return 1e30f;
    const float s = sqrtf(disc);
    const float t0 = -b - s, t1 = -b + s;
    if (t0 > 1e-3f) return t0;
    if (t1 > 1e-3f) return t1;
    return 1e30f;
}
static inline v3 xform_point(const float *m, v3 p) {           /* m * vec4(p, 1), :339-341 */
    return mk(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12] * 1.0f, ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13] * 1.0f,
              ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14] * 1.0f);
}
static inline v3 xform_vector(const float *m, v3 v) {          /* m * vec4(v, 0), :343-345 */
    return mk(((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * 0.0f, ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * 0.0f,
              ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * 0.0f);
}
static inline v3 xform_normal(const float *w2o, v3 n) {        /* transpose(world_to_object) * vec4(n, 0), :355-360 */
    const float *m = w2o;
    return normalize3(mk(((m[0] * n.x + m[1] * n.y) + m[2] * n.z) + m[3] * 0.0f, ((m[4] * n.x + m[5] * n.y) + m[6] * n.z) + m[7] * 0.0f,
                         ((m[8] * n.x + m[9] * n.y) + m[10] * n.z) + m[11] * 0.0f));
}
/* watertight ray/triangle, :113-178; returns 1 and t, geometric normal */
static inline int ray_triangle_intersect(v3 o, v3 d, float tmin, float tmax, v3 v0, v3 v1, v3 v2, float *t_out, v3 *n_out) {
    const v3 A = sub(v0, o), B = sub(v1, o), C = sub(v2, o);
    const float adx = fabsf(d.x), ady = fabsf(d.y), adz = fabsf(d.z);
    uint32_t kz = 2u, kx = 0u, ky = 1u;
    if (adx > ady && adx > adz) { kz = 0u; kx = 1u; ky = 2u; }
    else if (ady > adz) { kz = 1u; kx = 2u; ky = 0u; }
    const float Sz = 1.0f / comp(d, kz), Sx = comp(d, kx) * Sz, Sy = comp(d, ky) * Sz;
    const float ax = comp(A, kx) - Sx * comp(A, kz), ay = comp(A, ky) - Sy * comp(A, kz);
    const float bx = comp(B, kx) - Sx * comp(B, kz), by = comp(B, ky) - Sy * comp(B, kz);
    const float cx = comp(C, kx) - Sx * comp(C, kz), cy = comp(C, ky) - Sy * comp(C, kz);
    const float az = comp(A, kz) * Sz, bz = comp(B, kz) * Sz, cz = comp(C, kz) * Sz;
    const float U = (bx * cy) - (by * cx), V = (cx * ay) - (cy * ax), W = (ax * by) - (ay * bx);
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    const float det = (U + V) + W;
    if (det == 0.0f) return 0;
    const float T = (U * az + V * bz) + W * cz;
    const float t = T / det;
    if (t > tmin && t < tmax) {
        *t_out = t;
        *n_out = normalize3(cross3(sub(v1, v0), sub(v2, v0)));
        return 1;
    }
    return 0;
}
/* closest hit of one BLAS: bvh_intersect_mesh(_desc), :180-238 / 362-414 -- swept in index order, see the header */
static int mesh_closest(const wfo_mesh *m, v3 o, v3 d, float tmin, float tmax, float *t_out, v3 *n_out) {
    int any = 0;
    float best = tmax;
    for (uint32_t k = 0u; k < m->triangle_count; k++) {
        const uint32_t i0 = m->indices[3u * k], i1 = m->indices[3u * k + 1u], i2 = m->indices[3u * k + 2u];
        if (i0 >= m->vertex_count || i1 >= m->vertex_count || i2 >= m->vertex_count) continue;
        float t;
        v3 n;
        if (ray_triangle_intersect(o, d, tmin, best, ld(m->vertices + 3u * i0), ld(m->vertices + 3u * i1),
                                   ld(m->vertices + 3u * i2), &t, &n)) {
            best = t; *t_out = t; *n_out = n; any = 1;       /* t < best by the range test: strict, first wins ties */
        }
    }
    return any;
}
static inline v3 tangent_from_normal(v3 n) {                   /* :15-18 (carried in the hit record; unused by surfaces) */
    const v3 a = fabsf(n.x) > 0.9f ? mk(0.0f, 1.0f, 0.0f) : mk(1.0f, 0.0f, 0.0f);
    return normalize3(cross3(a, n));
}

typedef struct {
    v3 p, n, wo, throughput;
    float t, pdf;
    uint32_t mat, pixel, depth, rng_hi, rng_lo;
    uint32_t flags; /* bit 0 = is_hair */
    v3 tangent;
} hit_t;

/* ray_cylinder_segment, pt_intersect.wgsl:21-57 */
static int ray_cylinder_segment(const ray_t *ray, v3 p0, v3 p1, float r, float *t_out, v3 *n_out) {
    const v3 axis = sub(p1, p0);
    const float L = length3(axis);
    if (L < 1e-6f || r <= 0.0f) return 0;
    const v3 n = mk(axis.x / L, axis.y / L, axis.z / L);
    const v3 w0 = sub(ray->o, p0);
    const float d_par = dot3(ray->d, n);
    const v3 d_perp = sub(ray->d, scale(n, d_par)), w_perp = sub(w0, scale(n, dot3(w0, n)));
    const float A = dot3(d_perp, d_perp), B = 2.0f * dot3(d_perp, w_perp), C = dot3(w_perp, w_perp) - r * r;
    if (A < 1e-12f) return 0;
    const float disc = B * B - (4.0f * A) * C;
    if (disc < 0.0f) return 0;
    const float sdisc = sqrtf(fmaxf(disc, 0.0f));
    const float t0 = (-B - sdisc) / (2.0f * A), t1 = (-B + sdisc) / (2.0f * A);
    float thit = 1e30f;
    if (t0 > ray->tmin && t0 < ray->tmax) thit = t0;
    if (t1 > ray->tmin && t1 < thit) thit = t1;
    if (thit >= 1e20f) return 0;
    const float s = dot3(add(w0, scale(ray->d, thit)), n);
    if (s < 0.0f || s > L) return 0;
    *t_out = thit;
    *n_out = normalize3(add(w_perp, scale(d_perp, thit)));
    return 1;
}

/* main, :431-558.  Returns 1 on hit. */
static int intersect(const wfo_scene *sc, const ray_t *ray, hit_t *hit) {
    float t_best = 1e30f;
    v3 hit_normal = mk(0.0f, 1.0f, 0.0f);
    uint32_t material_idx = 0u;
    for (uint32_t i = 0u; i < sc->sphere_count; i++) {
        const wfo_sphere *s = &sc->spheres[i];
        const float t = ray_sphere_t(ray->o, ray->d, ld(s->center), s->radius);
        if (t >= ray->tmin && t < fminf(t_best, ray->tmax)) {
            t_best = t;
            const v3 hp = add(ray->o, scale(ray->d, t));
            hit_normal = normalize3(sub(hp, ld(s->center)));
            material_idx = i;
        }
    }
    if (sc->instance_count == 0u) {
        float t;
        v3 n;
        if (sc->mesh_count > 0u && mesh_closest(&sc->meshes[0], ray->o, ray->d, ray->tmin, ray->tmax, &t, &n) && t < t_best) {
            t_best = t; hit_normal = n; material_idx = 0u;
        }
    } else {
        for (uint32_t ii = 0u; ii < sc->instance_count; ii++) {
            const wfo_instance *inst = &sc->instances[ii];
            if (inst->blas_index >= sc->mesh_count) continue;
            const v3 o_obj = xform_point(inst->world_to_object, ray->o);
            const v3 d_obj = normalize3(xform_vector(inst->world_to_object, ray->d));
            float t;
            v3 n;
            if (mesh_closest(&sc->meshes[inst->blas_index], o_obj, d_obj, ray->tmin, ray->tmax, &t, &n) && t < t_best) {
                t_best = t;
                hit_normal = xform_normal(inst->world_to_object, n);
                material_idx = sc->sphere_count > 0u ? (inst->material_id < sc->sphere_count - 1u ? inst->material_id : sc->sphere_count - 1u) : 0u;
            }
        }
    }
    int is_hair = 0;
    v3 hair_tangent = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t hi = 0u; hi < sc->hair_count; hi++) { /* :499-521 */
        const struct wfo_hair *seg = &sc->hair[hi];
        const float r = fmaxf(0.0f, (0.5f * (seg->r0 + seg->r1)) * 1.0f); /* HAIR_RADIUS_SCALE = 1.0 */
        float t;
        v3 n;
        if (ray_cylinder_segment(ray, ld(seg->p0), ld(seg->p1), r, &t, &n) && t < t_best) {
            t_best = t; hit_normal = n;
            material_idx = sc->sphere_count > 0u ? (seg->material_id < sc->sphere_count - 1u ? seg->material_id : sc->sphere_count - 1u) : 0u;
            is_hair = 1;
            hair_tangent = normalize3(sub(ld(seg->p1), ld(seg->p0)));
        }
    }
    if (sc->terrain) { /* the heightfield: terrain_trace with tmax = the closest hit so far */
        const float o[3] = {ray->o.x, ray->o.y, ray->o.z}, d[3] = {ray->d.x, ray->d.y, ray->d.z};
        float t, n[3];
        if (f3do_terrain_trace(sc->terrain, o, ray->tmin, d, t_best, 0, &t, n) && t < t_best) {
            t_best = t; hit_normal = mk(n[0], n[1], n[2]);
            material_idx = sc->sphere_count > 0u ? (sc->terrain_material < sc->sphere_count - 1u ? sc->terrain_material : sc->sphere_count - 1u) : 0u;
            is_hair = 0;
        }
    }
    if (!(t_best < 1e20f)) return 0;
    hit->p = add(ray->o, scale(ray->d, t_best));
    hit->t = t_best;
    hit->n = hit_normal;
    hit->wo = normalize3(neg(ray->d));
    hit->mat = material_idx;
    hit->throughput = ray->throughput; hit->pdf = ray->pdf; hit->pixel = ray->pixel; hit->depth = ray->depth;
    hit->rng_hi = ray->rng_hi; hit->rng_lo = ray->rng_lo;
    hit->flags = is_hair ? 1u : 0u;                       /* :538-544 */
    hit->tangent = is_hair ? hair_tangent : tangent_from_normal(hit_normal);
    return 1;
}

/* ---- shadow, pt_shadow.wgsl --------------------------------------------------------------------- */
static inline int ray_sphere_any(v3 ro, v3 rd, v3 c, float r, float tmin, float tmax) { /* :151-166 */
    const v3 oc = sub(ro, c);
    const float b = dot3(oc, rd);
    const float cterm = dot3(oc, oc) - r * r;
    const float disc = b * b - cterm;
    if (disc <= 0.0f) return 0;
    const float s = sqrtf(disc);
    const float t0 = -b - s, t1 = -b + s;
    return (t0 > tmin && t0 < tmax) || (t1 > tmin && t1 < tmax);
}
static int mesh_any_hit(const wfo_mesh *m, v3 ro, v3 rd, float tmin, float tmax) {   /* :185-246, Moller-Trumbore */
    for (uint32_t k = 0u; k < m->triangle_count; k++) {
        const uint32_t i0 = m->indices[3u * k], i1 = m->indices[3u * k + 1u], i2 = m->indices[3u * k + 2u];
        if (i0 >= m->vertex_count || i1 >= m->vertex_count || i2 >= m->vertex_count) continue;
        const v3 v0 = ld(m->vertices + 3u * i0), v1 = ld(m->vertices + 3u * i1), v2 = ld(m->vertices + 3u * i2);
        const v3 e1 = sub(v1, v0), e2 = sub(v2, v0);
        const v3 h = cross3(rd, e2);
        const float a = dot3(e1, h);
        if (fabsf(a) < 1e-7f) continue;
        const float f = 1.0f / a;
        const v3 s = sub(ro, v0);
        const float u = f * dot3(s, h);
        if (u < 0.0f || u > 1.0f) continue;
        const v3 q = cross3(s, e1);
        const float v = f * dot3(rd, q);
        if (v < 0.0f || u + v > 1.0f) continue;
        const float t = f * dot3(e2, q);
        if (t > tmin && t < tmax) return 1;
    }
    return 0;
}
static int occluded(const wfo_scene *sc, v3 ro, v3 rd, float tmin, float tmax) {     /* main, :248-294 */
    if (sc->terrain) {
        const float o[3] = {ro.x, ro.y, ro.z}, d[3] = {rd.x, rd.y, rd.z};
        if (f3do_terrain_trace(sc->terrain, o, tmin, d, tmax, 1, NULL, NULL)) return 1;
    }
    for (uint32_t i = 0u; i < sc->sphere_count; i++)
        if (ray_sphere_any(ro, rd, ld(sc->spheres[i].center), sc->spheres[i].radius, tmin, tmax)) return 1;
    if (sc->instance_count == 0u) return sc->mesh_count > 0u && mesh_any_hit(&sc->meshes[0], ro, rd, tmin, tmax);
    for (uint32_t ii = 0u; ii < sc->instance_count; ii++) {
        const wfo_instance *inst = &sc->instances[ii];
        if (inst->blas_index >= sc->mesh_count) continue;
        const v3 ro_obj = xform_point(inst->world_to_object, ro);
        const v3 rd_obj = normalize3(xform_vector(inst->world_to_object, rd));
        if (mesh_any_hit(&sc->meshes[inst->blas_index], ro_obj, rd_obj, tmin, tmax)) return 1;
    }
    return 0;
}

/* ---- shade, pt_shade.wgsl ------------------------------------------------------------------------ */
typedef struct { v3 t, b, n; } basis_t;                        /* columns of make_tangent_basis, :351-360 */
static inline basis_t make_tangent_basis(v3 n) {
    const float sign = n.z < 0.0f ? -1.0f : 1.0f;
    const float a = -1.0f / (sign + n.z);
    const float b = (n.x * n.y) * a;
    basis_t r;
    r.t = mk(1.0f + ((sign * n.x) * n.x) * a, sign * b, -sign * n.x);
    r.b = mk(b, sign + (n.y * n.y) * a, -n.y);
    r.n = n;
    return r;
}
static inline v3 to_world(const basis_t *m, v3 v) {            /* basis * v, :374-376 */
    return mk((m->t.x * v.x + m->b.x * v.y) + m->n.x * v.z, (m->t.y * v.x + m->b.y * v.y) + m->n.y * v.z,
              (m->t.z * v.x + m->b.z * v.y) + m->n.z * v.z);
}
static inline v3 sample_cosine_hemisphere(float u1, float u2) { /* :362-370 */
    const float r = sqrtf(u1);
    float s, c;
    sincos_turn(u2, &s, &c);
    return mk(r * c, r * s, sqrtf(fmaxf(0.0f, 1.0f - u1)));
}
static inline v3 fresnel_schlick(float cos_theta, v3 F0) {     /* :378-381 */
    const float w = pow5(1.0f - saturate(cos_theta));
    return mk(F0.x + (1.0f - F0.x) * w, F0.y + (1.0f - F0.y) * w, F0.z + (1.0f - F0.z) * w);
}
static inline float ggx_D(float n_dot_h, float alpha) {        /* :383-389 */
    const float a2 = alpha * alpha, ndh2 = n_dot_h * n_dot_h;
    const float q = ndh2 * (a2 - 1.0f) + 1.0f;
    const float denom = WF_PI * (q * q);
    return a2 / fmaxf(denom, 1e-6f);
}
static inline float smith_G1(float n_dot_v, float alpha) {     /* :391-396 */
    const float a1 = alpha + 1.0f;
    const float k = (a1 * a1) / 8.0f;
    return n_dot_v / (n_dot_v * (1.0f - k) + k);
}
static inline float smith_G(float n_dot_l, float n_dot_v, float alpha) { return smith_G1(n_dot_l, alpha) * smith_G1(n_dot_v, alpha); }
static inline v3 sample_ggx_isotropic(float u1, float u2, float alpha) { /* :402-412 */
    const float a2 = alpha * alpha;
    const float cos_theta_h = sqrtf((1.0f - u1) / (1.0f + (a2 - 1.0f) * u1));
    const float sin_theta_h = sqrtf(fmaxf(0.0f, 1.0f - cos_theta_h * cos_theta_h));
    float s, c;
    sincos_turn(u2, &s, &c);
    return mk(sin_theta_h * c, sin_theta_h * s, cos_theta_h);
}
static inline float ggx_D_aniso(v3 h, v3 t, v3 b, v3 n, float ax, float ay) { /* :414-423 */
    const float hx = dot3(h, t), hy = dot3(h, b), hz = fmaxf(dot3(h, n), 0.0f);
    const float x2 = (hx * hx) / (ax * ax + 1e-8f), y2 = (hy * hy) / (ay * ay + 1e-8f);
    const float denom = (x2 + y2) + hz * hz;
    return 1.0f / fmaxf((((WF_PI * ax) * ay) * denom) * denom, 1e-6f);
}
static inline float smith_G1_aniso(v3 v, v3 t, v3 b, v3 n, float ax, float ay) { /* :425-433 */
    const float vx = dot3(v, t), vy = dot3(v, b), vz = fmaxf(dot3(v, n), 0.0f);
    const float alpha_v = sqrtf((vx * vx) * (ax * ax) + (vy * vy) * (ay * ay)) / fmaxf(vz, 1e-6f);
    return 2.0f / (1.0f + sqrtf(1.0f + alpha_v * alpha_v));
}
static inline float smith_G_aniso(v3 l, v3 v, v3 t, v3 b, v3 n, float ax, float ay) {
    return smith_G1_aniso(l, t, b, n, ax, ay) * smith_G1_aniso(v, t, b, n, ax, ay);
}
static inline v3 sample_ggx_anisotropic(float u1, float u2, float ax, float ay) { /* :443-455 */
    float s2, c2;
    sincos_turn(u2, &s2, &c2);                                 /* tan(2 pi u2) = sin / cos */
    float phi = det_atan((ay / fmaxf(ax, 1e-6f)) * (s2 / c2));
    if (u2 > 0.5f) phi = phi + WF_PI;
    float sinPhi, cosPhi;
    sincos_rad(phi, &sinPhi, &cosPhi);
    const float denom = (cosPhi * cosPhi) / fmaxf(ax * ax, 1e-8f) + (sinPhi * sinPhi) / fmaxf(ay * ay, 1e-8f);
    const float ratio = u1 / fmaxf(1.0f - u1, 1e-6f);          /* guarded_ggx_u1_ratio, :439-441 */
    const float cosTheta = 1.0f / sqrtf(1.0f + ratio * denom);
    const float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
    return mk(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}

typedef struct { v3 f; float pdf; } brdf_eval_t;
/* bsdf_eval_pdf, :43-100.  NB the reference reads the tangent frame as ROWS of the basis matrix (:73-75). */
static brdf_eval_t bsdf_eval_pdf(v3 wo, v3 wi, v3 n, v3 albedo, float metallic, float roughness, float ax, float ay) {
    brdf_eval_t out;
    const float n_dot_l = fmaxf(dot3(n, wi), 0.0f), n_dot_v = fmaxf(dot3(n, wo), 0.0f);
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) { out.f = mk(0.0f, 0.0f, 0.0f); out.pdf = 0.0f; return out; }
    const float kd = saturate(1.0f - metallic);
    const v3 fd = scale(mk(albedo.x / WF_PI, albedo.y / WF_PI, albedo.z / WF_PI), kd);
    const float pdf_d = n_dot_l / WF_PI;
    const float m = fmaxf(0.02f, roughness * roughness);
    const v3 h = normalize3(add(wi, wo));
    const float n_dot_h = fmaxf(dot3(n, h), 0.0f), v_dot_h = fmaxf(dot3(wo, h), 0.0f);
    float D, G;
    if (fabsf(ax - ay) < 1e-4f) {
        D = ggx_D(n_dot_h, m);
        G = smith_G(n_dot_l, n_dot_v, m);
    } else {
        const basis_t bs = make_tangent_basis(n);
        const v3 t = mk(bs.t.x, bs.b.x, bs.n.x), bb = mk(bs.t.y, bs.b.y, bs.n.y), nn = mk(bs.t.z, bs.b.z, bs.n.z);
        D = ggx_D_aniso(h, t, bb, nn, ax, ay);
        G = smith_G_aniso(wi, wo, t, bb, nn, ax, ay);
    }
    const float sm = saturate(metallic);
    const v3 F0 = mk(mixf(0.04f, albedo.x, sm), mixf(0.04f, albedo.y, sm), mixf(0.04f, albedo.z, sm));
    const v3 F = fresnel_schlick(v_dot_h, F0);
    const float spec = (D * G) / fmaxf((4.0f * n_dot_l) * n_dot_v, 1e-6f);
    const v3 fs = scale(F, spec);
    const float pdf_s = (D * n_dot_h) / fmaxf(4.0f * v_dot_h, 1e-6f);
    const float ks = 1.0f - kd;
    const float pdf_mix = kd * pdf_d + ks * pdf_s;
    out.f = add(fd, fs);
    out.pdf = fmaxf(pdf_mix, 1e-8f);
    return out;
}
static inline v3 env_color(const wfo_scene *sc, v3 wi) {       /* :105-108 */
    return mix3(ld(sc->env_ground), ld(sc->env_sky), 0.5f * (wi.y + 1.0f));
}
static inline v3 sample_power_cosine_about_up(float u1, float u2, float m) { /* :155-163 */
    float s, c;
    sincos_turn(u2, &s, &c);
    const float cosTheta = det_pow(1.0f - u1, 1.0f / (m + 1.0f));
    const float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
    return mk(sinTheta * c, cosTheta, sinTheta * s);
}
static inline float power_cosine_pdf_about_up(v3 w) {          /* :165-169, m = 16 */
    const float c = fmaxf(dot3(mk(0.0f, 1.0f, 0.0f), normalize3(w)), 0.0f);
    return ((16.0f + 1.0f) * pow16(c)) / (2.0f * WF_PI);
}
typedef struct { v3 wi; float pdf; } env_sample_t;
static env_sample_t sample_env_mixture(v3 n, const basis_t *basis, float u1, float u2, float u3) { /* :174-191 */
    env_sample_t out;
    const float p = 0.5f;
    if (u1 < p) {
        out.wi = sample_power_cosine_about_up(u2, u3, 16.0f);
    } else {
        out.wi = to_world(basis, sample_cosine_hemisphere(u2, u3));
    }
    const float pdf_up = power_cosine_pdf_about_up(out.wi);
    const float pdf_cos = fmaxf(dot3(n, out.wi), 0.0f) / WF_PI;
    out.pdf = p * pdf_up + (1.0f - p) * pdf_cos;
    return out;
}
typedef struct { v3 wi, Li; float pdf, dist, cos_on_light; } area_sample_t;
static area_sample_t sample_area_light_disc(v3 P, v3 N, const wfo_area_light *L, float u1, float u2) { /* :118-152 */
    area_sample_t out;
    memset(&out, 0, sizeof out);
    const v3 nL = normalize3(ld(L->normal));
    const basis_t bl = make_tangent_basis(nL);
    const v3 tL = mk(bl.t.x, bl.b.x, bl.n.x), bL = mk(bl.t.y, bl.b.y, bl.n.y);  /* rows again, :122-123 */
    const float rad = fmaxf(L->radius, 1e-6f);
    const float r = sqrtf(u1) * rad;
    float s, c;
    sincos_turn(u2, &s, &c);
    const float dx = r * c, dy = r * s;
    const v3 X = add(add(ld(L->position), scale(tL, dx)), scale(bL, dy));
    const v3 dir = sub(X, P);
    const float d = length3(dir);
    if (d <= 1e-6f) return out;
    const v3 wi = mk(dir.x / d, dir.y / d, dir.z / d);
    const float cos_surf = fmaxf(dot3(N, wi), 0.0f), cos_on_light = fmaxf(dot3(nL, neg(wi)), 0.0f);
    out.dist = d;
    if (cos_surf <= 0.0f || cos_on_light <= 0.0f) return out;
    const float area = (WF_PI * rad) * rad;
    const float p_area = 1.0f / area;
    out.wi = wi;
    out.pdf = (p_area * (d * d)) / fmaxf(cos_on_light, 1e-6f);
    out.cos_on_light = cos_on_light;
    out.Li = scale(ld(L->color), L->intensity);
    return out;
}
/* importance-weighted light pick shared by the directional and area blocks, :622-633 / :667-678 */
static uint32_t pick_light(uint32_t count, const float *importance, uint32_t stride, uint32_t *rng, float *sum_out) {
    float sum_imp = 0.0f;
    for (uint32_t i = 0u; i < count; i++) sum_imp = sum_imp + fmaxf(importance[i * stride], 0.0f);
    uint32_t idx = 0u;
    if (sum_imp > 0.0f) {
        const float rsel = xorshift32(rng) * sum_imp;
        float acc = 0.0f;
        for (uint32_t i = 0u; i < count; i++) {
            acc = acc + fmaxf(importance[i * stride], 0.0f);
            if (rsel <= acc) { idx = i; break; }
        }
    } else {
        const float f = floorf(xorshift32(rng) * (float)count);
        idx = f >= 4294967296.0f ? 0xFFFFFFFFu : (f > 0.0f ? (uint32_t)f : 0u);
    }
    *sum_out = sum_imp;
    return idx < count - 1u ? idx : count - 1u;
}

/* One hit through the shade stage (main, :460-862) with its shadow rays resolved on the spot (pt_shadow main).
 * Returns 1 and the next ray when the path continues. */
static int shade(const wfo_scene *sc, const frame_t *u, const hit_t *h, float *accum, ray_t *next) {
    const uint32_t mat_idx = h->mat < sc->sphere_count ? h->mat : 0u;
    const wfo_sphere *M = &sc->spheres[mat_idx];
    const v3 albedo = ld(M->albedo), emissive = ld(M->emissive);
    const float metallic = M->metallic, roughness = M->roughness, ior = M->ior;
    if (emissive.x > 0.0f || emissive.y > 0.0f || emissive.z > 0.0f) {
        const v3 e = mul(h->throughput, emissive);
        accum[0] = accum[0] + e.x; accum[1] = accum[1] + e.y; accum[2] = accum[2] + e.z;
    }
    uint32_t rng_state = h->rng_hi ^ (h->pixel * 26699u) ^ (u->frame_index * 30977u);
    const v3 n = normalize3(h->n), wo = normalize3(h->wo);
    const float n_dot_v = fmaxf(dot3(n, wo), 0.0f);
    /* homogeneous medium, :328-338 / :500-520 (media_transmittance = exp(-max(d, 0) max(mu, 0)); fixed-polynomial exp) */
    const int medium_on = sc->medium_enabled > 0.5f;
    const float mu = sc->medium_sigma_t * sc->medium_density;
    const float mtrans = medium_on ? det_exp(-fmaxf(h->t, 0.0f) * fmaxf(mu, 0.0f)) : 1.0f;
    if (medium_on && h->depth == 0u) { /* in-scattered environment on the primary segment */
        const v3 fog_col = scale(env_color(sc, neg(wo)), 1.0f - mtrans);
        accum[0] = accum[0] + fog_col.x; accum[1] = accum[1] + fog_col.y; accum[2] = accum[2] + fog_col.z;
    }
    const basis_t basis = make_tangent_basis(n);
    const float a = fmaxf(0.02f, roughness * roughness);
    const float ax = fmaxf(0.002f, M->ax), ay = fmaxf(0.002f, M->ay);
    const float sm = saturate(metallic);
    const v3 F0 = mk(mixf(0.04f, albedo.x, sm), mixf(0.04f, albedo.y, sm), mixf(0.04f, albedo.z, sm));
    const float imp = mat_idx < sc->importance_count ? sc->object_importance[mat_idx] : 1.0f;
    const v3 so = add(h->p, scale(n, 1e-3f));                 /* shadow ray origin */
    /* environment NEE, :565-595 */
    {
        const float u1 = xorshift32(&rng_state), u2 = xorshift32(&rng_state), u3 = xorshift32(&rng_state);
        const env_sample_t s = sample_env_mixture(n, &basis, u1, u2, u3);
        const float cos_surf = fmaxf(dot3(n, s.wi), 0.0f);
        if (cos_surf > 0.0f) {
            const v3 L_env = env_color(sc, s.wi);
            const brdf_eval_t br = bsdf_eval_pdf(wo, s.wi, n, albedo, metallic, roughness, ax, ay);
            const float w_mis = s.pdf / fmaxf(s.pdf + br.pdf, 1e-8f);
            const float k = (((cos_surf / fmaxf(s.pdf, 1e-8f)) * w_mis) * imp) * mtrans;
            const v3 contrib = scale(mul(mul(h->throughput, br.f), L_env), k);
            if (!occluded(sc, so, s.wi, 1e-3f, 1e30f)) {
                accum[0] = accum[0] + contrib.x; accum[1] = accum[1] + contrib.y; accum[2] = accum[2] + contrib.z;
            }
        }
    }
    /* directional NEE, :598-640 */
    if (sc->dir_light_count > 0u) {
        float sum_imp;
        const uint32_t idx = pick_light(sc->dir_light_count, &sc->dir_lights[0].importance, sizeof(wfo_dir_light) / 4u, &rng_state, &sum_imp);
        const wfo_dir_light *L = &sc->dir_lights[idx];
        const v3 wi = normalize3(neg(ld(L->direction)));
        const float cos_surf = fmaxf(dot3(n, wi), 0.0f);
        if (cos_surf > 0.0f) {
            const brdf_eval_t br = bsdf_eval_pdf(wo, wi, n, albedo, metallic, roughness, ax, ay);
            const v3 Li = scale(ld(L->color), L->intensity);
            const float p_sel = sum_imp > 0.0f ? fmaxf(L->importance, 0.0f) / fmaxf(sum_imp, 1e-8f) : 1.0f / (float)sc->dir_light_count;
            const float k = ((cos_surf / fmaxf(p_sel, 1e-8f)) * imp) * mtrans;
            const v3 contrib = scale(mul(mul(h->throughput, br.f), Li), k);
            if (!occluded(sc, so, wi, 1e-3f, 1e30f)) {
                accum[0] = accum[0] + contrib.x; accum[1] = accum[1] + contrib.y; accum[2] = accum[2] + contrib.z;
            }
        }
    }
    /* area-disc NEE, :643-697 */
    if (sc->area_light_count > 0u) {
        float sum_imp;
        const uint32_t idx = pick_light(sc->area_light_count, &sc->area_lights[0].importance, sizeof(wfo_area_light) / 4u, &rng_state, &sum_imp);
        const wfo_area_light *L = &sc->area_lights[idx];
        const float u1 = xorshift32(&rng_state), u2 = xorshift32(&rng_state);
        const area_sample_t s = sample_area_light_disc(h->p, n, L, u1, u2);
        if (s.pdf > 0.0f && s.cos_on_light > 0.0f) {
            const brdf_eval_t br = bsdf_eval_pdf(wo, s.wi, n, albedo, metallic, roughness, ax, ay);
            const float cos_surf = fmaxf(dot3(n, s.wi), 0.0f);
            if (cos_surf > 0.0f) {
                const float p_sel = sum_imp > 0.0f ? fmaxf(L->importance, 0.0f) / fmaxf(sum_imp, 1e-8f) : 1.0f / (float)sc->area_light_count;
                const float pdf_light = p_sel * s.pdf;
                const float w_mis = pdf_light / fmaxf(pdf_light + br.pdf, 1e-8f);
                const float k = (((cos_surf / fmaxf(pdf_light, 1e-8f)) * w_mis) * imp) * mtrans;
                const v3 contrib = scale(mul(mul(h->throughput, br.f), s.Li), k);
                if (!occluded(sc, so, s.wi, 1e-3f, s.dist - 1e-3f)) {
                    accum[0] = accum[0] + contrib.x; accum[1] = accum[1] + contrib.y; accum[2] = accum[2] + contrib.z;
                }
            }
        }
    }
    /* continuation, :699-806 */
    v3 wi, new_throughput;
    float pdf;
    if ((h->flags & 1u) == 1u) { /* Kajiya-Kay, :708-729 (pow(x, 20) and pow(x, 80) by squaring) */
        const v3 T = normalize3(h->tangent);
        const float u1 = xorshift32(&rng_state), u2 = xorshift32(&rng_state);
        wi = normalize3(to_world(&basis, sample_cosine_hemisphere(u1, u2)));
        const float lobe = fmaxf(0.0f, dot3(normalize3(reflect3(neg(wo), T)), wi));
        const float l2 = lobe * lobe, l4 = l2 * l2, l8 = l4 * l4, l16 = l8 * l8, l64 = (l16 * l16) * (l16 * l16);
        const float f1 = l16 * l4, f2 = l64 * l16;
        const float kd = 0.2f, ks = 1.0f - kd, spec = ks * (0.6f * f1 + 0.4f * f2);
        const v3 f = mk(kd * (albedo.x / WF_PI) + F0.x * spec, kd * (albedo.y / WF_PI) + F0.y * spec, kd * (albedo.z / WF_PI) + F0.z * spec);
        const float cos_theta = fmaxf(0.0f, dot3(n, wi));
        pdf = cos_theta / WF_PI + 1e-8f;
        new_throughput = scale(mul(h->throughput, f), cos_theta / pdf);
    } else if (metallic > 0.5f) {
        const float u1 = xorshift32(&rng_state), u2 = xorshift32(&rng_state);
        const v3 t = mk(basis.t.x, basis.b.x, basis.n.x), bb = mk(basis.t.y, basis.b.y, basis.n.y), nn = mk(basis.t.z, basis.b.z, basis.n.z);
        const int aniso = !(fabsf(ax - ay) < 1e-4f);
        v3 h_world;
        if (!aniso) {
            h_world = normalize3(to_world(&basis, sample_ggx_isotropic(u1, u2, a)));
        } else {
            const v3 hl = sample_ggx_anisotropic(u1, u2, ax, ay);
            h_world = normalize3(add(add(scale(t, hl.x), scale(bb, hl.y)), scale(nn, hl.z)));
        }
        wi = normalize3(reflect3(neg(wo), h_world));
        const float n_dot_l = fmaxf(dot3(n, wi), 0.0f), n_dot_h = fmaxf(dot3(n, h_world), 0.0f), v_dot_h = fmaxf(dot3(wo, h_world), 0.0f);
        if (!(n_dot_l > 0.0f && n_dot_v > 0.0f)) return 0;   /* invalid sample: `continue` */
        const float D = aniso ? ggx_D_aniso(h_world, t, bb, nn, ax, ay) : ggx_D(n_dot_h, a);
        const float G = aniso ? smith_G_aniso(wi, wo, t, bb, nn, ax, ay) : smith_G(n_dot_l, n_dot_v, a);
        const v3 F = fresnel_schlick(v_dot_h, F0);
        const v3 spec = scale(F, (D * G) / fmaxf((4.0f * n_dot_l) * n_dot_v, 1e-6f));
        pdf = (D * n_dot_h) / fmaxf(4.0f * v_dot_h, 1e-6f);
        new_throughput = scale(mul(h->throughput, spec), n_dot_l / fmaxf(pdf, 1e-6f));
    } else if (ior > 1.01f) {
        const float cosi = saturate(dot3(n, wo));
        const float r0 = (ior - 1.0f) / (ior + 1.0f);
        const float F0s = r0 * r0;
        const float F = F0s + (1.0f - F0s) * pow5(1.0f - cosi);
        const float uu = xorshift32(&rng_state);
        if (uu < F) {
            wi = normalize3(reflect3(neg(wo), n));
        } else {
            const int entering = dot3(n, wo) > 0.0f;
            const float eta = entering ? 1.0f / ior : ior / 1.0f;
            const v3 N = entering ? n : neg(n), I = neg(wo);
            const float ni = dot3(N, I);                      /* WGSL refract(I, N, eta) */
            const float kk = 1.0f - (eta * eta) * (1.0f - ni * ni);
            if (kk < 0.0f) wi = normalize3(reflect3(neg(wo), n));   /* total internal reflection: the fallback the reference intends */
            else wi = normalize3(sub(scale(I, eta), scale(N, eta * ni + sqrtf(kk))));
        }
        pdf = 1.0f;
        new_throughput = mul(h->throughput, mk(fmaxf(albedo.x, 0.0f), fmaxf(albedo.y, 0.0f), fmaxf(albedo.z, 0.0f)));
    } else {
        const float u1 = xorshift32(&rng_state), u2 = xorshift32(&rng_state);
        wi = normalize3(to_world(&basis, sample_cosine_hemisphere(u1, u2)));
        const float cos_theta = fmaxf(0.0f, dot3(n, wi));
        pdf = cos_theta / WF_PI + 1e-8f;
        const v3 brdf = mk(albedo.x / WF_PI, albedo.y / WF_PI, albedo.z / WF_PI);
        new_throughput = scale(mul(h->throughput, brdf), cos_theta / pdf);
    }
    /* Russian roulette, :808-827 (adaptive threshold 0) */
    float rr_scale = 1.0f;
    if (h->depth >= 4u) {
        const float max_c = fmaxf(new_throughput.x, fmaxf(new_throughput.y, new_throughput.z));
        const float q = clampf(1.0f - max_c, 0.0f, 0.95f);
        const float uu = xorshift32(&rng_state);
        if (uu < q) return 0;
        rr_scale = 1.0f / (1.0f - q);
    }
    if (!((h->depth + 1u) < 16u)) return 0;
    next->o = add(h->p, scale(normalize3(h->n), 1e-3f));
    next->tmin = 1e-3f; next->d = wi; next->tmax = 1e30f;
    next->throughput = scale(new_throughput, rr_scale);
    next->pdf = pdf; next->pixel = h->pixel; next->depth = h->depth + 1u;
    next->rng_hi = rng_state; next->rng_lo = h->rng_lo ^ u->seed_lo;
    return 1;
}

/* Add frames [first_frame, first_frame + frame_count) to the running sums `accum` (W*H*4 floats, alpha untouched). */
int wfo_render(const wfo_scene *sc, uint32_t width, uint32_t height, uint32_t first_frame, uint32_t frame_count, float *accum) {
    if (!sc || !accum || width == 0u || height == 0u) return 1;
    const float half_h = tanf(0.5f * sc->cam_fov_y);
    const float aspect = (float)width / (float)height;
    const int64_t pixels = (int64_t)width * height;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t p = 0; p < pixels; p++) {
        float *acc = accum + 4 * p;
        for (uint32_t f = first_frame; f < first_frame + frame_count; f++) {
            frame_t u;
            u.width = width; u.height = height; u.frame_index = f;
            u.seed_hi = splitmix32(sc->seed_hi ^ f);                    /* adjudication.rs:231-232 */
            u.seed_lo = splitmix32(sc->seed_lo ^ (f * 0x00009E3Du));
            u.origin = ld(sc->cam_origin); u.right = ld(sc->cam_right); u.up = ld(sc->cam_up); u.forward = ld(sc->cam_forward);
            u.half_h = half_h; u.half_w = aspect * half_h;
            float total[3] = {0.0f, 0.0f, 0.0f};                        /* this frame's contributions, in stage order */
            ray_t ray = raygen(&u, (uint32_t)p);
            for (uint32_t it = 0u; it < 16u; it++) {                   /* MAX_DEPTH * 2 iterations, render.rs:115 */
                hit_t hit;
                if (!intersect(sc, &ray, &hit)) {                      /* pt_scatter.wgsl:113-131 */
                    const v3 sky = mix3(ld(sc->miss_ground), ld(sc->miss_sky), 0.5f * (ray.d.y + 1.0f));
                    const v3 c = mul(ray.throughput, sky);
                    total[0] = total[0] + c.x; total[1] = total[1] + c.y; total[2] = total[2] + c.z;
                    break;
                }
                ray_t next;
                if (!shade(sc, &u, &hit, total, &next)) break;
                ray = next;
            }
            acc[0] = acc[0] + total[0]; acc[1] = acc[1] + total[1]; acc[2] = acc[2] + total[2];
        }
    }
    return 0;
}

/* adjudication.rs:296-306 (mean, alpha 1) + core/tonemap.rs:11-30 (Reinhard, sRGB, u8) */
void wfo_resolve(const float *accum, uint64_t pixels, uint32_t frames, float exposure, float *hdr, uint8_t *rgba) {
    const float inv = 1.0f / (float)frames;
    for (uint64_t p = 0; p < pixels; p++) {
        for (int c = 0; c < 3; c++) {
            const float m = accum[4 * p + c] * inv;
            if (hdr) hdr[4 * p + c] = m;
            if (rgba) {
                const float x = fmaxf(m, 0.0f) * exposure;
                const float t = x / (1.0f + x);
                const float s = t <= 0.0031308f ? 12.92f * t : 1.055f * det_pow(t, 1.0f / 2.4f) - 0.055f;
                rgba[4 * p + c] = (uint8_t)(clampf(s, 0.0f, 1.0f) * 255.0f + 0.5f);
            }
        }
        if (hdr) hdr[4 * p + 3] = 1.0f;
        if (rgba) rgba[4 * p + 3] = 255u;
    }
}
