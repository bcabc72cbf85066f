/* oracle/smoke_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked or imported by the product).
 *
 * Restates the reference's CPU smoke ray-marcher, one function per reference function:
 *   SmokeVolume::raymarch_rgba             src/smoke/render.rs:7-96
 *   SmokeVolume::raymarch_projection_rgba  src/smoke/render.rs:98-178
 *   sample_render_fields                   src/smoke/render.rs:180-190
 *   march_ray_rgba                         src/smoke/render.rs:192-288
 *   sun_transmittance                      src/smoke/render.rs:290-330
 *   smoke_color / ray_box_intersection / henyey_greenstein / tone_map / render_smoothstep / to_u8
 *                                          src/smoke/render.rs:343-418
 *   sample_scalar / lerp / hash01          src/smoke/sampling.rs:1-34, :88-103
 *   grid_coord_from_world / bounds_*       src/smoke/types.rs:375-397
 *   SmokeRenderSettings::validate          src/smoke/types.rs:271-316
 * PARITY PIN: the reference (Rust) cannot be built here and ships no golden image of this path; the oracle is
 * pinned by the reference's own unit tests restated as known-answer properties (render.rs:420-592) in
 * tests/test_smoke.py -- "parity pinned by KAT properties only" (DESIGN.md).  Arithmetic: IEEE f32, no
 * contraction (-ffp-contract=off), glam's scalar Vec3 operation order; e^x is the fixed polynomial exp_det
 * shared (restated, not included) with the HIP kernel, powf(d, 1.5) = d * sqrtf(d): both within 2 ulp of the
 * libm calls Rust makes, which the 8-bit outputs absorb. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef struct {
    const float *density, *temperature, *soot, *humidity, *emission, *age;
    uint32_t dims[3];
    float voxel_size[3], origin[3];
    uint32_t frame_index;
} smoke_volume;

typedef struct {
    float density_scale, extinction, scattering, absorption, phase_g, step_size;
    uint32_t max_steps;
    int32_t self_shadow;
    uint32_t shadow_steps;
    float shadow_step_size, jitter_strength, exposure;
    float thin_color[3], dense_color[3];
    float soot_absorption, fire_glow;
} smoke_settings;

typedef struct { float x, y, z; } v3;

static v3 v3_add(v3 a, v3 b) { return (v3){a.x + b.x, a.y + b.y, a.z + b.z}; }
static v3 v3_sub(v3 a, v3 b) { return (v3){a.x - b.x, a.y - b.y, a.z - b.z}; }
static v3 v3_mul(v3 a, v3 b) { return (v3){a.x * b.x, a.y * b.y, a.z * b.z}; }
static v3 v3_scale(v3 a, float s) { return (v3){a.x * s, a.y * s, a.z * s}; }
static float v3_dot(v3 a, v3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
static v3 v3_cross(v3 a, v3 b) { return (v3){a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
static v3 normalize_or_zero(v3 a) { /* glam Vec3::normalize_or_zero */
    float rcp = 1.0f / sqrtf(v3_dot(a, a));
    if (isfinite(rcp) && rcp > 0.0f) return v3_scale(a, rcp);
    return (v3){0.0f, 0.0f, 0.0f};
}
static v3 normalize(v3 a) { return v3_scale(a, 1.0f / sqrtf(v3_dot(a, a))); }
static float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static float lerpf(float a, float b, float t) { return a + (b - a) * t; }

/* e^x, fixed polynomial (cephes expf scheme), every operation spelled */
static float exp_det(float x) {
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
    if (e < -126) { /* two-step scaling keeps the intermediate normal */
        union { uint32_t u; float f; } a = {(uint32_t)(e + 64 + 127) << 23};
        return (y * a.f) * 5.42101086242752217e-20f; /* 2^-64 */
    }
    union { uint32_t u; float f; } s = {(uint32_t)(e + 127) << 23};
    return y * s.f;
}

static float hash01(uint32_t v) {
    v ^= v >> 16;
    v *= 0x7FEB352Du;
    v ^= v >> 15;
    v *= 0x846CA68Bu;
    v ^= v >> 16;
    return (float)v / 4294967296.0f; /* u32::MAX as f32 rounds to 2^32 */
}

static size_t vidx(const uint32_t d[3], uint32_t x, uint32_t y, uint32_t z) { return ((size_t)z * d[1] + y) * d[0] + x; }

static float sample_scalar(const float *f, const uint32_t d[3], const float p[3]) {
    float x = clampf(p[0], 0.0f, (float)(d[0] - 1)), y = clampf(p[1], 0.0f, (float)(d[1] - 1)),
          z = clampf(p[2], 0.0f, (float)(d[2] - 1));
    uint32_t x0 = (uint32_t)floorf(x), y0 = (uint32_t)floorf(y), z0 = (uint32_t)floorf(z);
    uint32_t x1 = x0 + 1 < d[0] - 1 ? x0 + 1 : d[0] - 1, y1 = y0 + 1 < d[1] - 1 ? y0 + 1 : d[1] - 1,
             z1 = z0 + 1 < d[2] - 1 ? z0 + 1 : d[2] - 1;
    float fx = x - (float)x0, fy = y - (float)y0, fz = z - (float)z0;
    float c00 = lerpf(f[vidx(d, x0, y0, z0)], f[vidx(d, x1, y0, z0)], fx);
    float c10 = lerpf(f[vidx(d, x0, y1, z0)], f[vidx(d, x1, y1, z0)], fx);
    float c01 = lerpf(f[vidx(d, x0, y0, z1)], f[vidx(d, x1, y0, z1)], fx);
    float c11 = lerpf(f[vidx(d, x0, y1, z1)], f[vidx(d, x1, y1, z1)], fx);
    return lerpf(lerpf(c00, c10, fy), lerpf(c01, c11, fy), fz);
}

typedef struct { float density, temperature, soot, humidity, emission, age; } render_sample;

static render_sample sample_render_fields(const smoke_volume *v, v3 pos) {
    float p[3] = {(pos.x - v->origin[0]) / v->voxel_size[0] - 0.5f, (pos.y - v->origin[1]) / v->voxel_size[1] - 0.5f,
                  (pos.z - v->origin[2]) / v->voxel_size[2] - 0.5f};
    render_sample s;
    s.density = sample_scalar(v->density, v->dims, p);
    s.temperature = sample_scalar(v->temperature, v->dims, p);
    s.soot = sample_scalar(v->soot, v->dims, p);
    s.humidity = sample_scalar(v->humidity, v->dims, p);
    s.emission = sample_scalar(v->emission, v->dims, p);
    s.age = fmaxf(sample_scalar(v->age, v->dims, p), 0.0f);
    return s;
}

static v3 bounds_min(const smoke_volume *v) { return (v3){v->origin[0], v->origin[1], v->origin[2]}; }
static v3 bounds_max(const smoke_volume *v) {
    return (v3){v->origin[0] + (float)v->dims[0] * v->voxel_size[0], v->origin[1] + (float)v->dims[1] * v->voxel_size[1],
                v->origin[2] + (float)v->dims[2] * v->voxel_size[2]};
}

static int ray_box_intersection(v3 o, v3 d, v3 mn, v3 mx, float *near_out, float *far_out) {
    v3 inv = {fabsf(d.x) > 1.0e-12f ? 1.0f / d.x : INFINITY, fabsf(d.y) > 1.0e-12f ? 1.0f / d.y : INFINITY,
              fabsf(d.z) > 1.0e-12f ? 1.0f / d.z : INFINITY};
    v3 t0 = v3_mul(v3_sub(mn, o), inv), t1 = v3_mul(v3_sub(mx, o), inv);
    v3 tmin = {fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z)};
    v3 tmax = {fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z)};
    float nr = fmaxf(fmaxf(tmin.x, tmin.y), tmin.z), fr = fminf(fminf(tmax.x, tmax.y), tmax.z);
    if (fr >= fmaxf(nr, 0.0f)) {
        *near_out = nr;
        *far_out = fr;
        return 1;
    }
    return 0;
}

static float henyey_greenstein(float c, float g) {
    float g2 = g * g;
    float denom = fmaxf(1.0f + g2 - 2.0f * g * c, 1.0e-4f);
    return (1.0f - g2) / (4.0f * 3.14159265358979323846f * (denom * sqrtf(denom)));
}

static float render_smoothstep(float e0, float e1, float x) {
    float t = clampf((x - e0) / fmaxf(e1 - e0, 1.0e-6f), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

static v3 mix_vec3(v3 a, v3 b, float t) { return (v3){lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t)}; }

static v3 smoke_color(render_sample s, const smoke_settings *st) {
    float body = clampf(s.density * 1.45f + s.soot * 1.35f, 0.0f, 1.0f);
    v3 thin = {st->thin_color[0], st->thin_color[1], st->thin_color[2]};
    v3 dense = {st->dense_color[0], st->dense_color[1], st->dense_color[2]};
    v3 c = mix_vec3(thin, dense, body);
    float aged = clampf(s.age / 9.0f, 0.0f, 1.0f);
    c = mix_vec3(c, (v3){0.36f, 0.39f, 0.43f}, aged * 0.42f);
    float milk = clampf(s.humidity, 0.0f, 1.0f) * (0.18f + 0.42f * body);
    c = mix_vec3(c, (v3){0.93f, 0.92f, 0.84f}, clampf(milk, 0.0f, 0.38f));
    float freshness = clampf(1.0f - s.age / 17.0f, 0.0f, 1.0f);
    float heat = clampf(s.temperature * 0.12f * freshness, 0.0f, 1.0f);
    return mix_vec3(c, (v3){0.95f, 0.62f, 0.28f}, heat * 0.07f);
}

static uint8_t to_u8(float v) { return (uint8_t)(clampf(v, 0.0f, 1.0f) * 255.0f + 0.5f); }

float smoke_oracle_sun_transmittance(const smoke_volume *v, const float start_[3], const float sun_[3], float step,
                                     uint32_t steps, const smoke_settings *st) {
    v3 start = {start_[0], start_[1], start_[2]}, sun = {sun_[0], sun_[1], sun_[2]};
    float t0, t1;
    if (!ray_box_intersection(v3_add(start, v3_scale(sun, step)), sun, bounds_min(v), bounds_max(v), &t0, &t1)) return 1.0f;
    t0 = fmaxf(t0, 0.0f);
    float od = 0.0f;
    for (uint32_t i = 0; i < steps; i++) {
        float t = t0 + ((float)i + 0.5f) * step;
        if (t > t1) break;
        render_sample s = sample_render_fields(v, v3_add(start, v3_scale(sun, step + t)));
        float age_t = render_smoothstep(1.6f, 17.0f, s.age);
        float gate = 0.50f + 0.50f * render_smoothstep(0.045f, 0.34f, s.density);
        od += s.density * st->density_scale * (1.0f - 0.58f * age_t) * gate * st->extinction *
              (1.0f + s.soot * st->soot_absorption) * step;
        if (od > 8.0f) break;
    }
    return clampf(exp_det(-od), 0.0f, 1.0f);
}

static void march_ray_rgba(const smoke_volume *v, v3 origin, v3 dir, float t0, float t1, uint32_t seed, float step,
                           float shadow_step, v3 sun, const smoke_settings *st, uint8_t out[4]) {
    float jitter = (hash01(seed) - 0.5f) * st->jitter_strength * step;
    float t = fmaxf(t0 + jitter, 0.0f), transmittance = 1.0f;
    v3 rgb = {0.0f, 0.0f, 0.0f};
    uint32_t steps = 0;
    const float sun_a[3] = {sun.x, sun.y, sun.z};
    while (t < t1 && steps < st->max_steps && transmittance > 0.01f) {
        v3 p = v3_add(origin, v3_scale(dir, t));
        render_sample s = sample_render_fields(v, p);
        float age_t = render_smoothstep(1.6f, 17.0f, s.age);
        float gate = 0.50f + 0.50f * render_smoothstep(0.045f, 0.34f, s.density);
        float density = fmaxf(s.density * st->density_scale * (1.0f - 0.58f * age_t) * gate, 0.0f);
        if (density > 1.0e-5f) {
            float sigma_t = density * st->extinction * (1.0f + s.soot * st->soot_absorption * 0.85f);
            float seg_tr = clampf(exp_det(-sigma_t * step), 0.0f, 1.0f);
            float seg_w = sigma_t > 1.0e-6f ? (1.0f - seg_tr) / sigma_t : step;
            const float pa[3] = {p.x, p.y, p.z};
            float light = st->self_shadow ? smoke_oracle_sun_transmittance(v, pa, sun_a, shadow_step, st->shadow_steps, st) : 1.0f;
            float cos_theta = clampf(v3_dot(dir, sun), -1.0f, 1.0f);
            float phase = henyey_greenstein(cos_theta, st->phase_g);
            v3 col = smoke_color(s, st);
            float albedo = clampf(st->scattering / (st->scattering + st->absorption + s.soot * 0.55f + 1.0e-5f), 0.02f, 0.98f);
            float sigma_s = sigma_t * albedo;
            v3 sun_rad = v3_scale((v3){1.0f, 0.96f, 0.84f}, 11.5f);
            v3 sky = v3_scale(v3_scale((v3){0.52f, 0.60f, 0.72f}, 0.36f + 0.26f * clampf(1.0f - light, 0.0f, 1.0f)),
                              clampf(1.0f - s.soot * 0.32f, 0.50f, 1.0f));
            v3 bounce = v3_scale(v3_scale((v3){0.58f, 0.54f, 0.48f}, 0.070f),
                                 clampf(1.0f - p.y / fmaxf(bounds_max(v).y, 1.0f), 0.0f, 1.0f));
            float powder = clampf(1.0f - exp_det(-sigma_t * step * 2.2f), 0.0f, 1.0f);
            float pw = powder * 0.055f * sqrtf(light);
            v3 multiple = v3_mul(v3_scale(col, sigma_s), v3_add(v3_add(sky, bounce), (v3){pw, pw, pw}));
            v3 direct = v3_scale(v3_scale(v3_mul(v3_scale(col, sigma_s), sun_rad), phase), light);
            float freshness = clampf(1.0f - s.age / 17.0f, 0.0f, 1.0f);
            float fresh_heat = s.temperature * freshness * freshness;
            v3 emission = v3_scale((v3){1.0f, 0.30f, 0.055f}, clampf((fresh_heat * 0.10f + s.emission * 1.18f) * st->fire_glow, 0.0f, 5.0f));
            v3 source = v3_add(v3_add(direct, multiple), emission);
            rgb = v3_add(rgb, v3_scale(v3_scale(source, seg_w), transmittance));
            transmittance *= seg_tr;
        }
        t += step;
        steps++;
    }
    float alpha = clampf(1.0f - transmittance, 0.0f, 1.0f);
    v3 straight = alpha > 1.0e-5f ? (v3){rgb.x / alpha, rgb.y / alpha, rgb.z / alpha} : rgb;
    v3 e = v3_scale(straight, st->exposure);
    out[0] = to_u8(e.x / (1.0f + e.x));
    out[1] = to_u8(e.y / (1.0f + e.y));
    out[2] = to_u8(e.z / (1.0f + e.z));
    out[3] = to_u8(alpha);
}

static int fail(char *err, size_t n, const char *msg) {
    if (err && n) snprintf(err, n, "%s", msg);
    return 1;
}

static int validate_settings(const smoke_settings *s, char *err, size_t n) { /* types.rs:271-316 */
    const char *names[11] = {"density_scale", "extinction", "scattering", "absorption", "phase_g", "step_size",
                             "shadow_step_size", "jitter_strength", "exposure", "soot_absorption", "fire_glow"};
    const float vals[11] = {s->density_scale, s->extinction, s->scattering, s->absorption, s->phase_g, s->step_size,
                            s->shadow_step_size, s->jitter_strength, s->exposure, s->soot_absorption, s->fire_glow};
    char buf[96];
    for (int i = 0; i < 11; i++)
        if (!isfinite(vals[i])) {
            snprintf(buf, sizeof buf, "%s must be finite", names[i]);
            return fail(err, n, buf);
        }
    if (s->density_scale < 0.0f || s->extinction < 0.0f || s->scattering < 0.0f)
        return fail(err, n, "density_scale, extinction, and scattering must be >= 0");
    if (s->absorption < 0.0f || s->soot_absorption < 0.0f || s->fire_glow < 0.0f)
        return fail(err, n, "absorption, soot_absorption, and fire_glow must be >= 0");
    if (!(s->phase_g >= -0.99f && s->phase_g <= 0.99f)) return fail(err, n, "phase_g must be in [-0.99, 0.99]");
    if (s->step_size < 0.0f || s->shadow_step_size < 0.0f) return fail(err, n, "step sizes must be >= 0");
    if (s->max_steps == 0 || s->shadow_steps == 0) return fail(err, n, "max_steps and shadow_steps must be >= 1");
    if (!(s->jitter_strength >= 0.0f && s->jitter_strength <= 1.0f)) return fail(err, n, "jitter_strength must be in [0, 1]");
    for (int c = 0; c < 2; c++)
        for (int a = 0; a < 3; a++) {
            float v = c ? s->dense_color[a] : s->thin_color[a];
            if (!isfinite(v) || v < 0.0f) {
                snprintf(buf, sizeof buf, "%s[%d] must be finite and >= 0", c ? "dense_color" : "thin_color", a);
                return fail(err, n, buf);
            }
        }
    return 0;
}

static void step_sizes(const smoke_volume *v, const smoke_settings *st, float *step, float *shadow_step) {
    float mn = fmaxf(fminf(fminf(fminf(INFINITY, v->voxel_size[0]), v->voxel_size[1]), v->voxel_size[2]), 1.0e-4f);
    *step = st->step_size > 0.0f ? st->step_size : mn * 0.75f;
    *shadow_step = st->shadow_step_size > 0.0f ? st->shadow_step_size : *step * 2.0f;
}

int smoke_oracle_raymarch_rgba(const smoke_volume *v, uint32_t width, uint32_t height, const float cam[3],
                               const float target_[3], const float up_[3], float fovy_deg, const float sun_[3],
                               const smoke_settings *st, uint8_t *out, char *err, size_t errlen) {
    if (validate_settings(st, err, errlen)) return 1;
    if (width == 0 || height == 0) return fail(err, errlen, "width and height must be >= 1");
    if (!isfinite(fovy_deg) || fovy_deg <= 0.0f || fovy_deg >= 179.0f)
        return fail(err, errlen, "fovy_deg must be finite and in (0, 179)");
    v3 eye = {cam[0], cam[1], cam[2]}, target = {target_[0], target_[1], target_[2]};
    v3 forward = normalize_or_zero(v3_sub(target, eye));
    if (v3_dot(forward, forward) < 1.0e-12f) return fail(err, errlen, "camera_pos and target must not be equal");
    v3 up = normalize_or_zero((v3){up_[0], up_[1], up_[2]});
    if (v3_dot(up, up) < 1.0e-12f) return fail(err, errlen, "up vector must not be zero");
    v3 right = normalize_or_zero(v3_cross(forward, up));
    v3 camera_up = normalize_or_zero(v3_cross(right, forward));
    v3 sun = normalize_or_zero((v3){sun_[0], sun_[1], sun_[2]});
    if (v3_dot(sun, sun) < 1.0e-12f) return fail(err, errlen, "sun_direction must not be zero");
    float step, shadow_step;
    step_sizes(v, st, &step, &shadow_step);
    float tan_half = tanf((fovy_deg * (3.14159265358979323846f / 180.0f)) * 0.5f);
    float aspect = (float)width / (float)height;
    v3 bmin = bounds_min(v), bmax = bounds_max(v);
    memset(out, 0, (size_t)width * height * 4);
#pragma omp parallel for schedule(dynamic, 4)
    for (long y = 0; y < (long)height; y++)
        for (uint32_t x = 0; x < width; x++) {
            float px = (((float)x + 0.5f) / (float)width * 2.0f - 1.0f) * aspect * tan_half;
            float py = (1.0f - ((float)y + 0.5f) / (float)height * 2.0f) * tan_half;
            v3 dir = normalize(v3_add(v3_add(forward, v3_scale(right, px)), v3_scale(camera_up, py)));
            float t0, t1;
            if (!ray_box_intersection(eye, dir, bmin, bmax, &t0, &t1)) continue;
            t0 = fmaxf(t0, 0.0f);
            uint32_t seed = x * 73856093u + (uint32_t)y * 19349663u + v->frame_index;
            march_ray_rgba(v, eye, dir, t0, t1, seed, step, shadow_step, sun, st, out + ((size_t)y * width + x) * 4);
        }
    return 0;
}

int smoke_oracle_raymarch_projection_rgba(const smoke_volume *v, uint32_t width, uint32_t height, const float view_[3],
                                          const float sun_[3], const smoke_settings *st, uint8_t *out, char *err,
                                          size_t errlen) {
    if (validate_settings(st, err, errlen)) return 1;
    if (width == 0 || height == 0) return fail(err, errlen, "width and height must be >= 1");
    v3 dir = normalize_or_zero((v3){view_[0], view_[1], view_[2]});
    if (v3_dot(dir, dir) < 1.0e-12f) return fail(err, errlen, "view_direction must not be zero");
    v3 sun = normalize_or_zero((v3){sun_[0], sun_[1], sun_[2]});
    if (v3_dot(sun, sun) < 1.0e-12f) return fail(err, errlen, "sun_direction must not be zero");
    float step, shadow_step;
    step_sizes(v, st, &step, &shadow_step);
    v3 bmin = bounds_min(v), bmax = bounds_max(v);
    v3 ext = v3_sub(bmax, bmin);
    float diagonal = fmaxf(sqrtf(v3_dot(ext, ext)), step * 2.0f);
    memset(out, 0, (size_t)width * height * 4);
#pragma omp parallel for schedule(dynamic, 4)
    for (long py = 0; py < (long)height; py++) {
        float fz = ((float)py + 0.5f) / (float)height;
        float z = lerpf(bmin.z, bmax.z, fz);
        for (uint32_t px = 0; px < width; px++) {
            float fx = ((float)px + 0.5f) / (float)width;
            float x = lerpf(bmin.x, bmax.x, fx);
            v3 plane = {x, (bmin.y + bmax.y) * 0.5f, z};
            v3 origin = v3_sub(plane, v3_scale(dir, diagonal));
            float t0, t1;
            if (!ray_box_intersection(origin, dir, bmin, bmax, &t0, &t1)) continue;
            t0 = fmaxf(t0, 0.0f);
            uint32_t seed = px * 73856093u + (uint32_t)py * 19349663u + v->frame_index + 0x9e3779b9u;
            march_ray_rgba(v, origin, dir, t0, t1, seed, step, shadow_step, sun, st, out + ((size_t)py * width + px) * 4);
        }
    }
    return 0;
}

/* ---- hooks onto the marcher's primitives (tests/test_smoke.py: pinned against vectors written by the REFERENCE's own NumPy
 * helpers, python/forge3d/smoke.py:734-941, through tests/golden/make_smoke_vectors.py).  The NumPy helpers are the
 * reference's second, independent statement of the same primitives (its example renderer), so a misreading of
 * src/smoke/render.rs shared by this file and the HIP kernel cannot hide behind them. ---- */
void smoke_oracle_hook_sample_scalar(const float *field, const uint32_t dims[3], const float *points, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; i++) out[i] = sample_scalar(field, dims, points + 3u * i); /* voxel-index coordinates (x, y, z) */
}
void smoke_oracle_hook_ray_box(const float *origins, const float dir[3], const float mn[3], const float mx[3], uint32_t n,
                               float *near_out, float *far_out, int32_t *valid) {
    for (uint32_t i = 0; i < n; i++) {
        float a = 0.0f, b = 0.0f;
        valid[i] = ray_box_intersection((v3){origins[3u * i], origins[3u * i + 1u], origins[3u * i + 2u]}, (v3){dir[0], dir[1], dir[2]},
                                        (v3){mn[0], mn[1], mn[2]}, (v3){mx[0], mx[1], mx[2]}, &a, &b);
        near_out[i] = a;
        far_out[i] = b;
    }
}
float smoke_oracle_hook_henyey_greenstein(float cos_theta, float g) { return henyey_greenstein(cos_theta, g); }
float smoke_oracle_hook_smoothstep(float e0, float e1, float x) { return render_smoothstep(e0, e1, x); }
float smoke_oracle_hook_exp(float x) { return exp_det(x); }
