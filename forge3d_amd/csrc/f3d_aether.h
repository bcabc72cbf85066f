// forge3d_amd/csrc/f3d_aether.h -- AETHER aerial-perspective post of the terrain path tracer.
//
// Reference: src/shaders/atmosphere/prometheus_aerial.wgsl:99-231 (the post kernel) with the LUT evaluation core
// src/shaders/atmosphere/evaluation_core.wgsl:29-344 and the host side src/path_tracing/hybrid_compute/
// aether_post.rs:42-186 (uniform block).  One thread per pixel at resolve time:
//   L_out = L_surface * T_segment + max(S(camera) - T_segment * S(endpoint), 0)        (hits)
//   L_out = S(camera)                                                                  (sky)
// with S = the accumulated-scattering table (quadrilinear, nonlinear mu / nu / height coordinates), T_segment =
// a 16-sample spectral (11 wavelengths -> CIE -> RGB) segment transmittance anchored to the aerial froxel's
// mean, bounded below by the boundary transmittance table; then the unchanged Reinhard resolve.
// The three tables are plain float4 arrays here (decoded from RGBA16F once on the host); a pixel reads
// 2 x 16 + 2 texels.  e^x is f3d_math.h exp_det (the reference's det_exp is exp2(x log2 e) on the driver).
#pragma once

#include "f3d_math.h"

namespace f3d {

struct AetherDev {
    const float4 *transmittance;  // [height][mu]            dims t_mu x t_h
    const float4 *scattering;     // [h * nu + n][sun][view]   dims s_view x s_sun x (s_h * s_nu)
    const float4 *aerial;         // [height][mu][distance]  dims a_dist x a_mu x a_h
    uint32_t t_mu, t_h, s_view, s_sun, s_h, s_nu, a_dist, a_mu, a_h;
    float bottom_radius, top_radius, max_aerial_distance, ozone_du, turbidity;
    float sun_intensity, exposure;  // both clamped to [0, 65504] (aether_eval_clamp_radiometric_scale)
    uint32_t enabled;
};

F3D_HD float aether_clamp_scale(float v) { return f_min(f_max(v, 0.0f), 65504.0f); }
F3D_HD V3 aether_clamp_hdr(V3 c) {
    return V3{f_min(f_max(c.x, 0.0f), 65504.0f), f_min(f_max(c.y, 0.0f), 65504.0f), f_min(f_max(c.z, 0.0f), 65504.0f)};
}
F3D_HD int aether_round_index(float unit, uint32_t n) {  // i32(round(unit * f32(max(n, 1) - 1))), round = ties to even
    return (int)f_rint(unit * (float)((n > 1u ? n : 1u) - 1u));
}

// aether_eval_mu_to_unit / nu_to_unit / scattering_height_to_unit, evaluation_core.wgsl:82-100
F3D_HD float aether_mu_to_unit(float mu) {
    const float b = f_clamp(mu, -1.0f, 1.0f), m = f_sqrt(f_abs(b));
    return 0.5f * ((b >= 0.0f ? m : -m) + 1.0f);
}
F3D_HD float aether_nu_to_unit(float nu) { return 1.0f - f_sqrt(f_max(0.5f * (1.0f - f_clamp(nu, -1.0f, 1.0f)), 0.0f)); }

// aether_eval_sample_accumulated_scattering, evaluation_core.wgsl:119-177
F3D_HD V3 aether_scattering(const AetherDev &A, float height_unit, float mu_sun, float mu_view, float nu) {
    const int hc = (int)A.s_h > 2 ? (int)A.s_h : 2, nc = (int)A.s_nu > 2 ? (int)A.s_nu : 2;
    const float c[4] = {aether_mu_to_unit(mu_view) * (float)(A.s_view - 1u), aether_mu_to_unit(mu_sun) * (float)(A.s_sun - 1u),
                        f_sqrt(f_clamp(height_unit, 0.0f, 1.0f)) * (float)(hc - 1), aether_nu_to_unit(nu) * (float)(nc - 1)};
    const int hi_lim[4] = {(int)A.s_view - 1, (int)A.s_sun - 1, hc - 1, nc - 1};
    int lo[4], hi[4];
    float fr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float fl = f_floor(c[k]);
        lo[k] = (int)fl;
        hi[k] = lo[k] + 1 < hi_lim[k] ? lo[k] + 1 : hi_lim[k];
        fr[k] = c[k] - fl;  // fract
    }
    const int depth = (int)(A.s_h * A.s_nu);
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (int hs = 0; hs < 2; hs++)
        for (int ns = 0; ns < 2; ns++)
            for (int ss = 0; ss < 2; ss++)
                for (int vs = 0; vs < 2; vs++) {
                    const int vi = vs ? hi[0] : lo[0], si = ss ? hi[1] : lo[1], hi_ = hs ? hi[2] : lo[2], ni = ns ? hi[3] : lo[3];
                    const float w = (vs ? fr[0] : 1.0f - fr[0]) * (ss ? fr[1] : 1.0f - fr[1]) * (hs ? fr[2] : 1.0f - fr[2]) *
                                    (ns ? fr[3] : 1.0f - fr[3]);
                    int x = vi, y = si, z = hi_ * nc + ni;  // aether_eval_load_scattering_texel: clamped coordinate
                    x = x < 0 ? 0 : (x > (int)A.s_view - 1 ? (int)A.s_view - 1 : x);
                    y = y < 0 ? 0 : (y > (int)A.s_sun - 1 ? (int)A.s_sun - 1 : y);
                    z = z < 0 ? 0 : (z > depth - 1 ? depth - 1 : z);
                    const float4 t = A.scattering[((size_t)z * A.s_sun + (size_t)y) * A.s_view + (size_t)x];
                    ax = ax + w * t.x;
                    ay = ay + w * t.y;
                    az = az + w * t.z;
                }
    return V3{f_max(ax, 0.0f), f_max(ay, 0.0f), f_max(az, 0.0f)};
}

// aether_eval_spherical_radius_m / altitude / endpoint_mus, evaluation_core.wgsl:179-236
F3D_HD float aether_radius(float cam_h, float view_mu, float dist, float bottom) {
    const float r = f_max(bottom, 1.0f) + f_clamp(cam_h, 0.0f, 100000.0f);
    const float d = f_clamp(dist, 0.0f, 20000000.0f);
    return f_sqrt(f_max(r * r + d * d + 2.0f * r * d * f_clamp(view_mu, -1.0f, 1.0f), 0.0f));
}
F3D_HD float aether_altitude(float cam_h, float view_mu, float dist, float bottom) {
    return f_clamp(aether_radius(cam_h, view_mu, dist, bottom) - f_max(bottom, 1.0f), 0.0f, 100000.0f);
}

constexpr float kAetherWavelengths[11] = {380.0f, 420.0f, 460.0f, 500.0f, 540.0f, 580.0f, 620.0f, 660.0f, 700.0f, 740.0f, 780.0f};
constexpr float kAetherCie[11][3] = {{0.001368f, 0.000039f, 0.006450f}, {0.134380f, 0.004000f, 0.645600f}, {0.290800f, 0.060000f, 1.669200f},
                                     {0.004900f, 0.323000f, 0.272000f}, {0.290400f, 0.954000f, 0.020300f}, {0.916300f, 0.870000f, 0.001650f},
                                     {0.854450f, 0.381000f, 0.000190f}, {0.164900f, 0.061000f, 0.000000f}, {0.011359f, 0.004102f, 0.000000f},
                                     {0.000690f, 0.000249f, 0.000000f}, {0.000042f, 0.000015f, 0.000000f}};

// aether_eval_segment_transmittance (16 explicit midpoint samples; spectral -> XYZ -> RGB), evaluation_core.wgsl:238-344
F3D_HD V3 aether_segment_transmittance(float dist, float cam_h, float view_mu, float bottom, float density_scale, float turbidity,
                                       float ozone_du) {
    const float d = f_clamp(dist, 0.0f, 20000000.0f), ch = f_clamp(cam_h, 0.0f, 100000.0f);
    float ray = 0.0f, mie = 0.0f, ozo = 0.0f;
    float hs[16];
#pragma unroll
    for (int i = 0; i < 16; i++) hs[i] = aether_altitude(ch, view_mu, d * ((float)(2 * i + 1) * 0.03125f), bottom);
#pragma unroll
    for (int i = 0; i < 16; i++) ray = i == 0 ? exp_det(-hs[0] / 8000.0f) : ray + exp_det(-hs[i] / 8000.0f);
#pragma unroll
    for (int i = 0; i < 16; i++) mie = i == 0 ? exp_det(-hs[0] / 1200.0f) : mie + exp_det(-hs[i] / 1200.0f);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float o = f_max(1.0f - f_abs((hs[i] - 25000.0f) / 15000.0f), 0.0f);
        ozo = i == 0 ? o : ozo + o;
    }
    const float per = d * density_scale * 0.0625f;
    const float ray_col = per * ray, mie_col = per * mie, ozo_col = per * ozo * ozone_du / 300.0f;
    float X = 0.0f, Y = 0.0f, Z = 0.0f;
#pragma unroll
    for (int w = 0; w < 11; w++) {  // aether_eval_spectral_xyz, evaluation_core.wgsl:50-80
        const float ratio = 550.0f / kAetherWavelengths[w], r2 = ratio * ratio;
        const float ray_beta = 1.2989e-5f * r2 * r2, mie_beta = 1.0e-5f * turbidity * ratio;
        const float od = (kAetherWavelengths[w] - 600.0f) / 85.0f;
        const float ozo_beta = 1.2e-6f * exp_det(-0.5f * od * od);
        const float t = exp_det(-f_max(ray_beta * ray_col + mie_beta * mie_col + ozo_beta * ozo_col, 0.0f));
        const float ew = (w == 0 || w == 10) ? 0.5f : 1.0f;
        const float cx = kAetherCie[w][0] * t * ew, cy = kAetherCie[w][1] * t * ew, cz = kAetherCie[w][2] * t * ew;
        X = w == 0 ? cx : X + cx;
        Y = w == 0 ? cy : Y + cy;
        Z = w == 0 ? cz : Z + cz;
    }
    const V3 xyz = V3{X, Y, Z};  // aether_eval_xyz_to_rgb, evaluation_core.wgsl:42-48
    const V3 rgb = V3{dot(V3{3.2404542f, -1.5371385f, -0.4985314f}, xyz) / 3.2613921f,
                      dot(V3{-0.9692660f, 1.8760108f, 0.0415560f}, xyz) / 2.5069624f,
                      dot(V3{0.0556434f, -0.2040259f, 1.0572252f}, xyz) / 2.3679786f};
    return V3{f_clamp(rgb.x, 0.0f, 1.0f), f_clamp(rgb.y, 0.0f, 1.0f), f_clamp(rgb.z, 0.0f, 1.0f)};
}

// prometheus_aerial.wgsl main (:99-231): linear accumulation mean, frame-0 depth + visibility -> Reinhard-mapped LDR
// (before the RGBA16F store).  `ray` = the unjittered pixel ray, `cam_y` = camera height above the datum.
F3D_HD V3 aether_resolve(const AetherDev &A, V3 mean_radiance, float depth, bool visible, V3 ray, V3 sun_dir, float cam_y) {
    const V3 surface = aether_clamp_hdr(mean_radiance);
    const float atmosphere_height = f_max(A.top_radius - A.bottom_radius, 1.0f);
    const float cam_h = f_max(cam_y, 0.0f), cam_unit = f_clamp(cam_h / atmosphere_height, 0.0f, 1.0f);
    const float nu = dot(ray, sun_dir);
    V3 hdr;
    if (!visible) {
        hdr = aether_clamp_hdr(aether_scattering(A, cam_unit, sun_dir.y, ray.y, nu) * A.sun_intensity);
    } else {
        const float end_h = aether_altitude(cam_h, ray.y, depth, A.bottom_radius);
        // aether_eval_spherical_endpoint_mus
        const float r = f_max(A.bottom_radius, 1.0f) + f_clamp(cam_h, 0.0f, 100000.0f), bd = f_clamp(depth, 0.0f, 20000000.0f);
        const float end_r = f_max(aether_radius(cam_h, ray.y, bd, A.bottom_radius), 1.0f);
        const float end_view_mu = f_clamp((r * f_clamp(ray.y, -1.0f, 1.0f) + bd) / end_r, -1.0f, 1.0f);
        const float end_sun_mu = f_clamp((r * f_clamp(sun_dir.y, -1.0f, 1.0f) + bd * f_clamp(nu, -1.0f, 1.0f)) / end_r, -1.0f, 1.0f);
        const V3 seg = aether_segment_transmittance(depth, cam_h, ray.y, A.bottom_radius, 1.0f, A.turbidity, A.ozone_du);
        // prometheus_load_boundary_transmittance (linear mu axis)
        const int bx = aether_round_index(0.5f * (f_clamp(ray.y, -1.0f, 1.0f) + 1.0f), A.t_mu), by = aether_round_index(f_clamp(cam_unit, 0.0f, 1.0f), A.t_h);
        const float4 bt4 = A.transmittance[(size_t)by * A.t_mu + (size_t)bx];
        const V3 boundary = V3{f_clamp(bt4.x, 0.0f, 1.0f), f_clamp(bt4.y, 0.0f, 1.0f), f_clamp(bt4.z, 0.0f, 1.0f)};
        const V3 cam_s = aether_scattering(A, cam_unit, sun_dir.y, ray.y, nu) * A.sun_intensity;
        const float end_unit = f_clamp(end_h / atmosphere_height, 0.0f, 1.0f);
        const V3 end_s = aether_scattering(A, end_unit, end_sun_mu, end_view_mu, nu) * A.sun_intensity;
        // prometheus_load_aerial_transmittance (alpha channel only)
        const float dist_unit = depth / f_max(A.max_aerial_distance, 1.0f);
        const int ax = aether_round_index(f_clamp(dist_unit, 0.0f, 1.0f), A.a_dist),
                  ay = aether_round_index(0.5f * (f_clamp(ray.y, -1.0f, 1.0f) + 1.0f), A.a_mu), az = aether_round_index(f_clamp(cam_unit, 0.0f, 1.0f), A.a_h);
        const float aerial_t = f_clamp(A.aerial[((size_t)az * A.a_mu + (size_t)ay) * A.a_dist + (size_t)ax].w, 0.0f, 1.0f);
        const float mean_t = dot(seg, V3{0.2126f, 0.7152f, 0.0722f});
        const float k = aerial_t / f_max(mean_t, 1.0e-6f);
        const V3 tr = V3{f_max(f_clamp(seg.x * k, 0.0f, 1.0f), boundary.x), f_max(f_clamp(seg.y * k, 0.0f, 1.0f), boundary.y),
                         f_max(f_clamp(seg.z * k, 0.0f, 1.0f), boundary.z)};
        const V3 ins = V3{f_max(cam_s.x - tr.x * end_s.x, 0.0f), f_max(cam_s.y - tr.y * end_s.y, 0.0f), f_max(cam_s.z - tr.z * end_s.z, 0.0f)};
        hdr = aether_clamp_hdr(V3{surface.x * tr.x + ins.x, surface.y * tr.y + ins.y, surface.z * tr.z + ins.z});
    }
    const V3 e = hdr * A.exposure;  // tonemap_reinhard, tonemap_common.wgsl:18-21
    return V3{e.x / (1.0f + e.x), e.y / (1.0f + e.y), e.z / (1.0f + e.z)};
}

}  // namespace f3d
