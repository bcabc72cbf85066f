// forge3d_amd/csrc/f3d_aether_ref_host.h -- host side of the AETHER acceptance reference shared by the library
// (f3d_aether_ref.hip) and the host emulator (tests/emul): validation, uniforms, finalisation
// (reference src/path_tracing/hybrid_compute/aether_reference.rs).
#pragma once

#include <algorithm>
#include <cmath>

#include "../../include/f3d_terrain_pt.h"
#include "f3d_aether_ref.h"
#include "f3d_setup.h"

namespace f3d {
namespace aref {

// validate_desc, aether_reference.rs:83-153 (same order, same messages)
inline void validate_ref_desc(const f3d_aether_ref_desc &d) {
    auto invalid = [](const char *m) { fail(F3D_STATUS_RENDER, "%s", m); };
    if (d.width == 0u || d.height == 0u) invalid("AETHER spectral reference requires non-zero width and height");
    if (d.spp == 0u || d.spp > 4096u) invalid("AETHER spectral reference spp must be in 1..=4096");
    const unsigned long long paths = (unsigned long long)d.width * d.height * d.spp * kWavelengths;
    if (paths > 8000000ull)
        fail(F3D_STATUS_RENDER, "AETHER spectral reference request has %llu wavelength paths; acceptance lane limit is 8000000", paths);
    if (!(std::isfinite(d.spacing_x) && d.spacing_x > 0.0f && std::isfinite(d.spacing_z) && d.spacing_z > 0.0f))
        invalid("AETHER spectral reference spacing must be finite and positive");
    if (!(std::isfinite(d.exaggeration) && d.exaggeration > 0.0f)) invalid("AETHER spectral reference exaggeration must be finite and positive");
    if (!(finite3(d.cam_origin) && finite3(d.cam_look_at) && finite3(d.cam_up))) invalid("AETHER spectral reference camera vectors must be finite");
    const V3 origin{d.cam_origin[0], d.cam_origin[1], d.cam_origin[2]};
    const V3 forward = V3{d.cam_look_at[0], d.cam_look_at[1], d.cam_look_at[2]} - origin;
    if (length3(forward) < 1e-6f || length3(cross(normalize(forward), V3{d.cam_up[0], d.cam_up[1], d.cam_up[2]})) < 1e-6f)
        invalid("AETHER spectral reference camera basis is degenerate");
    const float observer_altitude = length3(origin - planet_center()) - kBottomRadius;
    if (!(observer_altitude >= 0.0f && observer_altitude < 100000.0f)) invalid("AETHER spectral reference camera must be inside the 0..100 km atmosphere");
    if (!(std::isfinite(d.fov_y_deg) && d.fov_y_deg > 0.0f && d.fov_y_deg < 180.0f)) invalid("AETHER spectral reference fov_y_deg must be in (0, 180)");
    if (!(std::isfinite(d.sun_azimuth_deg) && std::isfinite(d.sun_elevation_deg) && std::isfinite(d.sun_intensity) && d.sun_intensity >= 0.0f))
        invalid("AETHER spectral reference sun inputs must be finite and intensity non-negative");
    // AtmosphereConfig::validate, core/atmosphere/bake.rs:178-205
    const char *problem = nullptr;
    if (!(d.turbidity >= 1.0f && d.turbidity <= 10.0f)) problem = "turbidity must be in [1, 10]";
    else if (!(d.ozone_du >= 0.0f && d.ozone_du <= 600.0f)) problem = "ozone must be in [0, 600] DU";
    else if (!(d.mie_g >= 0.0f && d.mie_g <= 0.99f)) problem = "mie_g must be in [0, 0.99]";
    else if (!(d.ground_albedo >= 0.0f && d.ground_albedo <= 1.0f)) problem = "ground_albedo must be in [0, 1]";
    if (problem) fail(F3D_STATUS_RENDER, "invalid canonical AETHER settings for spectral reference: %s", problem);
    if (!(std::isfinite(d.variance_threshold) && d.variance_threshold > 0.0f))
        invalid("AETHER spectral reference variance_threshold must be finite and positive");
}

// The DEM itself: finite, at least 2 x 2, and wholly inside the top sphere -- |p - c| <= hypot(R + y_max, half diagonal)
// < R_top -- so that a terrain hit of a ray from inside the atmosphere always precedes its top-of-atmosphere exit (what
// intersect_shadow_ray(ray, top_t) compares, prometheus_spectral_reference.wgsl:172).
inline void check_ref_terrain(const f3d_aether_ref_desc &d) {
    if (!d.heights || d.dem_width < 2u || d.dem_height < 2u) fail(F3D_STATUS_UPLOAD, "terrain heightfield must be at least 2x2 texels");
    float h_max = -INFINITY;
    for (size_t i = 0; i < (size_t)d.dem_width * d.dem_height; i++) {
        if (!std::isfinite(d.heights[i])) fail(F3D_STATUS_UPLOAD, "terrain heightfield contains non-finite samples");
        h_max = std::max(h_max, d.heights[i]);
    }
    const double hx = 0.5 * (d.dem_width - 1.0) * d.spacing_x, hz = 0.5 * (d.dem_height - 1.0) * d.spacing_z;
    const double y = std::max(0.0, (double)h_max * d.exaggeration), r = (double)kBottomRadius + y;
    if (!(std::sqrt(r * r + hx * hx + hz * hz) < (double)kTopRadius))
        fail(F3D_STATUS_RENDER, "AETHER spectral reference terrain must lie inside the 100 km atmosphere");
}

// Everything of the scene but the table pointers: placement, camera, sun, seeds (aether_reference.rs:220-285)
inline void fill_ref_scene(const f3d_aether_ref_desc &d, RefScene &S) {
    S.terrain.origin_x = -0.5f * ((float)d.dem_width - 1.0f) * d.spacing_x;  // terrain_heightfield.rs:359-360
    S.terrain.origin_z = -0.5f * ((float)d.dem_height - 1.0f) * d.spacing_z;
    S.terrain.spacing_x = d.spacing_x;
    S.terrain.spacing_z = d.spacing_z;
    S.terrain.inv_spacing_x = 1.0f / d.spacing_x;
    S.terrain.inv_spacing_z = 1.0f / d.spacing_z;
    S.terrain.inv_two_r_prime = 0.0f;
    S.terrain.curvature_enabled = 0u;
    S.terrain.horizon = nullptr;
    const float kDegF = 0.017453292519943295f;
    const V3 origin{d.cam_origin[0], d.cam_origin[1], d.cam_origin[2]};
    const V3 forward = normalize(V3{d.cam_look_at[0], d.cam_look_at[1], d.cam_look_at[2]} - origin);
    const V3 right = normalize(cross(forward, V3{d.cam_up[0], d.cam_up[1], d.cam_up[2]}));
    const V3 up = normalize(cross(right, forward));
    const float az = d.sun_azimuth_deg * kDegF, el = d.sun_elevation_deg * kDegF;
    S.cam.origin = origin;
    S.cam.right = right;
    S.cam.up = up;
    S.cam.forward = forward;
    S.cam.half_h = tanf(0.5f * (d.fov_y_deg * kDegF));
    S.cam.half_w = ((float)d.width / (float)d.height) * S.cam.half_h;
    S.cam.exposure = 1.0f;
    S.cam.width = d.width;
    S.cam.height = d.height;
    S.cam.seed_hi = d.seed;
    S.cam.seed_lo = ((d.seed << 16) | (d.seed >> 16)) ^ 0x85EBCA6Bu;  // rotate_left(16): a plain xor would cancel against seed_hi
    S.sun_direction = normalize(V3{cosf(az) * cosf(el), sinf(el), sinf(az) * cosf(el)});
    S.sun_radiance = dot(V3{d.sun_intensity, d.sun_intensity, d.sun_intensity}, V3{0.2126f, 0.7152f, 0.0722f});
    S.turbidity = d.turbidity;
    S.mie_g = d.mie_g;
    S.ozone_scale = d.ozone_du / 300.0f;
    S.ground_albedo = d.ground_albedo;
    S.spp = d.spp;
    S.frame_index = 0u;
}

// aether_xyz_to_signed_linear_rgb, aether_reference.rs:66-72
inline void xyz_to_signed_rgb(const float *xyz, float *rgb) {
    rgb[0] = (3.2404542f * xyz[0] - 1.5371385f * xyz[1] - 0.4985314f * xyz[2]) / 3.2613921f;
    rgb[1] = (-0.9692660f * xyz[0] + 1.8760108f * xyz[1] + 0.0415560f * xyz[2]) / 2.5069624f;
    rgb[2] = (0.0556434f * xyz[0] - 0.2040259f * xyz[1] + 1.0572252f * xyz[2]) / 2.3679786f;
}

// accum: 4 floats per pixel (sum_xyz, primary hits); welford: 2 per pixel (mean_y, m2_y); aether_reference.rs:500-548
inline void finalize_ref(const f3d_aether_ref_desc &d, const float *accum, const float *welford, f3d_aether_ref_out &out) {
    const size_t pixels = (size_t)d.width * d.height;
    for (size_t i = 0; i < 4 * pixels; i++)
        if (!std::isfinite(accum[i])) fail(F3D_STATUS_RENDER, "AETHER spectral reference produced non-finite transport output");
    for (size_t i = 0; i < 2 * pixels; i++)
        if (!std::isfinite(welford[i])) fail(F3D_STATUS_RENDER, "AETHER spectral reference produced non-finite transport output");
    const float inverse_count = 1.0f / (float)d.spp;
    uint64_t primary_hits = 0;
    float variance = 0.0f;
    for (size_t p = 0; p < pixels; p++) {
        const float mean[3] = {accum[4 * p] * inverse_count, accum[4 * p + 1] * inverse_count, accum[4 * p + 2] * inverse_count};
        float rgb[3];
        xyz_to_signed_rgb(mean, rgb);
        for (int c = 0; c < 3; c++) {
            out.mean_xyz[3 * p + c] = mean[c];
            out.linear_rgb[3 * p + c] = std::max(rgb[c], 0.0f);
        }
        primary_hits += (uint64_t)std::max(std::round(accum[4 * p + 3]), 0.0f);
        if (d.spp > 1u) variance = std::max(variance, welford[2 * p + 1] / ((float)d.spp * (float)(d.spp - 1u)));
    }
    out.variance = variance;
    out.converged = (d.spp > 1u && variance <= d.variance_threshold) ? 1 : 0;
    out.terrain_primary_hits = primary_hits;
}

}  // namespace aref
}  // namespace f3d
