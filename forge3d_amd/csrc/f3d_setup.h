// forge3d_amd/csrc/f3d_setup.h -- HIP-free host logic of the terrain path tracer:
// trust-boundary validation, WGS84/refraction model, table layout and the uniform block.
// Included by f3d_host.hip (the product) and by the CPU emulation harness under tests/emul
// (test infrastructure that runs the kernel code on the host to debug without a GPU).
#pragma once

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/f3d_terrain_pt.h"
#include "f3d_build.h"
#include "f3d_bvh.h"
#include "f3d_scene.h"

namespace f3d {

struct Failure {
    int status;
    std::string message;
};

[[noreturn]] inline void fail(int status, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Failure{status, buf};
}

inline int report(const Failure &f, char *err, size_t errlen) {
    if (err && errlen) snprintf(err, errlen, "%s", f.message.c_str());
    return f.status;
}

inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
inline uint32_t pad8(uint32_t v) { return v < 8u ? 8u : ((v + 7u) & ~7u); }

inline bool finite3(const float *v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

// Rust's {:.Ne} formatting (no exponent padding) for the non-convergence message.
inline std::string rust_exp(double v, int prec) {
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    if (std::isnan(v)) return "NaN";
    char tmp[64];
    snprintf(tmp, sizeof(tmp), "%.*e", prec, v);
    char *e = strchr(tmp, 'e');
    int ex = atoi(e + 1);
    *e = 0;
    return std::string(tmp) + "e" + std::to_string(ex);
}

// ---- geo::refraction (reference src/geo/refraction.rs:6-13, :57-77, :100-148, :160-165) ----
constexpr double kWgs84A = 6378137.0;
constexpr double kWgs84E2 = 6.6943799901413165e-3;
constexpr double kDeg = 0.017453292519943295;

inline double effective_radius(int earth, double lat, double sphere_r, int refr, double pressure, double temp, double k_in,
                        double azimuth_deg) {
    if (earth == F3D_EARTH_FLAT && refr != F3D_REFRACTION_NONE)
        fail(F3D_STATUS_RENDER, "flat earth only supports refraction_model='none'");
    if (!std::isfinite(azimuth_deg)) fail(F3D_STATUS_RENDER, "azimuth must be finite");
    double radius;
    switch (earth) {
        case F3D_EARTH_FLAT: radius = INFINITY; break;
        case F3D_EARTH_SPHERE:
            if (!(std::isfinite(sphere_r) && sphere_r > 0.0))
                fail(F3D_STATUS_RENDER, "sphere radius must be finite and positive");
            radius = sphere_r;
            break;
        case F3D_EARTH_ELLIPSOID: {
            if (!(std::isfinite(lat) && lat >= -90.0 && lat <= 90.0))
                fail(F3D_STATUS_RENDER, "latitude must be finite and in [-90, 90]");
            const double sp = std::sin(lat * kDeg);
            const double w = std::sqrt(1.0 - kWgs84E2 * (sp * sp));
            const double meridional = kWgs84A * (1.0 - kWgs84E2) / (w * w * w);
            const double prime_vertical = kWgs84A / w;
            const double az = azimuth_deg * kDeg;
            const double ca = std::cos(az), sa = std::sin(az);
            radius = 1.0 / ((ca * ca) / meridional + (sa * sa) / prime_vertical);
            break;
        }
        default: fail(F3D_STATUS_VALUE, "unsupported earth_model %d", earth);
    }
    double k;
    switch (refr) {
        case F3D_REFRACTION_NONE: k = 0.0; break;
        case F3D_REFRACTION_EFFECTIVE_RADIUS: k = k_in; break;
        case F3D_REFRACTION_BENNETT:
        case F3D_REFRACTION_SAEMUNDSSON: {
            const double base = refr == F3D_REFRACTION_BENNETT ? 0.13 : 1.0 / 7.0;
            if (!std::isfinite(pressure) || pressure <= 0.0 || temp <= -273.15)
                fail(F3D_STATUS_RENDER, "pressure must be positive and temperature above absolute zero");
            k = base * (pressure / 1013.25) * (288.15 / (273.15 + temp));
            break;
        }
        default: fail(F3D_STATUS_VALUE, "unsupported refraction_model %d", refr);
    }
    if (!(std::isfinite(k) && k < 1.0)) fail(F3D_STATUS_RENDER, "refraction k must be finite and less than 1");
    return radius / (1.0 - k);
}

// ---- trust-boundary validation (render_terrain.rs:474-557) ----
inline void validate_desc(const f3d_terrain_ref_desc &d) {
    if (d.width == 0 || d.height == 0 || d.max_frames == 0)
        fail(F3D_STATUS_RENDER, "terrain reference requires non-zero width/height/max_frames");
    if (d.min_frames > d.max_frames)
        fail(F3D_STATUS_RENDER, "min_frames (%u) must be <= max_frames (%u)", d.min_frames, d.max_frames);
    if (d.spp == 0 || d.spp > 64) fail(F3D_STATUS_RENDER, "spp must be in 1..=64, got %u", d.spp);
    if (!(std::isfinite(d.exaggeration) && d.exaggeration > 0.0f))
        fail(F3D_STATUS_RENDER, "terrain exaggeration must be finite and > 0");
    if (!(finite3(d.cam_origin) && finite3(d.cam_look_at) && finite3(d.cam_up)))
        fail(F3D_STATUS_RENDER, "camera origin/look_at/up must be finite");
    const V3 origin{d.cam_origin[0], d.cam_origin[1], d.cam_origin[2]};
    const V3 fwd = V3{d.cam_look_at[0], d.cam_look_at[1], d.cam_look_at[2]} - origin;
    if (f_sqrt(dot(fwd, fwd)) < 1e-6f) fail(F3D_STATUS_RENDER, "camera look_at must differ from origin");
    const V3 c = cross(normalize(fwd), V3{d.cam_up[0], d.cam_up[1], d.cam_up[2]});
    if (f_sqrt(dot(c, c)) < 1e-6f)
        fail(F3D_STATUS_RENDER, "camera up vector must not be parallel to the view direction");
    if (!(std::isfinite(d.fov_y_deg) && d.fov_y_deg > 0.0f && d.fov_y_deg < 180.0f))
        fail(F3D_STATUS_RENDER, "fov_y must be finite and in (0, 180) degrees, got %g", (double)d.fov_y_deg);
    if (!(std::isfinite(d.exposure) && d.exposure > 0.0f)) fail(F3D_STATUS_RENDER, "exposure must be finite and > 0");
    if (!(std::isfinite(d.sun_azimuth_deg) && std::isfinite(d.sun_elevation_deg)))
        fail(F3D_STATUS_RENDER, "sun azimuth/elevation must be finite");
    if (!(std::isfinite(d.sun_intensity) && d.sun_intensity >= 0.0f))
        fail(F3D_STATUS_RENDER, "sun intensity must be finite and >= 0");
    if (!finite3(d.sun_color) || d.sun_color[0] < 0.0f || d.sun_color[1] < 0.0f || d.sun_color[2] < 0.0f)
        fail(F3D_STATUS_RENDER, "sun color must have three finite non-negative components");
    if (!(std::isfinite(d.env_intensity) && d.env_intensity >= 0.0f))
        fail(F3D_STATUS_RENDER, "env intensity must be finite and >= 0");
    if (!(std::isfinite(d.variance_threshold) && d.variance_threshold > 0.0f))
        fail(F3D_STATUS_RENDER, "variance threshold must be finite and > 0");
    if (!(std::isfinite(d.spacing_x) && d.spacing_x > 0.0f && std::isfinite(d.spacing_z) && d.spacing_z > 0.0f))
        fail(F3D_STATUS_RENDER, "terrain spacing must be finite and > 0, got (%g, %g)", (double)d.spacing_x,
             (double)d.spacing_z);
    if (d.mesh_vertices || d.mesh_indices) {
        if (!d.mesh_vertices || d.mesh_vertex_count == 0)
            fail(F3D_STATUS_RENDER, "mesh vertices must be a non-empty flat [x,y,z] list");
        if (!d.mesh_indices || d.mesh_index_count == 0 || d.mesh_index_count % 3 != 0)
            fail(F3D_STATUS_RENDER, "mesh indices must be a non-empty multiple of 3");
        for (size_t i = 0; i < (size_t)d.mesh_vertex_count * 3; i++)
            if (!std::isfinite(d.mesh_vertices[i])) fail(F3D_STATUS_RENDER, "mesh vertices contain non-finite values");
        for (uint32_t i = 0; i < d.mesh_index_count; i++)
            if (d.mesh_indices[i] >= d.mesh_vertex_count)
                fail(F3D_STATUS_RENDER, "mesh indices reference out-of-bounds vertices");
    }
}

// TerrainPtScene::new checks (terrain_heightfield.rs:132-148, :402-438)
// One pass over a DEM: two independent 64-bit hashes of its bytes (the scene cache's key, f3d_host.hip) and whether every
// sample is finite (validate_scene's question).  Round 3 made three passes -- two hashes whose multiply chains ran at
// 2.5 ms each for the 2048^2 headline DEM, and an isfinite loop -- 5.9 of the 6 ms a render of a cached DEM paid before
// its first frame.  Here four independent lanes per hash keep the multipliers busy: the pass runs at memory speed.
struct DemFingerprint {
    uint64_t key = 0, key2 = 0;
    bool finite = true;
};
inline DemFingerprint dem_fingerprint(const float *heights, size_t count) {
    const uint8_t *p = reinterpret_cast<const uint8_t *>(heights);
    const size_t n = count * sizeof(float);
    constexpr uint64_t kM1 = 0xFF51AFD7ED558CCDull, kM2 = 0xC4CEB9FE1A85EC53ull;
    uint64_t a[4] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull};
    uint64_t b[4] = {0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    uint64_t bad = 0;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t v[4];
        memcpy(v, p + i, 32);
        for (int k = 0; k < 4; k++) {
            a[k] = (a[k] ^ v[k]) * kM1;
            a[k] ^= a[k] >> 32;
            b[k] = (b[k] + v[k]) * kM2;
            b[k] ^= b[k] >> 29;
            // an exponent field of all ones (inf / NaN) carries into the sign position of its half
            bad |= ((v[k] & 0x7F8000007F800000ull) + 0x0080000000800000ull) & 0x8000000080000000ull;
        }
    }
    for (; i + 4 <= n; i += 4) {  // (a DEM is whole floats)
        uint32_t v;
        memcpy(&v, p + i, 4);
        a[(i >> 2) & 3u] = (a[(i >> 2) & 3u] ^ v) * kM1;
        b[(i >> 2) & 3u] = (b[(i >> 2) & 3u] + v) * kM2;
        bad |= (uint64_t)((v & 0x7F800000u) == 0x7F800000u);
    }
    DemFingerprint f;
    f.key = n * 0x9E3779B97F4A7C15ull;
    f.key2 = ~n * 0xD6E8FEB86659FD93ull;
    for (int k = 0; k < 4; k++) {
        f.key = (f.key ^ a[k]) * kM2;
        f.key ^= f.key >> 31;
        f.key2 = (f.key2 + b[k]) * kM1;
        f.key2 ^= f.key2 >> 33;
    }
    f.finite = bad == 0;
    return f;
}

// heights_finite: the caller has looked at every sample already (dem_fingerprint); < 0: look here.
inline void validate_scene(const f3d_terrain_ref_desc &d, int heights_finite = -1) {
    if (!finite3(d.albedo) || d.albedo[0] < 0.0f || d.albedo[1] < 0.0f || d.albedo[2] < 0.0f)
        fail(F3D_STATUS_UPLOAD, "terrain albedo must be finite and >= 0");
    if (d.dem_width < 2 || d.dem_height < 2)
        fail(F3D_STATUS_UPLOAD, "terrain heightfield must be at least 2x2 texels, got %ux%u", d.dem_width,
             d.dem_height);
    if (d.dem_width > 8193 || d.dem_height > 8193)
        fail(F3D_STATUS_UPLOAD, "terrain heightfield larger than 8193 texels per side is not supported "
             "(reference node packing, hybrid_terrain_traversal.wgsl:143-146)");
    if (!d.heights) fail(F3D_STATUS_UPLOAD, "heightfield length 0 does not match %ux%u", d.dem_width, d.dem_height);
    const size_t n = (size_t)d.dem_width * d.dem_height;
    if (heights_finite < 0) heights_finite = dem_fingerprint(d.heights, n).finite ? 1 : 0;
    if (!heights_finite) fail(F3D_STATUS_UPLOAD, "terrain heightfield contains non-finite samples");
    if (d.env_map) {
        if (d.env_width == 0 || d.env_height == 0) fail(F3D_STATUS_UPLOAD, "env map dims do not match data length");
        const size_t m = (size_t)d.env_width * d.env_height * 3;
        for (size_t i = 0; i < m; i++)
            if (!std::isfinite(d.env_map[i])) fail(F3D_STATUS_UPLOAD, "env map contains non-finite samples");
    }
}


// ---- layout of the tiled acceleration tables (f3d_scene.h) ----
struct TableLayout {
    uint32_t levels = 0;
    uint32_t level_w[kMaxLevels]{}, level_h[kMaxLevels]{};  // logical pow2 dims (reference MinMaxMips::dims)
    uint32_t dim_x[kMaxLevels]{}, dim_y[kMaxLevels]{};      // padded to multiples of 8
    uint32_t tiles_x[kMaxLevels]{};
    uint32_t node_offset[kMaxLevels]{};
    uint64_t leaf_count = 0, node_count = 0;
    uint32_t cell_w = 0, cell_h = 0;
    // row-major band tables of the march (all levels, pitch = level_w = 2^band_shift; only the rows that
    // hold cells are stored: the march never visits a node whose first cell lies outside the grid)
    uint32_t band_offset[kMaxLevels]{}, band_shift[kMaxLevels]{}, band_rows[kMaxLevels]{};
    uint64_t band_count = 0;
};

inline TableLayout table_layout(uint32_t w, uint32_t h) {
    TableLayout t;
    t.cell_w = w - 1;
    t.cell_h = h - 1;
    uint32_t lw = next_pow2(t.cell_w), lh = next_pow2(t.cell_h);
    for (;;) {
        if (t.levels >= kMaxLevels)
            fail(F3D_STATUS_UPLOAD, "terrain heightfield needs more than %u mip levels", kMaxLevels);
        const uint32_t l = t.levels;
        t.level_w[l] = lw;
        t.level_h[l] = lh;
        // the leaf table covers the cell grid only (the reference's level 0 is padded to the power of two with
        // (+inf,-inf) records nobody reads: terrain_heightfield.rs:154-155); node levels keep the pow2 padding
        t.dim_x[l] = l == 0 ? pad8(t.cell_w) : pad8(lw);
        t.dim_y[l] = l == 0 ? pad8(t.cell_h) : pad8(lh);
        t.tiles_x[l] = t.dim_x[l] / 8u;
        if (l >= 1) {
            t.node_offset[l] = (uint32_t)t.node_count;
            t.node_count += (uint64_t)t.dim_x[l] * t.dim_y[l];
        }
        t.levels++;
        if (lw == 1 && lh == 1) break;
        lw = lw / 2 > 1 ? lw / 2 : 1;
        lh = lh / 2 > 1 ? lh / 2 : 1;
    }
    t.leaf_count = (uint64_t)t.dim_x[0] * t.dim_y[0];
    for (uint32_t l = 0; l < t.levels; l++) {
        uint32_t shift = 0;
        while ((1u << shift) < t.level_w[l]) shift++;
        uint32_t rows = (t.cell_h + (1u << l) - 1u) >> l;
        rows = rows < 1u ? 1u : (rows < t.level_h[l] ? rows : t.level_h[l]);
        t.band_offset[l] = (uint32_t)t.band_count;
        t.band_shift[l] = shift;
        t.band_rows[l] = rows;
        t.band_count += (uint64_t)t.level_w[l] * rows;
    }
    return t;
}

inline void apply_layout(const TableLayout &t, TerrainDev &dev) {
    for (uint32_t l = 0; l < kMaxLevels; l++) {
        dev.node_offset[l] = t.node_offset[l];
        dev.tiles_x[l] = t.tiles_x[l];
        dev.band_offset[l] = t.band_offset[l];
        dev.band_shift[l] = t.band_shift[l];
    }
    dev.mip_count = t.levels;
    dev.cell_w = t.cell_w;
    dev.cell_h = t.cell_h;
}

inline BandBuildParams band_build_params(const TableLayout &t, uint32_t l, const LeafRec *leaves, const NodeRec *nodes,
                                         NodeRec *bands) {
    BandBuildParams b{};
    b.leaves = leaves;
    b.src = l >= 1 ? nodes + t.node_offset[l] : nullptr;
    b.dst = bands + t.band_offset[l];
    b.level = l;
    b.width = t.level_w[l];
    b.height = t.band_rows[l];
    b.shift = t.band_shift[l];
    b.src_tiles_x = t.tiles_x[l];
    b.cell_w = t.cell_w;
    b.cell_h = t.cell_h;
    return b;
}

inline PyramidBuildParams leaf_build_params(const TableLayout &t, const float *heights, uint32_t w, uint32_t h,
                                            float exaggeration, LeafRec *leaves) {
    PyramidBuildParams b{};
    b.heights = heights;
    b.w = w;
    b.h = h;
    b.exaggeration = exaggeration;
    b.leaves = leaves;
    b.leaf_tiles_x = t.tiles_x[0];
    b.leaf_dim_x = t.dim_x[0];
    b.leaf_dim_y = t.dim_y[0];
    return b;
}

inline LevelBuildParams level_build_params(const TableLayout &t, uint32_t l, const LeafRec *leaves, NodeRec *nodes) {
    LevelBuildParams b{};
    b.leaves = leaves;
    b.src = l >= 2 ? nodes + t.node_offset[l - 1] : nullptr;
    b.dst = nodes + t.node_offset[l];
    b.level = l;
    b.src_w = t.level_w[l - 1];
    b.src_h = t.level_h[l - 1];
    b.dst_w = t.level_w[l];
    b.dst_h = t.level_h[l];
    b.src_tiles_x = t.tiles_x[l - 1];
    b.dst_tiles_x = t.tiles_x[l];
    b.dst_dim_x = t.dim_x[l];
    b.dst_dim_y = t.dim_y[l];
    b.cell_w = t.cell_w;
    b.cell_h = t.cell_h;
    return b;
}

// Everything of FrameParams that does not point into device memory: terrain transform,
// curvature, camera, lighting (render_terrain.rs:571-576, :635-742).  Returns whether a
// sun-lit scene must end with valid reservoirs (render_terrain.rs:465-471).
inline bool fill_uniforms(const f3d_terrain_ref_desc &d, FrameParams &P) {
    const float kScaleMax = 65504.0f;  // AETHER_RADIOMETRIC_SCALE_MAX (reference src/core/atmosphere/mod.rs:11)
    const float exposure = f_clamp(d.exposure, 0.0f, kScaleMax);
    const float sun_intensity = f_clamp(d.sun_intensity, 0.0f, kScaleMax);
    const float sun_color[3] = {f_clamp(d.sun_color[0], 0.0f, kScaleMax), f_clamp(d.sun_color[1], 0.0f, kScaleMax),
                                f_clamp(d.sun_color[2], 0.0f, kScaleMax)};
    const float env_intensity = f_clamp(d.env_intensity, 0.0f, kScaleMax);

    // EarthCurvatureUniforms::new, terrain_heightfield.rs:52-84
    if (!std::isfinite(d.observer_latitude_deg) || d.observer_latitude_deg < -90.0 || d.observer_latitude_deg > 90.0 ||
        !std::isfinite(d.observer_longitude_deg) || d.observer_longitude_deg < -180.0 ||
        d.observer_longitude_deg > 180.0)
        fail(F3D_STATUS_RENDER, "ray-origin latitude/longitude must be finite and in [-90,90]/[-180,180]");
    const double radius = effective_radius(d.earth_model, d.observer_latitude_deg, d.sphere_radius_m, d.refraction_model,
                                           d.pressure_mbar, d.temperature_c, d.refraction_k, (double)d.sun_azimuth_deg);
    const bool curved = std::isfinite(radius);

    P.terrain.origin_x = -0.5f * ((float)d.dem_width - 1.0f) * d.spacing_x;  // terrain_heightfield.rs:359-360
    P.terrain.origin_z = -0.5f * ((float)d.dem_height - 1.0f) * d.spacing_z;
    P.terrain.spacing_x = d.spacing_x;
    P.terrain.spacing_z = d.spacing_z;
    P.terrain.inv_spacing_x = 1.0f / d.spacing_x;
    P.terrain.inv_spacing_z = 1.0f / d.spacing_z;
    P.terrain.inv_two_r_prime = curved ? (float)(0.5 / radius) : 0.0f;
    P.terrain.curvature_enabled = curved ? 1u : 0u;

    P.env.width = 0;
    P.env.height = 0;
    P.env.texels = nullptr;
    P.env.intensity = env_intensity;
    P.mesh.traversal_mode = 3u;
    P.mesh.vertices = nullptr;
    P.mesh.indices = nullptr;
    P.mesh.vertex_count = 0;
    P.mesh.index_count = 0;
    P.mesh.bvh_nodes = nullptr;
    P.mesh.bvh_tris = nullptr;
    P.mesh.bvh4_nodes = nullptr;
    P.mesh.bvh4_node_count = 0u;
    P.mesh.bvh_node_count = 0;

    const float kDegF = 0.017453292519943295f;
    const V3 origin{d.cam_origin[0], d.cam_origin[1], d.cam_origin[2]};
    const V3 forward = normalize(V3{d.cam_look_at[0], d.cam_look_at[1], d.cam_look_at[2]} - origin);
    const V3 right = normalize(cross(forward, V3{d.cam_up[0], d.cam_up[1], d.cam_up[2]}));
    const V3 up = normalize(cross(right, forward));
    const float az = d.sun_azimuth_deg * kDegF, el = d.sun_elevation_deg * kDegF;
    const V3 light_dir{cosf(az) * cosf(el), sinf(el), sinf(az) * cosf(el)};
    P.cam.origin = origin;
    P.cam.right = right;
    P.cam.up = up;
    P.cam.forward = forward;
    P.cam.half_h = tanf(0.5f * (d.fov_y_deg * kDegF));
    P.cam.half_w = ((float)d.width / (float)d.height) * P.cam.half_h;
    P.cam.exposure = exposure;
    P.cam.width = d.width;
    P.cam.height = d.height;
    P.cam.seed_hi = d.seed;
    P.cam.seed_lo = d.seed ^ 0x85EBCA6Bu;
    P.cam.cone_delta = pixel_cone_delta_of(P.cam.half_w, P.cam.half_h, d.width, d.height);
    P.light.wi = normalize(light_dir);
    P.light.wi_reuse = normalize(P.light.wi);
    P.light.color = V3{sun_intensity * sun_color[0], sun_intensity * sun_color[1], sun_intensity * sun_color[2]};
    P.light.albedo = V3{d.albedo[0], d.albedo[1], d.albedo[2]};
    P.light.shadows_enabled = 1u;
    P.spp = d.spp > 1u ? d.spp : 1u;
    return d.sun_elevation_deg > 0.0f && sun_intensity > 0.0f &&
           (sun_color[0] > 0.0f || sun_color[1] > 0.0f || sun_color[2] > 0.0f);
}

// vec4-padded copies the device wants (env: RGBA32F texels; mesh: reference MeshVertex)
inline std::vector<float> pad_rgb_to_rgba(const float *rgb, size_t n, float w) {
    std::vector<float> out(n * 4);
    for (size_t i = 0; i < n; i++) {
        out[4 * i] = rgb[3 * i];
        out[4 * i + 1] = rgb[3 * i + 1];
        out[4 * i + 2] = rgb[3 * i + 2];
        out[4 * i + 3] = w;
    }
    return out;
}

}  // namespace f3d
