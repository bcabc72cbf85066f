/* oracle/f3d_oracle.h
 *
 * TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of forge3d's PROMETHEUS
 * terrain path-trace hot path.  Nothing under forge3d_amd/ (the product) may
 * include, link, import or execute anything in this directory; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the
 * checker / CPU baseline.
 *
 * Pinning: the restatement is checked against the reference's own committed golden
 * (tests/golden/mini_dem_reference.png, gate SSIM >= 0.995 and mean-abs <= 2.0 from
 * reference tests/test_hybrid_terrain_pt.py:818-859) and against the Rust KATs of
 * src/path_tracing/hybrid_compute/terrain_heightfield.rs:529-616,1971-2127 restated
 * in tests/test_oracle_*.py.  The reference itself (Rust + wgpu + WGSL) cannot be
 * compiled or run in this image, so there is no oracle/_ref build.
 */
#ifndef F3D_ORACLE_H
#define F3D_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Canonical 80-byte ReSTIR reservoir (reference src/path_tracing/restir/types.rs:6-37,
 * WGSL mirror src/shaders/hybrid_terrain_traversal.wgsl:39-54). */
typedef struct {
    float position[3];
    uint32_t light_index;
    float direction[3];
    float intensity;
    uint32_t light_type;
    float params[3];
    uint32_t pad[4];
    float w_sum;
    uint32_t m;
    float weight;
    float target_pdf;
} f3do_reservoir;

/* AETHER LUT payload (reference AtmosphereLuts, src/core/atmosphere/bake.rs:410-423) decoded to f32 RGBA:
 * dims = {transmittance mu, height, scattering view, sun, height, nu, aerial distance, mu, height}. */
typedef struct {
    const float *transmittance, *scattering, *aerial;
    uint32_t dims[9];
    float turbidity, ozone_du, mie_g, bottom_radius_m, top_radius_m, rayleigh_scale_height_m, mie_scale_height_m,
        max_aerial_distance_m, ground_albedo;
    uint32_t scattering_orders;
} f3do_aether;

/* Scene description == reference TerrainReferenceDesc
 * (src/path_tracing/hybrid_compute/render_terrain.rs:239-282). */
typedef struct {
    const float *heights; /* row-major (dem_h, dem_w) */
    uint32_t dem_w, dem_h;
    float spacing_x, spacing_z;
    float exaggeration;
    float albedo[3];
    float cam_origin[3], cam_look_at[3], cam_up[3];
    float fov_y_deg, exposure;
    float sun_azimuth_deg, sun_elevation_deg, sun_intensity;
    float sun_color[3];
    double observer_lat_deg, observer_lon_deg;
    /* earth_model: 0 flat, 1 sphere, 2 ellipsoid; refraction_model: 0 none,
     * 1 bennett, 2 saemundsson, 3 effective_radius (src/geo/refraction.rs:15-43). */
    int32_t earth_model;
    int32_t refraction_model;
    double sphere_radius_m, refraction_k, pressure_mbar, temperature_c;
    const float *env_map; /* (env_h, env_w, 3) or NULL */
    uint32_t env_w, env_h;
    float env_intensity;
    const float *mesh_vertices; /* (n,3) or NULL */
    uint32_t mesh_vertex_count;
    const uint32_t *mesh_indices; /* flat, 3 per triangle */
    uint32_t mesh_index_count;
    uint32_t width, height;
    uint32_t seed, spp, max_frames, min_frames;
    float variance_threshold;
    const f3do_aether *atmosphere; /* NULL = no aerial-perspective post */
    /* Composition hook (no counterpart in the reference): (H*W, 4) radiance sums + frame count that REPLACE the path
     * tracer's own accumulation before the resolve / the AETHER post -- how BASELINE.json configs[2] combines the PBR
     * tracer's multi-bounce radiance over the DEM ("GI") with the atmosphere post (forge3d_amd.offline.render_terrain_gi).
     * NULL = the terrain tracer's own accumulation. */
    const float *accum_override;
} f3do_desc;

typedef struct {
    uint8_t *rgba;   /* (H,W,4) */
    float *albedo;   /* (H,W,3) */
    float *normal;   /* (H,W,3) */
    float *depth;    /* (H,W)   */
    /* optional internal state dumps (may be NULL) */
    float *accum;              /* (H*W,4) */
    float *welford;            /* (H*W,2) */
    f3do_reservoir *reservoir_prev; /* (H*W) */
    uint32_t frames;
    float variance;
    int32_t converged;
    uint64_t minmax_pyramid_bytes;
    /* traversal counters summed over every terrain_trace call of the render
     * (n_node = min-max texel fetches, n_leaf = leaf tests,
     * n_hit = accepted leaf hits with the normal re-fetch) and total samples. */
    uint64_t n_node, n_leaf, n_hit, n_samples, n_rays;
    double loop_seconds; /* wall time of the accumulation loop only */
} f3do_out;

/* Returns 0 on success; on failure writes a message into err and returns
 * 1 (validation error -> ValueError) or 2 (render error -> RuntimeError). */
int f3do_render(const f3do_desc *desc, f3do_out *out, char *err, size_t errlen);

/* build_minmax_mips (terrain_heightfield.rs:132-202).  levels_out receives the
 * levels back to back (finest first), each (ph_l, pw_l, 2) floats; dims_out gets
 * (pw_l, ph_l) pairs.  Pass NULL levels_out to query sizes.  Returns the level
 * count, or a negative error code. */
int f3do_build_minmax_mips(const float *heights, uint32_t w, uint32_t h, float *levels_out,
                           uint32_t *dims_out, uint32_t max_levels, uint64_t *total_floats);

/* terrain_trace on an arbitrary ray batch (test hook mirroring the reference's
 * main_helios_production_terrain_trace_proof entry, terrain_heightfield.rs:1646-1671).
 * rays: n x 8 floats (origin xyz, tmin, direction xyz, tmax).  origin_x/z = world xz
 * of texel (0,0).  out_hit n u32; out_t n floats; out_normal n x 3 floats (may be NULL). */
int f3do_terrain_trace_batch(const float *heights, uint32_t w, uint32_t h, float origin_x,
                             float origin_z, float spacing_x, float spacing_z,
                             float exaggeration, float inv_two_r_prime,
                             uint32_t curvature_enabled, const float *rays, uint32_t n,
                             int32_t any_hit, int32_t apply_curvature, uint32_t *out_hit,
                             float *out_t, float *out_normal, uint64_t *counters3);

/* effective_radius_m (src/geo/refraction.rs:137-148); returns 0 ok / 1 error. */
int f3do_effective_radius_m(int32_t earth_model, double latitude_deg, double sphere_radius_m,
                            int32_t refraction_model, double pressure_mbar,
                            double temperature_c, double k, double azimuth_deg,
                            double *radius_out, char *err, size_t errlen);

/* The leaf quadratic alone: does the parabola through d(0)=d0, d(1/2)=dm, d(1)=d1 have a
 * root in [0,1] (or, for any_hit, start at/below the surface)?  Mirrors the reference's
 * unit-test helper deviation_span_hit (terrain_heightfield.rs:722-729). */
int f3do_deviation_span_hit(float d0, float dm, float d1, int32_t any_hit);

/* f32 -> f16 (RNE) -> f32 round trip, exposed for tests. */
float f3do_f16_round(float v);

/* Deterministic sin/cos of 2*pi*u used by terrain_cosine_dir (exposed for tests). */
void f3do_sincos_2pi(float u, float *s, float *c);

int f3do_num_threads(void);
void f3do_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
