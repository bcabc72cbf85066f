/* f3d_wavefront.h -- C ABI of the multi-bounce PBR path tracer in libf3dhip.so.
 *
 * Replaces, for forge3d's AEQUITAS adjudication path (src/py_functions/adjudication.rs:19-165):
 *   path_tracing::adjudication::render_pt_reference      src/path_tracing/adjudication.rs:76-364
 *     = WavefrontScheduler::render_frame_simple          src/path_tracing/wavefront/render.rs:87-208
 *       over pt_raygen / pt_intersect / pt_shade / pt_shadow / pt_scatter.wgsl, one 1-spp frame per call
 *   core::tonemap::resolve_reference_hdr_to_rgba8        src/core/tonemap.rs:11-30
 * The structs mirror the buffers that driver binds: WavefrontGpuSphere (reference_scene.rs:90-104), GpuDirectionalLight
 * / GpuAreaLight (path_tracing/lighting.rs), accel::instancing::InstanceData, the mesh atlas (one entry per BLAS),
 * ReferenceEnvironmentRaw and WavefrontUniforms (adjudication.rs:22-38).  All pointers are HOST pointers read during
 * the call only.  Status codes and messages as in f3d_terrain_pt.h.  No CPU fallback: status 4 without a HIP device.
 */
#ifndef F3D_WAVEFRONT_H
#define F3D_WAVEFRONT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct f3d_wf_sphere { /* geometry + the material slot of the same index; radius 0 = material only */
    float center[3], radius, albedo[3], metallic, roughness, ior, emissive[3], ax, ay;
} f3d_wf_sphere;

typedef struct f3d_wf_dir_light {
    float direction[3]; /* the direction the light travels; GpuDirectionalLight::new normalises it on the host */
    float intensity, color[3], importance;
} f3d_wf_dir_light;

typedef struct f3d_wf_area_light { /* disc */
    float position[3], radius, normal[3], intensity, color[3], importance;
} f3d_wf_area_light;

typedef struct f3d_wf_instance {
    float object_to_world[16], world_to_object[16]; /* column-major 4x4 */
    uint32_t blas_index;  /* < mesh_count */
    uint32_t material_id; /* clamped to sphere_count - 1 like pt_intersect.wgsl:476-481 */
} f3d_wf_instance;

typedef struct f3d_wf_mesh { /* one BLAS */
    const float *vertices; /* vertex_count x 3 */
    uint32_t vertex_count;
    const uint32_t *indices; /* triangle_count x 3 */
    uint32_t triangle_count;
} f3d_wf_mesh;

/* Optional heightfield primitive: the terrain of f3d_terrain_ref_desc (same placement: centred on the world origin, y up,
 * DEM row = +z, heights * exaggeration) as one more object of the PBR tracer's scene.  NOT in the reference's wavefront
 * tracer (spheres + instanced meshes only, pt_intersect.wgsl:431-558); BASELINE.json configs[2] ("atmosphere + GI" over a
 * DEM) needs it.  Closest / any hits are those of terrain_trace (hybrid_terrain_traversal.wgsl:254-372), curvature off. */
typedef struct f3d_wf_terrain {
    const float *heights; /* (dem_height, dem_width) row-major f32 */
    uint32_t dem_width, dem_height;
    float spacing_x, spacing_z, exaggeration;
    uint32_t material_id; /* slot of the sphere / material table; clamped to sphere_count - 1 like instance materials */
} f3d_wf_terrain;

typedef struct f3d_wf_hair_segment { /* HairSegment, pt_intersect.wgsl:60-69: a world-space cylinder between p0 and p1 */
    float p0[3], r0, p1[3], r1; /* the radius traced is max(0, (r0 + r1) / 2) (HAIR_RADIUS_SCALE = 1) */
    uint32_t material_id;       /* slot of the sphere / material table, clamped to sphere_count - 1 */
    uint32_t pad[3];
} f3d_wf_hair_segment;

typedef struct f3d_wf_medium { /* MediumParams, pt_shade.wgsl:34-39 (WavefrontScheduler::set_medium_params, control.rs:114) */
    float g;       /* Henyey-Greenstein anisotropy: carried, unused by the reference's shader */
    float sigma_t; /* extinction coefficient */
    float density; /* scale: the homogeneous fog has mu = sigma_t * density */
    float enabled; /* > 0.5: every next-event contribution of a vertex is attenuated over the segment that reached it,
                      and a primary hit adds env(-wo) * (1 - T) */
} f3d_wf_medium;

typedef struct f3d_wf_scene {
    uint32_t struct_size; /* = sizeof(f3d_wf_scene) of the caller's header (F3D_ABI_VERSION, f3d_terrain_pt.h) */
    const f3d_wf_sphere *spheres;
    uint32_t sphere_count; /* >= 1: the material table */
    const f3d_wf_mesh *meshes;
    uint32_t mesh_count;
    const f3d_wf_instance *instances;
    uint32_t instance_count; /* 0 with a mesh: BLAS 0 in world space with material 0 (pt_intersect.wgsl:456-465) */
    const f3d_wf_dir_light *dir_lights;
    uint32_t dir_light_count;
    const f3d_wf_area_light *area_lights;
    uint32_t area_light_count;
    const float *object_importance; /* per material slot; slots beyond importance_count weigh 1 */
    uint32_t importance_count;
    float env_ground[4], env_sky[4];   /* environment light seen by next-event estimation: mix over 0.5 (wi.y + 1) */
    float miss_ground[4], miss_sky[4]; /* background seen by rays that leave the scene */
    float cam_origin[3], cam_right[3], cam_up[3], cam_forward[3]; /* orthonormal, as ReferenceSceneDesc::camera_basis */
    float cam_fov_y;                                             /* radians */
    float cam_exposure;
    uint32_t seed_hi, seed_lo; /* ReferenceSceneDesc seeds; frame f runs with splitmix32(seed ^ f ...) (adjudication.rs:231) */
    const f3d_wf_terrain *terrain; /* NULL: no heightfield */
    const f3d_wf_hair_segment *hair; /* hair strands as cylinder segments (Kajiya-Kay continuation, pt_shade.wgsl:708-729); they */
    uint32_t hair_count;             /* are visible to closest-hit rays only: the reference's shadow stage does not test them */
    f3d_wf_medium medium;            /* all zero: no fog */
    /* (ABI 5) optional DEVICE pointer, width x height records {f32 bits of t_clear, u32 level}: the terrain tracer's
     * primary-ray certificates for the SAME camera, image size and heightfield (f3d_session_primary_start of a full-frame
     * session; f3d_cone.h): every camera ray of a pixel is above every cell it passes before t_clear, so its march may
     * start there.  The paths are the same with and without it (tests/test_offline_gi.py); NULL: none. */
    const void *primary_start;
} f3d_wf_scene;

typedef struct f3d_wf_out {
    float *hdr;    /* optional, height x width x 4: mean radiance over all frames so far, alpha 1 */
    uint8_t *rgba; /* optional, height x width x 4: Reinhard(hdr * exposure) -> sRGB -> u8, alpha 255 */
    float *accum;  /* optional IN/OUT, height x width x 4: running per-pixel sums (NULL or zeros before frame 0); lets a caller
                      continue a render: pass the sums of frames [0, first_frame) and get those of [0, first + count) */
    double loop_seconds;    /* device time of the path-tracing + fold launches */
    uint64_t paths;         /* camera paths traced (= width * height * frame_count) */
    uint64_t path_vertices; /* closest-hit queries traced (path segments) */
} f3d_wf_out;

/* Accumulate frames [first_frame, first_frame + frame_count), one sample per pixel per frame.
 * frames_per_launch = frames traced per round (a round = one path-tracing launch over all pixels x those frames + one
 * fold launch; 16 bytes of device memory per pixel-frame of a round); 0 sizes rounds to a 4 GiB buffer.  The result does
 * not depend on it.  device < 0 keeps the current device. */
int f3d_wavefront_render(const f3d_wf_scene *scene, uint32_t width, uint32_t height, uint32_t first_frame,
                         uint32_t frame_count, uint32_t frames_per_launch, int32_t device, f3d_wf_out *out, char *err,
                         size_t errlen);

#ifdef __cplusplus
}
#endif
#endif /* F3D_WAVEFRONT_H */
