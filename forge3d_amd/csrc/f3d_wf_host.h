// forge3d_amd/csrc/f3d_wf_host.h -- host side of the multi-bounce PBR tracer that needs no HIP: validation of an
// f3d_wf_scene and its translation into the arrays the path code reads (f3d_wf_path.h).  Shared by the driver
// (f3d_wavefront.hip, which uploads the arrays) and by the test emulator (which reads them in place).
// Everything that does not depend on a hit -- normalised light directions, radiances, disc frames, clamped
// importances and their sums, the camera's half extents -- is computed here once, with the same f32 helpers the
// device uses, so the per-hit arithmetic of the reference (pt_shade.wgsl:118-152, 598-697) keeps its results.
#pragma once
#include <cstddef>
#include <cstring>

#include <cmath>
#include <vector>

#include "../../include/f3d_wavefront.h"
#include "f3d_setup.h"
#include "f3d_wf_path.h"

namespace f3d {
namespace wf {

inline bool finite_n(const float *v, int n) {
    for (int i = 0; i < n; i++)
        if (!std::isfinite(v[i])) return false;
    return true;
}

// The caller's f3d_wf_scene as this revision of the header lays it out: the struct has only ever grown at its end, so a
// caller built against an earlier revision (no `primary_start`) is read up to its own size and the rest is zero.
inline f3d_wf_scene scene_of_caller(const f3d_wf_scene *s) {
    uint32_t size = 0;
    memcpy(&size, s, sizeof(size));  // (struct_size is the first member in every revision)
    if (size != sizeof(f3d_wf_scene) && size != offsetof(f3d_wf_scene, primary_start))
        fail(F3D_STATUS_VALUE, "f3d_wf_scene.struct_size is %u, this library expects %zu: the caller was built against another revision of f3d_wavefront.h",
             size, sizeof(f3d_wf_scene));
    f3d_wf_scene full{};
    memcpy(&full, s, size);
    full.struct_size = (uint32_t)sizeof(f3d_wf_scene);
    return full;
}

inline void validate_scene(const f3d_wf_scene &s, uint32_t width, uint32_t height, uint32_t frame_count) {
    if (s.struct_size != sizeof(f3d_wf_scene))
        fail(F3D_STATUS_VALUE, "f3d_wf_scene.struct_size is %u, this library expects %zu: the caller was built against another revision of f3d_wavefront.h",
             s.struct_size, sizeof(f3d_wf_scene));
    if (width == 0u || height == 0u || frame_count == 0u)  // adjudication.rs:85-89
        fail(F3D_STATUS_RENDER, "adjudication PT reference requires non-zero width/height/spp");
    if ((uint64_t)width * height > (1ull << 31)) fail(F3D_STATUS_VALUE, "image too large");
    if (s.sphere_count == 0u || !s.spheres) fail(F3D_STATUS_VALUE, "the sphere / material table needs at least one entry");
    for (uint32_t i = 0; i < s.sphere_count; i++) {
        const f3d_wf_sphere &q = s.spheres[i];
        if (!finite_n(q.center, 3) || !std::isfinite(q.radius) || !finite_n(q.albedo, 3) || !finite_n(q.emissive, 3) ||
            !std::isfinite(q.metallic) || !std::isfinite(q.roughness) || !std::isfinite(q.ior) || !std::isfinite(q.ax) || !std::isfinite(q.ay))
            fail(F3D_STATUS_VALUE, "sphere %u has non-finite parameters", i);
    }
    if (s.mesh_count && !s.meshes) fail(F3D_STATUS_VALUE, "meshes is NULL");
    for (uint32_t m = 0; m < s.mesh_count; m++) {
        const f3d_wf_mesh &q = s.meshes[m];
        if ((q.vertex_count && !q.vertices) || (q.triangle_count && !q.indices)) fail(F3D_STATUS_VALUE, "mesh %u has NULL arrays", m);
        if (q.triangle_count > (1u << 28)) fail(F3D_STATUS_VALUE, "mesh %u has too many triangles", m);
        for (uint64_t k = 0; k < 3ull * q.vertex_count; k++)
            if (!std::isfinite(q.vertices[k])) fail(F3D_STATUS_VALUE, "mesh %u has a non-finite vertex", m);
    }
    if (s.instance_count && !s.instances) fail(F3D_STATUS_VALUE, "instances is NULL");
    for (uint32_t i = 0; i < s.instance_count; i++) {
        if (s.instances[i].blas_index >= s.mesh_count) fail(F3D_STATUS_VALUE, "instance %u refers to BLAS %u of %u", i, s.instances[i].blas_index, s.mesh_count);
        if (!finite_n(s.instances[i].world_to_object, 16)) fail(F3D_STATUS_VALUE, "instance %u has a non-finite transform", i);
    }
    if ((s.dir_light_count && !s.dir_lights) || (s.area_light_count && !s.area_lights) || (s.importance_count && !s.object_importance))
        fail(F3D_STATUS_VALUE, "light / importance arrays are NULL");
    for (uint32_t i = 0; i < s.dir_light_count; i++)
        if (!finite_n(s.dir_lights[i].direction, 3) || !finite_n(s.dir_lights[i].color, 3) || !std::isfinite(s.dir_lights[i].intensity) ||
            !std::isfinite(s.dir_lights[i].importance))
            fail(F3D_STATUS_VALUE, "directional light %u has non-finite parameters", i);
    for (uint32_t i = 0; i < s.area_light_count; i++)
        if (!finite_n(s.area_lights[i].position, 3) || !finite_n(s.area_lights[i].normal, 3) || !finite_n(s.area_lights[i].color, 3) ||
            !std::isfinite(s.area_lights[i].radius) || !std::isfinite(s.area_lights[i].intensity) || !std::isfinite(s.area_lights[i].importance))
            fail(F3D_STATUS_VALUE, "area light %u has non-finite parameters", i);
    if (!finite_n(s.env_ground, 3) || !finite_n(s.env_sky, 3) || !finite_n(s.miss_ground, 3) || !finite_n(s.miss_sky, 3))
        fail(F3D_STATUS_VALUE, "environment colours must be finite");
    if (!finite_n(s.cam_origin, 3) || !finite_n(s.cam_right, 3) || !finite_n(s.cam_up, 3) || !finite_n(s.cam_forward, 3))
        fail(F3D_STATUS_VALUE, "camera vectors must be finite");
    if (!(std::isfinite(s.cam_fov_y) && s.cam_fov_y > 0.0f && s.cam_fov_y < 3.14159265f)) fail(F3D_STATUS_VALUE, "cam_fov_y must be in (0, pi) radians");
    if (!(std::isfinite(s.cam_exposure) && s.cam_exposure >= 0.0f)) fail(F3D_STATUS_VALUE, "cam_exposure must be finite and >= 0");
    if (s.hair_count && !s.hair) fail(F3D_STATUS_VALUE, "hair_count is %u but hair is null", s.hair_count);
    for (uint32_t i = 0; i < s.hair_count; i++)
        if (!finite_n(s.hair[i].p0, 3) || !finite_n(s.hair[i].p1, 3) || !std::isfinite(s.hair[i].r0) || !std::isfinite(s.hair[i].r1))
            fail(F3D_STATUS_VALUE, "hair segment %u has non-finite parameters", i);
    if (!std::isfinite(s.medium.g) || !std::isfinite(s.medium.sigma_t) || !std::isfinite(s.medium.density) || !std::isfinite(s.medium.enabled))
        fail(F3D_STATUS_VALUE, "medium parameters must be finite");
    if (s.terrain) {  // the heightfield primitive: the terrain tracer's own input rules (render_terrain.rs:474-557)
        const f3d_wf_terrain &t = *s.terrain;
        if (!t.heights || t.dem_width < 2u || t.dem_height < 2u) fail(F3D_STATUS_UPLOAD, "terrain heightfield must be at least 2x2 texels, got %ux%u", t.dem_width, t.dem_height);
        if (t.dem_width > 8193u || t.dem_height > 8193u) fail(F3D_STATUS_UPLOAD, "terrain heightfield is limited to 8193 texels a side, got %ux%u", t.dem_width, t.dem_height);
        if (!(std::isfinite(t.spacing_x) && t.spacing_x > 0.0f && std::isfinite(t.spacing_z) && t.spacing_z > 0.0f))
            fail(F3D_STATUS_RENDER, "terrain spacing must be finite and positive");
        if (!(std::isfinite(t.exaggeration) && t.exaggeration > 0.0f)) fail(F3D_STATUS_RENDER, "terrain exaggeration must be finite and positive");
        for (size_t i = 0; i < (size_t)t.dem_width * t.dem_height; i++)
            if (!std::isfinite(t.heights[i])) fail(F3D_STATUS_UPLOAD, "terrain heightfield contains non-finite samples");
    }
}

inline V3 v3p(const float *p) { return V3{p[0], p[1], p[2]}; }


struct PreparedScene {
    std::vector<SphereDev> spheres;
    std::vector<MaterialDev> mats;
    std::vector<MeshBvh> bvh;
    std::vector<InstanceDev> inst;
    std::vector<DirLightDev> dir;
    std::vector<AreaLightDev> area;
    std::vector<HairDev> hair;
    SceneDev S;  // counts and scalars filled in; the array pointers are the caller's to set
};

inline PreparedScene prepare_scene(const f3d_wf_scene &s, uint32_t width, uint32_t height) {
    PreparedScene out;
    SceneDev &S = out.S;
    S = SceneDev{};
    out.spheres.resize(s.sphere_count);
    out.mats.resize(s.sphere_count);
    for (uint32_t i = 0; i < s.sphere_count; i++) {
        const f3d_wf_sphere &q = s.spheres[i];
        out.spheres[i] = SphereDev{v3p(q.center), q.radius};
        out.mats[i] = MaterialDev{v3p(q.albedo), q.metallic, v3p(q.emissive), q.roughness, q.ior, q.ax, q.ay,
                                  i < s.importance_count ? s.object_importance[i] : 1.0f};
    }
    S.sphere_count = s.sphere_count;
    out.bvh.reserve(s.mesh_count);
    for (uint32_t m = 0; m < s.mesh_count; m++)
        out.bvh.push_back(build_mesh_bvh(s.meshes[m].vertices, s.meshes[m].vertex_count, s.meshes[m].indices, 3u * s.meshes[m].triangle_count));
    S.blas_count = s.mesh_count;
    out.inst.resize(s.instance_count);
    for (uint32_t i = 0; i < s.instance_count; i++) {
        for (int k = 0; k < 16; k++) out.inst[i].w2o[k] = s.instances[i].world_to_object[k];
        out.inst[i].blas = s.instances[i].blas_index;
        out.inst[i].material = s.instances[i].material_id < s.sphere_count - 1u ? s.instances[i].material_id : s.sphere_count - 1u;
        out.inst[i].pad0 = out.inst[i].pad1 = 0u;
    }
    S.inst_count = s.instance_count;
    out.dir.resize(s.dir_light_count);
    S.dir_sum_imp = 0.0f;
    for (uint32_t i = 0; i < s.dir_light_count; i++) {
        const f3d_wf_dir_light &q = s.dir_lights[i];
        out.dir[i].wi = normalize(neg(v3p(q.direction)));
        out.dir[i].importance = f_max(q.importance, 0.0f);
        out.dir[i].Li = v3p(q.color) * q.intensity;
        out.dir[i].pad = 0.0f;
        S.dir_sum_imp = S.dir_sum_imp + out.dir[i].importance;
    }
    S.dir_count = s.dir_light_count;
    out.area.resize(s.area_light_count);
    S.area_sum_imp = 0.0f;
    for (uint32_t i = 0; i < s.area_light_count; i++) {
        const f3d_wf_area_light &q = s.area_lights[i];
        AreaLightDev &a = out.area[i];
        a.position = v3p(q.position);
        a.rad = f_max(q.radius, 1e-6f);
        a.nL = normalize(v3p(q.normal));
        const Frame3 fr = tangent_frame(a.nL);
        a.tL = V3{fr.t.x, fr.b.x, fr.n.x};  // rows of the basis matrix, like pt_shade.wgsl:122-123
        a.bL = V3{fr.t.y, fr.b.y, fr.n.y};
        a.importance = f_max(q.importance, 0.0f);
        a.p_area = 1.0f / ((kPi * a.rad) * a.rad);
        a.Li = v3p(q.color) * q.intensity;
        a.pad = a.pad2 = 0.0f;
        S.area_sum_imp = S.area_sum_imp + a.importance;
    }
    S.area_count = s.area_light_count;
    S.env_ground = v3p(s.env_ground);
    S.env_sky = v3p(s.env_sky);
    S.miss_ground = v3p(s.miss_ground);
    S.miss_sky = v3p(s.miss_sky);
    S.cam_origin = v3p(s.cam_origin);
    S.cam_right = v3p(s.cam_right);
    S.cam_up = v3p(s.cam_up);
    S.cam_neg_forward = neg(v3p(s.cam_forward));
    S.half_h = std::tan(0.5f * s.cam_fov_y);
    S.half_w = ((float)width / (float)height) * S.half_h;
    S.width = width;
    S.height = height;
    S.seed_hi = s.seed_hi;
    S.seed_lo = s.seed_lo;
    out.hair.resize(s.hair_count);
    for (uint32_t i = 0; i < s.hair_count; i++)
        out.hair[i] = HairDev{v3p(s.hair[i].p0), s.hair[i].r0, v3p(s.hair[i].p1), s.hair[i].r1, s.hair[i].material_id, 0u, 0u, 0u};
    S.hair_count = s.hair_count;
    S.medium_on = s.medium.enabled > 0.5f ? 1u : 0u;  // pt_shade.wgsl:500-502
    S.medium_mu = f_max(s.medium.sigma_t * s.medium.density, 0.0f);
    S.has_terrain = 0u;
    S.primary_start = s.terrain ? reinterpret_cast<const uint2 *>(s.primary_start) : nullptr;  // (a device pointer: f3d_wavefront.h)
    if (s.terrain) {  // placement and scalars as fill_uniforms (f3d_setup.h); tables are the caller's to attach
        const f3d_wf_terrain &t = *s.terrain;
        S.has_terrain = 1u;
        S.terrain_mat = t.material_id < s.sphere_count - 1u ? t.material_id : s.sphere_count - 1u;
        S.terrain.origin_x = -0.5f * ((float)t.dem_width - 1.0f) * t.spacing_x;
        S.terrain.origin_z = -0.5f * ((float)t.dem_height - 1.0f) * t.spacing_z;
        S.terrain.spacing_x = t.spacing_x;
        S.terrain.spacing_z = t.spacing_z;
        S.terrain.inv_spacing_x = 1.0f / t.spacing_x;
        S.terrain.inv_spacing_z = 1.0f / t.spacing_z;
        S.terrain.inv_two_r_prime = 0.0f;
        S.terrain.curvature_enabled = 0u;
    }
    return out;
}

}  // namespace wf
}  // namespace f3d
