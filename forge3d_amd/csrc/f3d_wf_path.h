// forge3d_amd/csrc/f3d_wf_path.h -- one pixel's paths of the multi-bounce PBR tracer, host + device.
//
// Reference: the wavefront tracer behind render_pt_reference (src/path_tracing/adjudication.rs:76-364):
// pt_raygen.wgsl:162-226 -> up to 16 x { pt_intersect.wgsl:431-558, pt_shade.wgsl:460-862, pt_shadow.wgsl:248-294,
// pt_scatter.wgsl:76-133 } with a host read-back of the queue header between bounces (wavefront/render.rs:87-208).
//
// MI355X form.  A pixel's rays never meet another pixel's, so the five queues, their atomics and the per-bounce host
// round trip are a schedule, not part of the result.  Here a lane owns a pixel for a whole batch of frames and runs a
// FLAT loop whose body is one path vertex: (new camera ray if the previous path ended) -> closest hit -> on a miss
// add the background and end the path, otherwise next-event estimation with its shadow rays traced on the spot,
// continuation sample, roulette.  A lane whose path ends starts its next frame's path in the very next iteration, so
// lanes of a wave never wait for the longest path of a frame.  A frame's contributions are summed from zero in the
// order the reference's stages add them and the frame totals are folded into the pixel in frame order -- the order
// oracle/wavefront_oracle.c fixes (the reference's own order is a race) -- so results are the oracle's bit for bit,
// and because a frame total does not depend on the running sum, frames of one pixel can be traced by different lanes
// (f3d_wavefront.hip) and folded afterwards.
// Instanced meshes are walked through the threaded BVH of f3d_bvh.h (one 32-byte record per visited node, no stack)
// with the reference's two triangle tests: watertight for closest hits (pt_intersect.wgsl:113-178), Moller-Trumbore
// for shadow rays (pt_shadow.wgsl:205-236); equal-t hits resolve to the lowest triangle index (the oracle's sweep).
//
// Terrain primitive (NOT in the reference: its wavefront tracer traces spheres and instanced meshes only,
// pt_intersect.wgsl:431-558, and its terrain tracer shades one bounce; BASELINE.json configs[2] asks for both, "GI" over
// a DEM): an optional heightfield whose closest / any hit is the terrain tracer's `terrain_trace`
// (hybrid_terrain_traversal.wgsl:254-372) -- here the stackless min-max march of f3d_march.h on the same tables, with the
// same results -- entered into closest() after spheres and meshes with the closest t so far as its tmax, and into
// shadowed() as one more any-hit test.  The hit carries the bilinear patch's normal and the scene's terrain material slot.
//
// Not here (off / empty in render_pt_reference): ReSTIR guiding, fog medium, hair segments.
// Numerics: f3d_math.h contract (no contraction; dot = fma chain; fixed-polynomial sincos/atan/exp/log).
#pragma once

#include "f3d_march.h"
#include "f3d_math.h"
#include "f3d_scene.h"

namespace f3d {
namespace wf {

struct SphereDev {
    V3 c;
    float r;
};
struct MaterialDev {  // Sphere's material half (pt_shade.wgsl:235-245) + object_importance[mat]
    V3 albedo;
    float metallic;
    V3 emissive;
    float roughness;
    float ior, ax, ay, importance;
};
struct BlasDev {
    const BvhNode *nodes;
    const float4 *tris;  // 3 float4 per triangle in leaf order, v0.w = original triangle index
    uint32_t node_count, pad;
};
struct InstanceDev {
    float w2o[16];  // world_to_object, column-major
    uint32_t blas, material, pad0, pad1;
};
struct HairDev {  // HairSegment, pt_intersect.wgsl:60-69
    V3 p0;
    float r0;
    V3 p1;
    float r1;
    uint32_t material, pad0, pad1, pad2;
};
struct DirLightDev {
    V3 wi;  // normalize(-direction)
    float importance;
    V3 Li;  // color * intensity
    float pad;
};
struct AreaLightDev {
    V3 position;
    float rad;  // max(radius, 1e-6)
    V3 nL;
    float importance;
    V3 tL;
    float p_area;
    V3 bL;
    float pad;
    V3 Li;
    float pad2;
};
struct SceneDev {
    const SphereDev *spheres;
    const MaterialDev *mats;
    const BlasDev *blas;
    const InstanceDev *inst;
    const DirLightDev *dir;
    const AreaLightDev *area;
    uint32_t sphere_count, blas_count, inst_count, dir_count, area_count;
    float dir_sum_imp, area_sum_imp;
    V3 env_ground, env_sky, miss_ground, miss_sky;
    V3 cam_origin, cam_right, cam_up, cam_neg_forward;
    float half_w, half_h;
    uint32_t width, height, seed_hi, seed_lo;
    // optional heightfield primitive (see the header): tables as the terrain tracer builds them, material slot of its hits
    TerrainDev terrain;
    uint32_t has_terrain, terrain_mat;
    const uint2 *primary_start;  // per pixel {t_clear bits, level}: the terrain tracer's primary-ray certificates (f3d_cone.h); nullptr: none
    // hair strands (closest hits only) and the homogeneous fog (pt_shade.wgsl:328-338, :500-520)
    const HairDev *hair;
    uint32_t hair_count;
    float medium_mu;      // max(sigma_t * density, 0)
    uint32_t medium_on;
};

constexpr float kTwoPiInv = 0.15915494309189533577f;

F3D_HD float f_saturate(float x) { return f_clamp(x, 0.0f, 1.0f); }
F3D_HD V3 scale3(V3 a, float s) { return a * s; }
F3D_HD float comp3(V3 a, uint32_t k) { return k == 0u ? a.x : (k == 1u ? a.y : a.z); }
F3D_HD V3 mix3(V3 a, V3 b, float t) { return V3{mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t)}; }
F3D_HD V3 reflect3(V3 i, V3 n) { return i - n * (2.0f * dot(n, i)); }
F3D_HD float pow5(float x) {
    const float x2 = x * x;
    return (x2 * x2) * x;
}
F3D_HD float pow16(float x) {
    float a = x * x;
    a = a * a;
    a = a * a;
    return a * a;
}
F3D_HD void sincos_rad(float a, float &s, float &c) {
    float u = a * kTwoPiInv;
    u = u - f_floor(u);
    sincos_turn(u, s, c);
}
F3D_HD uint32_t splitmix32(uint32_t x) {  // adjudication.rs:222-228
    x += 0x9E3779B9u;
    uint32_t z = x;
    z = (z ^ (z >> 16)) * 0x21F0AAADu;
    z = (z ^ (z >> 15)) * 0x735A2D97u;
    return z ^ (z >> 15);
}

// ---- camera ray, pt_raygen.wgsl:88-226 (spp 1 per frame, Van der Corput / Halton-3 + Cranley-Patterson + tent) ----
F3D_HD float tent(float u) { return u < 0.5f ? f_sqrt(2.0f * u) - 1.0f : 1.0f - f_sqrt(2.0f * (1.0f - u)); }
F3D_HD float halton3(uint32_t i) {
    float f = 1.0f, r = 0.0f;
    while (i != 0u) {
        f = f / 3.0f;
        r = r + (float)(i % 3u) * f;
        i = i / 3u;
    }
    return r;
}
F3D_HD float rotate01(float u, float r) {
    const float x = u + r;
    return x - f_floor(x);
}

struct PathState {
    V3 o, d, thr;
    float tmin;
    uint32_t depth, rng_hi;
};

F3D_HD void camera_ray(const SceneDev &S, uint32_t px, uint32_t py, uint32_t frame, uint32_t seed_hi, uint32_t seed_lo, PathState &P) {
    const uint32_t pixel = py * S.width + px;
#if defined(__HIP_DEVICE_COMPILE__)
    const float u1 = (float)__brev(frame) * 2.3283064365386963e-10f;  // radical_inverse_vdc: a bit reversal
#else
    uint32_t n = frame;
    n = (n << 16) | (n >> 16);
    n = ((n & 0x55555555u) << 1) | ((n & 0xAAAAAAAAu) >> 1);
    n = ((n & 0x33333333u) << 2) | ((n & 0xCCCCCCCCu) >> 2);
    n = ((n & 0x0F0F0F0Fu) << 4) | ((n & 0xF0F0F0F0u) >> 4);
    n = ((n & 0x00FF00FFu) << 8) | ((n & 0xFF00FF00u) >> 8);
    const float u1 = (float)n * 2.3283064365386963e-10f;
#endif
    const float u2 = halton3(frame);
    uint32_t rot = seed_lo ^ (px * 9781u) ^ (py * 6271u) ^ (seed_hi * 13007u);
    const float r1 = rng_next(rot), r2 = rng_next(rot);
    const float jx = tent(rotate01(u1, r1)) * 0.5f, jy = tent(rotate01(u2, r2)) * 0.5f;
    const float ndc_x = ((((float)px + 0.5f) + jx) / (float)S.width) * 2.0f - 1.0f;
    const float ndc_y = (1.0f - (((float)py + 0.5f) + jy) / (float)S.height) * 2.0f - 1.0f;
    const V3 c = normalize(V3{ndc_x * S.half_w, ndc_y * S.half_h, -1.0f});
    const V3 R = S.cam_right, U = S.cam_up, F = S.cam_neg_forward;
    P.d = normalize(V3{(c.x * R.x + c.y * U.x) + c.z * F.x, (c.x * R.y + c.y * U.y) + c.z * F.y, (c.x * R.z + c.y * U.z) + c.z * F.z});
    P.o = S.cam_origin;
    P.tmin = 1e-4f;
    P.thr = V3{1.0f, 1.0f, 1.0f};
    P.depth = 0u;
    P.rng_hi = seed_hi ^ (pixel * 9781u) ^ (frame * 6271u);
}

// ---- geometry -----------------------------------------------------------------------------------------------------
F3D_HD V3 xf_point(const float *m, V3 p) {
    return V3{((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12] * 1.0f, ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13] * 1.0f,
              ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14] * 1.0f};
}
F3D_HD V3 xf_vector(const float *m, V3 v) {
    return V3{((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * 0.0f, ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * 0.0f,
              ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * 0.0f};
}
F3D_HD V3 xf_normal(const float *m, V3 n) {  // transpose(world_to_object) * (n, 0), normalised
    return normalize(V3{((m[0] * n.x + m[1] * n.y) + m[2] * n.z) + m[3] * 0.0f, ((m[4] * n.x + m[5] * n.y) + m[6] * n.z) + m[7] * 0.0f,
                        ((m[8] * n.x + m[9] * n.y) + m[10] * n.z) + m[11] * 0.0f});
}

// watertight test, pt_intersect.wgsl:113-178 (t only; the normal is taken from the winning triangle afterwards)
F3D_HD bool tri_watertight(V3 o, V3 d, float tmin, float tmax, V3 v0, V3 v1, V3 v2, float &t_out) {
    const V3 A = v0 - o, B = v1 - o, C = v2 - o;
    const float adx = f_abs(d.x), ady = f_abs(d.y), adz = f_abs(d.z);
    uint32_t kz = 2u, kx = 0u, ky = 1u;
    if (adx > ady && adx > adz) {
        kz = 0u, kx = 1u, ky = 2u;
    } else if (ady > adz) {
        kz = 1u, kx = 2u, ky = 0u;
    }
    const float Sz = 1.0f / comp3(d, kz), Sx = comp3(d, kx) * Sz, Sy = comp3(d, ky) * Sz;
    const float ax = comp3(A, kx) - Sx * comp3(A, kz), ay = comp3(A, ky) - Sy * comp3(A, kz);
    const float bx = comp3(B, kx) - Sx * comp3(B, kz), by = comp3(B, ky) - Sy * comp3(B, kz);
    const float cx = comp3(C, kx) - Sx * comp3(C, kz), cy = comp3(C, ky) - Sy * comp3(C, kz);
    const float az = comp3(A, kz) * Sz, bz = comp3(B, kz) * Sz, cz = comp3(C, kz) * Sz;
    const float U = (bx * cy) - (by * cx), V = (cx * ay) - (cy * ax), W = (ax * by) - (ay * bx);
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = (U + V) + W;
    if (det == 0.0f) return false;
    const float T = (U * az + V * bz) + W * cz;
    const float t = T / det;
    if (t > tmin && t < tmax) {
        t_out = t;
        return true;
    }
    return false;
}
// Moller-Trumbore any-hit, pt_shadow.wgsl:205-236
F3D_HD bool tri_shadow(V3 ro, V3 rd, float tmin, float tmax, V3 v0, V3 v1, V3 v2) {
    const V3 e1 = v1 - v0, e2 = v2 - v0;
    const V3 h = cross(rd, e2);
    const float a = dot(e1, h);
    if (f_abs(a) < 1e-7f) return false;
    const float f = 1.0f / a;
    const V3 s = ro - v0;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    const V3 q = cross(s, e1);
    const float v = f * dot(rd, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = f * dot(e2, q);
    return t > tmin && t < tmax;
}

// Threaded-BVH walk of one BLAS.  ANY: Moller-Trumbore, stop at the first hit.  Otherwise: watertight test, the
// smallest t wins and the lowest original index among equal t (what a sweep in index order with `t < best` returns).
template <bool ANY>
F3D_HD bool walk_blas(const BlasDev &M, V3 o, V3 d, float tmin, float tmax, float &t_best, V3 &n_best) {
    const float ix = (d.x < 0.0f ? -1.0f : 1.0f) / f_max(f_abs(d.x), 1e-12f);
    const float iy = (d.y < 0.0f ? -1.0f : 1.0f) / f_max(f_abs(d.y), 1e-12f);
    const float iz = (d.z < 0.0f ? -1.0f : 1.0f) / f_max(f_abs(d.z), 1e-12f);
    const float4 *nodes = reinterpret_cast<const float4 *>(M.nodes);
    bool any = false;
    uint32_t best_tri = 0xFFFFFFFFu, best_slot = 0u;
    t_best = tmax;
    uint32_t node = 0u;
    while (node < M.node_count) {
        const float4 lo = nodes[2u * node], hi = nodes[2u * node + 1u];
        const float ax = (lo.x - o.x) * ix, bx = (hi.x - o.x) * ix;
        const float ay = (lo.y - o.y) * iy, by = (hi.y - o.y) * iy;
        const float az = (lo.z - o.z) * iz, bz = (hi.z - o.z) * iz;
        const float enter = f_max(f_max(f_min(ax, bx), f_min(ay, by)), f_max(f_min(az, bz), tmin));
        const float exit = f_min(f_min(f_max(ax, bx), f_max(ay, by)), f_min(f_max(az, bz), t_best));
        if (!(enter <= exit * 1.00001f + 1e-6f)) {  // conservative: boxes are padded, ties are kept
            node = f_bits(lo.w);
            continue;
        }
        const uint32_t leaf = f_bits(hi.w);
        if (leaf == 0u) {
            node = node + 1u;
            continue;
        }
        const uint32_t first = leaf >> 3, count = leaf & 7u;
        for (uint32_t k = 0u; k < count; k++) {
            const float4 a = M.tris[3u * (first + k)], b = M.tris[3u * (first + k) + 1u], c = M.tris[3u * (first + k) + 2u];
            const V3 v0{a.x, a.y, a.z}, v1{b.x, b.y, b.z}, v2{c.x, c.y, c.z};
            if (ANY) {
                if (tri_shadow(o, d, tmin, tmax, v0, v1, v2)) return true;
            } else {
                float t;
                if (tri_watertight(o, d, tmin, tmax, v0, v1, v2, t)) {
                    const uint32_t tri = f_bits(a.w);
                    if (t < t_best || (t == t_best && tri < best_tri)) {
                        t_best = t;
                        best_tri = tri;
                        best_slot = first + k;
                        any = true;
                    }
                }
            }
        }
        node = f_bits(lo.w);
    }
    if (!ANY && any) {
        const float4 a = M.tris[3u * best_slot], b = M.tris[3u * best_slot + 1u], c = M.tris[3u * best_slot + 2u];
        const V3 v0{a.x, a.y, a.z};
        n_best = normalize(cross(V3{b.x, b.y, b.z} - v0, V3{c.x, c.y, c.z} - v0));
    }
    return any;
}

struct SurfaceHitWf {
    V3 p, n;
    float t;
    uint32_t mat;
    bool hair;   // Hit.flags bit 0
    V3 tangent;  // strand axis (hair hits only)
};

// ray_cylinder_segment, pt_intersect.wgsl:21-57: the open cylinder of radius r around p0 p1
F3D_HD bool hair_segment(V3 o, V3 d, float tmin, float tmax, V3 p0, V3 p1, float r, float &t_out, V3 &n_out) {
    const V3 axis = p1 - p0;
    const float L = f_sqrt(dot(axis, axis));
    if (L < 1e-6f || r <= 0.0f) return false;
    const V3 n{axis.x / L, axis.y / L, axis.z / L};
    const V3 w0 = o - p0;
    const float d_par = dot(d, n);
    const V3 d_perp = d - n * d_par, w_perp = w0 - n * dot(w0, n);
    const float A = dot(d_perp, d_perp), B = 2.0f * dot(d_perp, w_perp), C = dot(w_perp, w_perp) - r * r;
    if (A < 1e-12f) return false;
    const float disc = B * B - (4.0f * A) * C;
    if (disc < 0.0f) return false;
    const float sdisc = f_sqrt(f_max(disc, 0.0f));
    const float t0 = (-B - sdisc) / (2.0f * A), t1 = (-B + sdisc) / (2.0f * A);
    float thit = 1e30f;
    if (t0 > tmin && t0 < tmax) thit = t0;
    if (t1 > tmin && t1 < thit) thit = t1;
    if (thit >= 1e20f) return false;
    const float along_axis = dot(w0 + d * thit, n);
    if (along_axis < 0.0f || along_axis > L) return false;
    t_out = thit;
    n_out = normalize(w_perp + d_perp * thit);
    return true;
}

// pt_intersect.wgsl main, :431-558 (+ the terrain primitive)
// camera_pixel: the pixel whose CAMERA ray this is (its march may start where the pixel's certificate ends); kNoPixel: any other ray
constexpr uint32_t kNoPixel = 0xFFFFFFFFu;
// have: the lane has a ray.  trace_frames calls with EVERY lane of the wave in a scene whose primitives are the heightfield and
// spheres: a lane without a ray goes through the heightfield's march with an empty one (tmax < tmin: it never marches) and is,
// inside, a lane whose ray has ended -- what the sharing of closest-hit rays deals the last rays of a wave to (f3d_march.h
// march_shared_closest).
template <class Wave>
F3D_HD bool closest(const SceneDev &S, V3 o, V3 d, float tmin, SurfaceHitWf &H, Wave &wave, uint32_t camera_pixel = kNoPixel, bool have = true) {
    const float tmax = 1e30f;
    float t_best = 1e30f;
    V3 n = V3{0.0f, 1.0f, 0.0f};
    uint32_t mat = 0u;
    for (uint32_t i = 0u; have && i < S.sphere_count; i++) {
        const SphereDev s = S.spheres[i];
        const V3 oc = o - s.c;
        const float b = dot(oc, d);
        const float cterm = dot(oc, oc) - s.r * s.r;
        const float disc = b * b - cterm;
        float t = 1e30f;
        if (disc > 0.0f) {
            const float q = f_sqrt(disc);
            const float t0 = -b - q, t1 = -b + q;
            t = t0 > 1e-3f ? t0 : (t1 > 1e-3f ? t1 : 1e30f);
        }
        if (t >= tmin && t < f_min(t_best, tmax)) {
            t_best = t;
            const V3 hp = o + d * t;
            n = normalize(hp - s.c);
            mat = i;
        }
    }
    if (Wave::kLite) {
        // (a scene without meshes, hair, area lights and fog -- render_terrain_gi's -- runs an instantiation without their code)
    } else if (S.inst_count == 0u) {
        float t;
        V3 nn;
        if (S.blas_count > 0u && walk_blas<false>(S.blas[0], o, d, tmin, tmax, t, nn) && t < t_best) {
            t_best = t;
            n = nn;
            mat = 0u;
        }
    } else {
        for (uint32_t ii = 0u; ii < S.inst_count; ii++) {
            const InstanceDev &I = S.inst[ii];
            const V3 oo = xf_point(I.w2o, o);
            const V3 dd = normalize(xf_vector(I.w2o, d));
            float t;
            V3 nn;
            if (walk_blas<false>(S.blas[I.blas], oo, dd, tmin, tmax, t, nn) && t < t_best) {
                t_best = t;
                n = xf_normal(I.w2o, nn);
                mat = I.material;
            }
        }
    }
    bool hair = false;
    V3 tangent{0.0f, 0.0f, 0.0f};
    for (uint32_t i = 0u; !Wave::kLite && i < S.hair_count; i++) {  // hair strands, :499-521 (after spheres and meshes, like the reference)
        const HairDev seg = S.hair[i];
        float t;
        V3 nn;
        if (hair_segment(o, d, tmin, tmax, seg.p0, seg.p1, f_max(0.0f, (0.5f * (seg.r0 + seg.r1)) * 1.0f), t, nn) && t < t_best) {
            t_best = t;
            n = nn;
            mat = S.sphere_count > 0u ? (seg.material < S.sphere_count - 1u ? seg.material : S.sphere_count - 1u) : 0u;
            hair = true;
            tangent = normalize(seg.p1 - seg.p0);
        }
    }
    if (Wave::kTerrain && S.has_terrain != 0u) {  // terrain_trace(ray with tmax = the closest hit so far), curvature off
        const RayCtx r = make_ray(S.terrain, o, have ? tmin : 1.0f, d, have ? t_best : 0.0f, false);
        // A camera ray of a pixel with a certificate (round 5: the terrain tracer's, for the same camera -- f3d_cone.h
        // primary_start) starts where the certificate ends: every node before that is one the march would step over without
        // solving a leaf, so the hit is the same; 27 -> 14.5 steps per camera ray on the headline frame.
        MarchState m = march_begin(S.terrain, r, true);
        if (have && camera_pixel != kNoPixel && S.primary_start != nullptr) {
            const uint2 st = S.primary_start[camera_pixel];
            if (f_from_bits(st.x) > 0.0f) m = march_begin_at(S.terrain, r, f_from_bits(st.x), st.y);
        }
        const TraceHit th = march_terrain_from<false, true>(S.terrain, r, false, m, *wave.pend);
        if (th.hit && th.t < t_best) {
            t_best = th.t;
            n = th.n;
            mat = S.terrain_mat;
            hair = false;
        }
    }
    if (!have || !(t_best < 1e20f)) return false;
    H.p = o + d * t_best;
    H.t = t_best;
    H.n = n;
    H.mat = mat;
    H.hair = hair;
    H.tangent = tangent;
    return true;
}

// the spheres' part of pt_shadow.wgsl main (:248-294): any sphere hit in (tmin, tmax)
F3D_HD bool sphere_shadow(const SceneDev &S, V3 ro, V3 rd, float tmin, float tmax) {
    for (uint32_t i = 0u; i < S.sphere_count; i++) {
        const SphereDev s = S.spheres[i];
        const V3 oc = ro - s.c;
        const float b = dot(oc, rd);
        const float cterm = dot(oc, oc) - s.r * s.r;
        const float disc = b * b - cterm;
        if (disc <= 0.0f) continue;
        const float q = f_sqrt(disc);
        const float t0 = -b - q, t1 = -b + q;
        if ((t0 > tmin && t0 < tmax) || (t1 > tmin && t1 < tmax)) return true;
    }
    return false;
}

// pt_shadow.wgsl main, :248-294 (+ the terrain primitive)
template <class Wave>
F3D_HD bool shadowed(const SceneDev &S, V3 ro, V3 rd, float tmin, float tmax, Wave &wave) {
    if (Wave::kTerrain && S.has_terrain != 0u) {  // any hit of the heightfield in (tmin, tmax)
        const RayCtx r = make_ray(S.terrain, ro, tmin, rd, tmax, false);
        if (march_terrain<false>(S.terrain, r, true, true, *wave.pend).hit) return true;
    }
    if (sphere_shadow(S, ro, rd, tmin, tmax)) return true;
    if (Wave::kLite) return false;
    float t;
    V3 nn;
    if (S.inst_count == 0u) return S.blas_count > 0u && walk_blas<true>(S.blas[0], ro, rd, tmin, tmax, t, nn);
    for (uint32_t ii = 0u; ii < S.inst_count; ii++) {
        const InstanceDev &I = S.inst[ii];
        if (walk_blas<true>(S.blas[I.blas], xf_point(I.w2o, ro), normalize(xf_vector(I.w2o, rd)), tmin, tmax, t, nn)) return true;
    }
    return false;
}

// ---- BSDF and samplers, pt_shade.wgsl:43-455 ------------------------------------------------------------------------
struct Frame3 {
    V3 t, b, n;
};
F3D_HD Frame3 tangent_frame(V3 n) {  // make_tangent_basis, :351-360
    const float sign = n.z < 0.0f ? -1.0f : 1.0f;
    const float a = -1.0f / (sign + n.z);
    const float b = (n.x * n.y) * a;
    return Frame3{V3{1.0f + ((sign * n.x) * n.x) * a, sign * b, -sign * n.x}, V3{b, sign + (n.y * n.y) * a, -n.y}, n};
}
F3D_HD V3 to_world(const Frame3 &m, V3 v) {
    return V3{(m.t.x * v.x + m.b.x * v.y) + m.n.x * v.z, (m.t.y * v.x + m.b.y * v.y) + m.n.y * v.z,
              (m.t.z * v.x + m.b.z * v.y) + m.n.z * v.z};
}
F3D_HD V3 cosine_hemisphere(float u1, float u2) {
    const float r = f_sqrt(u1);
    float s, c;
    sincos_turn(u2, s, c);
    return V3{r * c, r * s, f_sqrt(f_max(0.0f, 1.0f - u1))};
}
F3D_HD V3 schlick(float cos_theta, V3 F0) {
    const float w = pow5(1.0f - f_saturate(cos_theta));
    return V3{F0.x + (1.0f - F0.x) * w, F0.y + (1.0f - F0.y) * w, F0.z + (1.0f - F0.z) * w};
}
F3D_HD float ggx_d(float n_dot_h, float alpha) {
    const float a2 = alpha * alpha, ndh2 = n_dot_h * n_dot_h;
    const float q = ndh2 * (a2 - 1.0f) + 1.0f;
    return a2 / f_max(kPi * (q * q), 1e-6f);
}
F3D_HD float smith_g1(float n_dot_v, float alpha) {
    const float a1 = alpha + 1.0f;
    const float k = (a1 * a1) / 8.0f;
    return n_dot_v / (n_dot_v * (1.0f - k) + k);
}
F3D_HD float ggx_d_aniso(V3 h, V3 t, V3 b, V3 n, float ax, float ay) {
    const float hx = dot(h, t), hy = dot(h, b), hz = f_max(dot(h, n), 0.0f);
    const float x2 = (hx * hx) / (ax * ax + 1e-8f), y2 = (hy * hy) / (ay * ay + 1e-8f);
    const float denom = (x2 + y2) + hz * hz;
    return 1.0f / f_max((((kPi * ax) * ay) * denom) * denom, 1e-6f);
}
F3D_HD float smith_g1_aniso(V3 v, V3 t, V3 b, V3 n, float ax, float ay) {
    const float vx = dot(v, t), vy = dot(v, b), vz = f_max(dot(v, n), 0.0f);
    const float alpha_v = f_sqrt((vx * vx) * (ax * ax) + (vy * vy) * (ay * ay)) / f_max(vz, 1e-6f);
    return 2.0f / (1.0f + f_sqrt(1.0f + alpha_v * alpha_v));
}

struct MatCtx {  // a hit's material, unpacked once
    V3 albedo, F0;
    float metallic, roughness, ax, ay, imp;
    bool aniso;
};
struct Bsdf {
    V3 f;
    float pdf;
};
// bsdf_eval_pdf, :43-100 (the anisotropic branch reads the frame as ROWS of the basis matrix, like the reference)
F3D_HD Bsdf bsdf_eval(const MatCtx &M, V3 wo, V3 wi, V3 n) {
    const float n_dot_l = f_max(dot(n, wi), 0.0f), n_dot_v = f_max(dot(n, wo), 0.0f);
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return Bsdf{V3{0.0f, 0.0f, 0.0f}, 0.0f};
    const float kd = f_saturate(1.0f - M.metallic);
    const V3 fd = V3{M.albedo.x / kPi, M.albedo.y / kPi, M.albedo.z / kPi} * kd;
    const float pdf_d = n_dot_l / kPi;
    const float m = f_max(0.02f, M.roughness * M.roughness);
    const V3 h = normalize(wi + wo);
    const float n_dot_h = f_max(dot(n, h), 0.0f), v_dot_h = f_max(dot(wo, h), 0.0f);
    float D, G;
    if (!M.aniso) {
        D = ggx_d(n_dot_h, m);
        G = smith_g1(n_dot_l, m) * smith_g1(n_dot_v, m);
    } else {
        const Frame3 bs = tangent_frame(n);
        const V3 t{bs.t.x, bs.b.x, bs.n.x}, bb{bs.t.y, bs.b.y, bs.n.y}, nn{bs.t.z, bs.b.z, bs.n.z};
        D = ggx_d_aniso(h, t, bb, nn, M.ax, M.ay);
        G = smith_g1_aniso(wi, t, bb, nn, M.ax, M.ay) * smith_g1_aniso(wo, t, bb, nn, M.ax, M.ay);
    }
    const V3 F = schlick(v_dot_h, M.F0);
    const float spec = (D * G) / f_max((4.0f * n_dot_l) * n_dot_v, 1e-6f);
    const float pdf_s = (D * n_dot_h) / f_max(4.0f * v_dot_h, 1e-6f);
    const float ks = 1.0f - kd;
    return Bsdf{fd + F * spec, f_max(kd * pdf_d + ks * pdf_s, 1e-8f)};
}
F3D_HD float up_lobe_pdf(V3 w) {  // power_cosine_pdf_about_up, m = 16, :165-169
    const float c = f_max(dot(V3{0.0f, 1.0f, 0.0f}, normalize(w)), 0.0f);
    return ((16.0f + 1.0f) * pow16(c)) / (2.0f * kPi);
}

// importance-weighted pick (:622-633, :667-678); `imp(i)` = max(importance_i, 0)
template <class Imp>
F3D_HD uint32_t pick(uint32_t count, float sum_imp, uint32_t &rng, Imp imp) {
    uint32_t idx = 0u;
    if (sum_imp > 0.0f) {
        const float rsel = rng_next(rng) * sum_imp;
        float acc = 0.0f;
        for (uint32_t i = 0u; i < count; i++) {
            acc = acc + imp(i);
            if (rsel <= acc) {
                idx = i;
                break;
            }
        }
    } else {
        uint32_t c = count;
        F3D_OPAQUE_UNIFORM(c);  // (a scene constant: its float is formed here, not kept in a vector register across the flat loop)
        idx = sat_u32(f_floor(rng_next(rng) * (float)c));
    }
    uint32_t last = count;
    F3D_OPAQUE_UNIFORM(last);
    last = last - 1u;
    return idx < last ? idx : last;
}

// ---- a lane's loop-carried path state (round 6: resident in the lane's LDS column) -------------------------------------------
// A march of the heightfield primitive needs ~70 of the 80 registers six waves a SIMD leave a lane.  Whatever the flat loop
// carries ACROSS a march therefore lived in scratch: 284 bytes a lane in round 4 (15 GB written per 1080p x 32-path launch),
// 168 in round 5, which parked the frame's total and the throughput in eight rows of the lane's LDS column.  Round 6 gives the
// PBR tracer's translation unit one-word leaf-FIFO entries (f3d_march.h F3D_FIFO_WORDS: the drain forms the interval again) --
// 5 FIFO rows instead of 15 in the same 6 400-byte block -- and keeps ALL the loop-carried state in the 18 rows that frees:
// every use reads its row, every change writes it, so no lane of a wave holds path state in a register while any other lane
// marches (a value that is only parked on the marching lanes' path stays allocated for the lanes that wait).
//   rows 0-2   the frame's running total              rows 3-5   the path throughput
//   row  6     one word: depth (bits 0-4), "the next vertex starts a new frame" (5), "a hit waits for its shading" (6),
//              frame - first (7-17: at most 1 023 frames a call), closest-hit queries made (18-31)
//   row  7     the path's RNG word
//   rows 8-10  the ray's direction: the camera ray or the continuation the last vertex sampled
//   rows 11-13 A: the origin of a continuing ray until closest() has taken it | the waiting hit's position | across a vertex's
//              shadow rays the environment sample's contribution
//   rows 14-16 B: the waiting hit's normal | across the shadow rays the directional light's contribution
//   row  17    the waiting hit's material
// Same values, same operations in the same order as the register form (Wave::kPark == 0: the kernel without the heightfield
// primitive, which has no LDS block and 128 registers): results are unchanged (tests/test_wavefront.py, test_offline_gi.py; the
// host emulator runs the row form with an array for the rows).
template <class Wave>
F3D_HD void park3(const Wave &w, uint32_t row, V3 v) {
    w.park(row, v.x);
    w.park(row + 1u, v.y);
    w.park(row + 2u, v.z);
}
template <class Wave>
F3D_HD V3 unpark3(const Wave &w, uint32_t row) {
    return V3{w.unpark(row), w.unpark(row + 1u), w.unpark(row + 2u)};
}
constexpr uint32_t kLaneRows = 18u;
constexpr uint32_t kMaxFramesPerCall = 1023u;  // (frame - first) has 11 bits and the query count 14: 16 queries a frame at most
template <class Wave>
struct LaneState {
    static constexpr bool kRows = Wave::kPark >= kLaneRows;
    static constexpr uint32_t kAcc = 0u, kThr = 3u, kWord = 6u, kRng = 7u, kDir = 8u, kA = 11u, kB = 14u, kMat = 17u;
    static constexpr uint32_t kDepthMask = 31u, kFresh = 32u, kPending = 64u, kFrameShift = 7u, kFrameMask = 2047u, kQueryShift = 18u;
    const Wave &w;
    V3 acc_{0.0f, 0.0f, 0.0f}, thr_{0.0f, 0.0f, 0.0f}, dir_{0.0f, 0.0f, 0.0f}, a_{0.0f, 0.0f, 0.0f}, b_{0.0f, 0.0f, 0.0f};
    uint32_t word_ = 0u, rng_ = 0u, mat_ = 0u;
    F3D_HD explicit LaneState(const Wave &wave) : w(wave) {}
    F3D_HD V3 get3(uint32_t row, const V3 &reg) const { return kRows ? unpark3(w, row) : reg; }
    F3D_HD void set3(uint32_t row, V3 &reg, V3 v) {
        if (kRows) park3(w, row, v);
        else reg = v;
    }
    F3D_HD V3 acc() const { return get3(kAcc, acc_); }
    F3D_HD void set_acc(V3 v) { set3(kAcc, acc_, v); }
    F3D_HD void add(V3 c) { set_acc(acc() + c); }
    F3D_HD V3 thr() const { return get3(kThr, thr_); }
    F3D_HD void set_thr(V3 v) { set3(kThr, thr_, v); }
    F3D_HD V3 dir() const { return get3(kDir, dir_); }
    F3D_HD void set_dir(V3 v) { set3(kDir, dir_, v); }
    F3D_HD V3 a() const { return get3(kA, a_); }
    F3D_HD void set_a(V3 v) { set3(kA, a_, v); }
    F3D_HD V3 b() const { return get3(kB, b_); }
    F3D_HD void set_b(V3 v) { set3(kB, b_, v); }
    F3D_HD uint32_t word() const { return kRows ? f_bits(w.unpark(kWord)) : word_; }
    F3D_HD void set_word(uint32_t v) {
        if (kRows) w.park(kWord, f_from_bits(v));
        else word_ = v;
    }
    F3D_HD uint32_t rng() const { return kRows ? f_bits(w.unpark(kRng)) : rng_; }
    F3D_HD void set_rng(uint32_t v) {
        if (kRows) w.park(kRng, f_from_bits(v));
        else rng_ = v;
    }
    F3D_HD uint32_t mat() const { return kRows ? f_bits(w.unpark(kMat)) : mat_; }
    F3D_HD void set_mat(uint32_t v) {
        if (kRows) w.park(kMat, f_from_bits(v));
        else mat_ = v;
    }
};

#if defined(F3D_WF_SHADOW_STREAM)
template <class Wave>
struct NeeShadowSource {  // the shadow rays of one vertex as a source of march_stream (-DF3D_WF_SHADOW_STREAM, see surface_vertex)
    const SceneDev &S;
    LaneState<Wave> &lane;
    V3 so, env_wi;
    uint32_t dir_light, nee_on;
    F3D_HD bool sphere_blocks(V3 rd) const { return sphere_shadow(S, so, rd, 1e-3f, 1e30f); }
    template <class Ctx>
    F3D_HD bool refill(bool &have, RayCtx &r, float &t_stop, uint32_t &tag, Ctx &ctx) {
        while (!have && nee_on != 0u) {
            const uint32_t k = (uint32_t)__builtin_ctz(nee_on);
            nee_on &= nee_on - 1u;
            const V3 wi = k == 0u ? env_wi : S.dir[dir_light].wi;
            if (sphere_blocks(wi)) continue;  // occluded whatever the terrain says: the contribution is dropped
            r = make_ray(S.terrain, so, 1e-3f, wi, 1e30f, false);
            t_stop = 3.0e38f;
            tag = k;
            have = true;
        }
        return ctx.any(nee_on != 0u);
    }
    F3D_HD void verdict(uint32_t tag, bool blocked) {
        if (!blocked) lane.add(tag == 0u ? lane.a() : lane.b());
    }
};

#endif
// One surface vertex: emission, NEE (environment / directional / area) with its shadow rays, continuation sample,
// roulette (pt_shade.wgsl main :460-862 + pt_shadow.wgsl main).  Returns true when the path continues in P.
// H: the hit (position, normal, material -- and, scenes with hair or fog, its distance, hair flag and strand axis);
// d_in: the direction the path arrived with.  Everything else the vertex reads and leaves is the lane's state L: it adds to
// the frame's total, and when the path continues it leaves the new throughput, direction, RNG word and depth there and the
// new ray's origin in rows A.  (`lane`, not L: the light records of the next-event blocks are called L.)
// The vertex's shadow rays are an OUTPUT (NeeRays): trace_frames traces them (vertex_shadows), in a scene whose occluders are the
// heightfield and spheres with EVERY lane of the wave in the call, so that lanes without a ray lend themselves to the ray sharing.
struct NeeRays {
    V3 so{0.0f, 0.0f, 0.0f}, env_wi{0.0f, 0.0f, 0.0f}, area_wi{0.0f, 0.0f, 0.0f}, area_c{0.0f, 0.0f, 0.0f};
    float area_tmax = 1e30f;
    uint32_t on = 0u, dir_light = 0u;  // on: bit 0 environment (contribution in rows A), bit 1 directional (rows B), bit 2 area (area_c)
};
template <class Wave>
F3D_HD bool vertex_shade(const SceneDev &S, uint32_t frame, const SurfaceHitWf &H, V3 d_in, LaneState<Wave> &lane, Wave &wave, NeeRays &nee) {
    using Lane = LaneState<Wave>;
    const uint32_t pixel = wave.pixel();
    const uint32_t depth_in = lane.word() & Lane::kDepthMask;
    const MaterialDev md = S.mats[H.mat];
    MatCtx M;
    M.albedo = md.albedo;
    M.metallic = md.metallic;
    M.roughness = md.roughness;
    M.ax = f_max(0.002f, md.ax);
    M.ay = f_max(0.002f, md.ay);
    M.aniso = !(f_abs(M.ax - M.ay) < 1e-4f);
    M.imp = md.importance;
    const float sm = f_saturate(md.metallic);
    M.F0 = V3{mix(0.04f, md.albedo.x, sm), mix(0.04f, md.albedo.y, sm), mix(0.04f, md.albedo.z, sm)};
    auto acc_add = [&](V3 c) F3D_LAMBDA { lane.add(c); };
    const V3 thr_in = lane.thr();  // the throughput the path arrives with
    if (md.emissive.x > 0.0f || md.emissive.y > 0.0f || md.emissive.z > 0.0f) acc_add(thr_in * md.emissive);

    uint32_t rng = lane.rng() ^ (pixel * 26699u) ^ (frame * 30977u);
    const V3 n = normalize(H.n), wo = normalize(normalize(neg(d_in)));
    const float n_dot_v = f_max(dot(n, wo), 0.0f);
    const Frame3 basis = tangent_frame(n);
    const V3 so = H.p + n * 1e-3f;
    // homogeneous fog, :328-338 / :500-520: next-event contributions of this vertex are attenuated over the segment that
    // reached it; a PRIMARY hit also adds the environment seen through the fog it looks through
    const float mtrans = (!Wave::kLite && S.medium_on != 0u) ? exp_det(-f_max(H.t, 0.0f) * S.medium_mu) : 1.0f;
    if (!Wave::kLite && S.medium_on != 0u && depth_in == 0u) {
        const V3 back = neg(wo);
        acc_add(mix3(S.env_ground, S.env_sky, 0.5f * (back.y + 1.0f)) * (1.0f - mtrans));
    }

    // Next-event estimation.  The three candidates are SAMPLED here, in the reference's order (the random numbers are drawn in
    // that order), but their shadow rays are traced at the end of the vertex, after the continuation has been sampled:
    // shadowed() draws no random numbers and the contributions are added in the same order as before, so every result is
    // unchanged -- and the march of a shadow ray then runs with the vertex's BSDF state (material, frame, half of the hit
    // record) dead instead of live across it, through ONE copy of the march code instead of three.
    const V3 zero3{0.0f, 0.0f, 0.0f};
    // The two contributions every scene has go to rows A and B as soon as they are formed (the hit's position and normal, which
    // those rows held, were read when the vertex began) and wait there for their shadow rays.
    V3 env_wi = zero3, area_wi = zero3, area_c = zero3;
    uint32_t nee_on = 0u, dir_light = 0u;  // (the directional light's direction is read from the light table again when its ray is traced)
    float area_tmax = 1e30f;
    {  // environment, mixture of a power-cosine lobe about +Y and the cosine hemisphere, balance heuristic
        const float u1 = rng_next(rng), u2 = rng_next(rng), u3 = rng_next(rng);
        V3 wi;
        if (u1 < 0.5f) {
            float s, c;
            sincos_turn(u3, s, c);
            const float ct = pow_det(1.0f - u2, 1.0f / (16.0f + 1.0f));
            const float st = f_sqrt(f_max(0.0f, 1.0f - ct * ct));
            wi = V3{st * c, ct, st * s};
        } else {
            wi = to_world(basis, cosine_hemisphere(u2, u3));
        }
        const float pdf_up = up_lobe_pdf(wi);
        const float cos_surf = f_max(dot(n, wi), 0.0f);
        const float pdf_light = 0.5f * pdf_up + (1.0f - 0.5f) * (cos_surf / kPi);
        if (cos_surf > 0.0f) {
            const V3 L_env = mix3(S.env_ground, S.env_sky, 0.5f * (wi.y + 1.0f));
            const Bsdf br = bsdf_eval(M, wo, wi, n);
            const float w_mis = pdf_light / f_max(pdf_light + br.pdf, 1e-8f);
            const float k = (((cos_surf / f_max(pdf_light, 1e-8f)) * w_mis) * M.imp) * mtrans;
            lane.set_a(((thr_in * br.f) * L_env) * k);
            env_wi = wi;
            nee_on |= 1u;
        }
    }
    if (S.dir_count > 0u) {  // delta lights: weight 1
        const uint32_t dir_count = S.dir_count;
        const float dir_sum_imp = S.dir_sum_imp;
        const uint32_t idx = pick(dir_count, dir_sum_imp, rng, [&](uint32_t i) { return S.dir[i].importance; });
        const DirLightDev L = S.dir[idx];
        const float cos_surf = f_max(dot(n, L.wi), 0.0f);
        if (cos_surf > 0.0f) {
            const Bsdf br = bsdf_eval(M, wo, L.wi, n);
            uint32_t n_dir = dir_count;
            float sum_dir = dir_sum_imp;
            F3D_OPAQUE_UNIFORM(n_dir);  // (scene constants: their reciprocal / float are formed here, not in front of the flat loop)
            F3D_OPAQUE_UNIFORM(sum_dir);
            const float p_sel = sum_dir > 0.0f ? L.importance / f_max(sum_dir, 1e-8f) : 1.0f / (float)n_dir;
            const float k = ((cos_surf / f_max(p_sel, 1e-8f)) * M.imp) * mtrans;
            lane.set_b(((thr_in * br.f) * L.Li) * k);
            dir_light = idx;
            nee_on |= 2u;
        }
    }
    if (!Wave::kLite && S.area_count > 0u) {  // discs, sampled uniformly by area, balance heuristic
        const uint32_t idx = pick(S.area_count, S.area_sum_imp, rng, [&](uint32_t i) { return S.area[i].importance; });
        const AreaLightDev L = S.area[idx];
        const float u1 = rng_next(rng), u2 = rng_next(rng);
        const float r = f_sqrt(u1) * L.rad;
        float s, c;
        sincos_turn(u2, s, c);
        const V3 X = (L.position + L.tL * (r * c)) + L.bL * (r * s);
        const V3 dir = X - H.p;
        const float dist = f_sqrt(dot(dir, dir));
        if (dist > 1e-6f) {
            const V3 wi{dir.x / dist, dir.y / dist, dir.z / dist};
            const float cos_surf = f_max(dot(n, wi), 0.0f), cos_on_light = f_max(dot(L.nL, neg(wi)), 0.0f);
            if (cos_surf > 0.0f && cos_on_light > 0.0f) {
                const float pdf = (L.p_area * (dist * dist)) / f_max(cos_on_light, 1e-6f);
                if (pdf > 0.0f) {
                    const Bsdf br = bsdf_eval(M, wo, wi, n);
                    const float p_sel = S.area_sum_imp > 0.0f ? L.importance / f_max(S.area_sum_imp, 1e-8f) : 1.0f / (float)S.area_count;
                    const float pdf_light = p_sel * pdf;
                    const float w_mis = pdf_light / f_max(pdf_light + br.pdf, 1e-8f);
                    const float k = (((cos_surf / f_max(pdf_light, 1e-8f)) * w_mis) * M.imp) * mtrans;
                    area_c = ((thr_in * br.f) * L.Li) * k;
                    area_wi = wi;
                    area_tmax = dist - 1e-3f;
                    nee_on |= 4u;
                }
            }
        }
    }

    const bool go_on = [&]() -> bool {
    V3 wi, thr;
    if (!Wave::kLite && H.hair) {  // Kajiya-Kay, :708-729: cosine-hemisphere continuation weighted by a diffuse term and two lobes about the strand
        const V3 T = normalize(H.tangent);
        const float u1 = rng_next(rng), u2 = rng_next(rng);
        wi = normalize(to_world(basis, cosine_hemisphere(u1, u2)));
        const float lobe = f_max(0.0f, dot(normalize(reflect3(neg(wo), T)), wi));
        const float l2 = lobe * lobe, l4 = l2 * l2, l16 = pow16(lobe), l64 = (l16 * l16) * (l16 * l16);
        const float f1 = l16 * l4, f2 = l64 * l16;  // exponents 20 and 80
        const float kd = 0.2f, ks = 1.0f - kd, spec = ks * (0.6f * f1 + 0.4f * f2);
        const V3 f{kd * (md.albedo.x / kPi) + M.F0.x * spec, kd * (md.albedo.y / kPi) + M.F0.y * spec, kd * (md.albedo.z / kPi) + M.F0.z * spec};
        const float cos_theta = f_max(0.0f, dot(n, wi));
        const float pdf = cos_theta / kPi + 1e-8f;
        thr = (thr_in * f) * (cos_theta / pdf);
    } else if (md.metallic > 0.5f) {  // GGX half-vector sampling
        const float u1 = rng_next(rng), u2 = rng_next(rng);
        const float a = f_max(0.02f, md.roughness * md.roughness);
        const V3 t{basis.t.x, basis.b.x, basis.n.x}, bb{basis.t.y, basis.b.y, basis.n.y}, nn{basis.t.z, basis.b.z, basis.n.z};
        V3 hw;
        if (!M.aniso) {
            const float a2 = a * a;
            const float ct = f_sqrt((1.0f - u1) / (1.0f + (a2 - 1.0f) * u1));
            const float st = f_sqrt(f_max(0.0f, 1.0f - ct * ct));
            float s, c;
            sincos_turn(u2, s, c);
            hw = normalize(to_world(basis, V3{st * c, st * s, ct}));
        } else {
            float s2, c2;
            sincos_turn(u2, s2, c2);
            float phi = atan_det((M.ay / f_max(M.ax, 1e-6f)) * (s2 / c2));
            if (u2 > 0.5f) phi = phi + kPi;
            float sp, cp;
            sincos_rad(phi, sp, cp);
            const float denom = (cp * cp) / f_max(M.ax * M.ax, 1e-8f) + (sp * sp) / f_max(M.ay * M.ay, 1e-8f);
            const float ratio = u1 / f_max(1.0f - u1, 1e-6f);
            const float ct = 1.0f / f_sqrt(1.0f + ratio * denom);
            const float st = f_sqrt(f_max(0.0f, 1.0f - ct * ct));
            hw = normalize((t * (st * cp) + bb * (st * sp)) + nn * ct);
        }
        wi = normalize(reflect3(neg(wo), hw));
        const float n_dot_l = f_max(dot(n, wi), 0.0f), n_dot_h = f_max(dot(n, hw), 0.0f), v_dot_h = f_max(dot(wo, hw), 0.0f);
        if (!(n_dot_l > 0.0f && n_dot_v > 0.0f)) return false;
        const float D = M.aniso ? ggx_d_aniso(hw, t, bb, nn, M.ax, M.ay) : ggx_d(n_dot_h, a);
        const float G = M.aniso ? smith_g1_aniso(wi, t, bb, nn, M.ax, M.ay) * smith_g1_aniso(wo, t, bb, nn, M.ax, M.ay)
                                : smith_g1(n_dot_l, a) * smith_g1(n_dot_v, a);
        const V3 spec = schlick(v_dot_h, M.F0) * ((D * G) / f_max((4.0f * n_dot_l) * n_dot_v, 1e-6f));
        const float pdf = (D * n_dot_h) / f_max(4.0f * v_dot_h, 1e-6f);
        thr = (thr_in * spec) * (n_dot_l / f_max(pdf, 1e-6f));
    } else if (md.ior > 1.01f) {  // smooth dielectric, Schlick-weighted reflect / refract
        const float cosi = f_saturate(dot(n, wo));
        const float r0 = (md.ior - 1.0f) / (md.ior + 1.0f);
        const float F0s = r0 * r0;
        const float F = F0s + (1.0f - F0s) * pow5(1.0f - cosi);
        if (rng_next(rng) < F) {
            wi = normalize(reflect3(neg(wo), n));
        } else {
            const bool entering = dot(n, wo) > 0.0f;
            const float eta = entering ? 1.0f / md.ior : md.ior / 1.0f;
            const V3 N = entering ? n : neg(n), I = neg(wo);
            const float ni = dot(N, I);
            const float kk = 1.0f - (eta * eta) * (1.0f - ni * ni);
            wi = kk < 0.0f ? normalize(reflect3(neg(wo), n)) : normalize(I * eta - N * (eta * ni + f_sqrt(kk)));
        }
        thr = thr_in * V3{f_max(md.albedo.x, 0.0f), f_max(md.albedo.y, 0.0f), f_max(md.albedo.z, 0.0f)};
    } else {  // Lambert
        const float u1 = rng_next(rng), u2 = rng_next(rng);
        wi = normalize(to_world(basis, cosine_hemisphere(u1, u2)));
        const float cos_theta = f_max(0.0f, dot(n, wi));
        const float pdf = cos_theta / kPi + 1e-8f;
        thr = (thr_in * V3{md.albedo.x / kPi, md.albedo.y / kPi, md.albedo.z / kPi}) * (cos_theta / pdf);
    }
    float rr = 1.0f;
    if (depth_in >= 4u) {
        const float q = f_clamp(1.0f - f_max(thr.x, f_max(thr.y, thr.z)), 0.0f, 0.95f);
        if (rng_next(rng) < q) return false;
        rr = 1.0f / (1.0f - q);
    }
    if (!((depth_in + 1u) < 16u)) return false;
    // (the new ray starts at `so` = H.p + normalize(H.n) * 1e-3 with tmin 1e-3: trace_frames takes both from rows A and the depth)
    lane.set_dir(wi);
    lane.set_thr(thr * rr);
    lane.set_word(lane.word() + 1u);  // depth + 1 (the depth sits in the word's low bits and stays below 16)
    lane.set_rng(rng);
    return true;
    }();
    nee.so = so;
    nee.env_wi = env_wi;
    if (!Wave::kLite) {  // (a LITE scene has no area lights)
        nee.area_wi = area_wi;
        nee.area_c = area_c;
        nee.area_tmax = area_tmax;
    }
    nee.on = nee_on;
    nee.dir_light = dir_light;
#if defined(F3D_WF_TIMING_NO_SHADOWS)  // timing experiments only (wrong image): where does the kernel's time go?
    nee.on = 0u;
#endif
#if defined(F3D_WF_TIMING_NO_BOUNCE)
    return false;
#endif
    return go_on;
}

// The deferred shadow rays of a vertex (pt_shadow.wgsl main): every lane takes ITS next one, in the order their contributions
// were added before (environment, directional, area), until no lane of the wave has one left.  The two contributions every
// scene has wait in rows A and B meanwhile; the directional light's direction is read again from the light table.
// ALL_LANES (wave-uniform call): every lane of the wave goes through the heightfield's march, the ones without a ray with an
// empty one -- inside the march they are lanes whose ray has ended, which is what the ray sharing of f3d_march.h deals the
// last rays of a wave to.  Verdicts do not depend on who walks a slice: same results.
template <bool ALL_LANES, class Wave>
F3D_HD void vertex_shadows(const SceneDev &S, NeeRays &nee, LaneState<Wave> &lane, Wave &wave) {
#if defined(F3D_WF_SHADOW_STREAM)
    // A/B (round 6): the rays as a STREAM (f3d_march.h march_stream): a lane whose ray is done takes ITS next one while the others
    // still march -- max over lanes of (env + dir steps) instead of max(env) + max(dir).  Measured: 48.5 ms against 34.4.
    if (Wave::kTerrain && S.has_terrain != 0u && (Wave::kLite || (S.blas_count == 0u && S.inst_count == 0u && S.area_count == 0u))) {
        NeeShadowSource<Wave> src{S, lane, nee.so, nee.env_wi, nee.dir_light, nee.on & 3u};
        march_stream<false>(S.terrain, src, *wave.pend, 16u);
        nee.on = 0u;
    }
#endif
    while (wave.count(nee.on != 0u) != 0u) {
        const bool have = nee.on != 0u;
        if (ALL_LANES) {
            uint32_t k = 0u;
            V3 wi{0.0f, 1.0f, 0.0f};
            if (have) {
                k = (uint32_t)__builtin_ctz(nee.on);
                nee.on &= nee.on - 1u;
                wi = k == 0u ? nee.env_wi : S.dir[nee.dir_light].wi;
            }
            // (tmax < tmin: the root interval of a lane without a ray is empty, it never marches)
            const RayCtx r = make_ray(S.terrain, nee.so, have ? 1e-3f : 1.0f, wi, have ? 1e30f : 0.0f, false);
            bool blocked = march_terrain<false>(S.terrain, r, true, true, *wave.pend).hit;
            if (have && !blocked) blocked = sphere_shadow(S, nee.so, wi, 1e-3f, 1e30f);
            if (have && !blocked) lane.add(k == 0u ? lane.a() : lane.b());
        } else if (have) {
            const uint32_t k = (uint32_t)__builtin_ctz(nee.on);
            nee.on &= nee.on - 1u;
            const V3 wi = k == 0u ? nee.env_wi : (k == 1u ? S.dir[nee.dir_light].wi : nee.area_wi);
            if (!shadowed(S, nee.so, wi, 1e-3f, k == 2u ? nee.area_tmax : 1e30f, wave))
                lane.add(k == 0u ? lane.a() : (k == 1u ? lane.b() : nee.area_c));
        }
    }
}

// How many lanes of the wave could use another closest-hit attempt (the host "wave" is one lane wide), the rows of the lane's
// state (LaneState), which pixel the lane is, and the traversal context of the terrain primitive (f3d_march.h Ctx: the device's
// LDS block, the emulator's arrays).  kTerrain: the kernel was compiled with the heightfield primitive.
template <class Pend>
struct SoloWave {
    static constexpr bool kTerrain = true;
    static constexpr bool kLite = false;
    static constexpr uint32_t kPark = kLaneRows;  // (the host runs the row form of the code, the rows being an array)
    static constexpr uint32_t kFrameStride = 1u;
    Pend *pend;
    uint32_t pixel_, width_;
    mutable uint32_t parked[kLaneRows];
    F3D_HD uint32_t count(bool flag) const { return flag ? 64u : 0u; }
    F3D_HD uint32_t pixel() const { return pixel_; }
    F3D_HD uint32_t px() const { return pixel_ % width_; }
    F3D_HD uint32_t py() const { return pixel_ / width_; }
    F3D_HD void park(uint32_t row, float v) const { parked[row] = f_bits(v); }
    F3D_HD float unpark(uint32_t row) const { return f_from_bits(parked[row]); }
};
#if defined(__HIPCC__)
// FRAME_LANES (round 6): neighbouring lanes of a wave are the SAME pixel taking its frames in turns (lane % FRAME_LANES = which
// turn), the wave's tile being 64 / FRAME_LANES pixels -- the terrain tracer's "sample lanes" (DESIGN.md 4.5): their camera rays
// are almost the same ray, their vertices almost the same point, so their marches are as long as each other and read the same
// table entries.  Every frame's total goes out under its own frame number and is folded in frame order: same results.
constexpr uint32_t wf_tile_w(uint32_t frame_lanes) { return frame_lanes <= 2u ? 8u : (frame_lanes <= 8u ? 4u : 2u); }
constexpr uint32_t wf_tile_h(uint32_t frame_lanes) { return 64u / frame_lanes / wf_tile_w(frame_lanes); }
template <class Pend, bool TERRAIN, bool LITE = false, uint32_t FRAME_LANES = 1u>
struct HipWave {
    static_assert(FRAME_LANES == 1u || FRAME_LANES == 2u || FRAME_LANES == 4u || FRAME_LANES == 8u || FRAME_LANES == 16u, "a power of two up to 16");
    static constexpr bool kTerrain = TERRAIN;
    static constexpr bool kLite = LITE;  // no meshes, hair, area lights, fog in the scene: their code is not in the kernel
    // rows of the lane's LDS column that hold the path's loop-carried state (0: no LDS block, the state lives in registers)
    static constexpr uint32_t kPark = TERRAIN ? (uint32_t)kPathParkRows : 0u;
    static constexpr uint32_t kFrameStride = FRAME_LANES, kTileW = wf_tile_w(FRAME_LANES), kTileH = wf_tile_h(FRAME_LANES);
    Pend *pend;
    uint32_t x0, y0, width;  // the wave's tile of kTileW x kTileH pixels (wave-uniform)
    __device__ uint32_t count(bool flag) const { return (uint32_t)__popcll(__ballot(flag)); }
    // the lane's pixel, formed where it is used from the hardware's lane id (f3d_lds.h lane_now: never hoisted, never kept)
    __device__ uint32_t px() const { return x0 + ((lane_now() / FRAME_LANES) % kTileW); }
    __device__ uint32_t py() const { return y0 + (lane_now() / (FRAME_LANES * kTileW)); }
    __device__ uint32_t pixel() const { return py() * width + px(); }
    __device__ void park(uint32_t row, float v) const { pend->col[(kPathParkRow0 + row) * kWave] = f_bits(v); }
    __device__ float unpark(uint32_t row) const { return f_from_bits(pend->col[(kPathParkRow0 + row) * kWave]); }
};
#endif

// A pixel's frames [first, first + count), count <= kMaxFramesPerCall: `sink(frame, total)` receives every frame's
// contributions, summed from zero in stage order; returns the number of path vertices (closest-hit queries) traced.
//
// The loop is FLAT: its body is one path vertex, and a lane whose path ended starts its next frame's camera ray in the
// same pass, so the lanes of a wave never wait for the longest path of a frame.  It is also TWO-PHASE: about half of
// all closest-hit queries are misses (every path ends with one), and a miss costs a sixth of a surface vertex; so the
// cheap phase (camera ray, closest hit, background) is repeated for the lanes without a pending surface hit for as
// long as at least F3D_WF_REFILL of them can still use it, and only then do the lanes with a hit run the expensive phase
// together.  Measured at the adjudication gate: 146 ms without refills, 131 ms with a threshold of 6...20, 141 ms at 48.  Each lane's own
// sequence of operations -- and therefore every result -- is the same for any wave width or refill threshold.
// With the heightfield primitive (BASELINE configs[2], six waves a SIMD): 8 (20 / 12 / 8 / 4 -> 23.3 / 22.8 / 22.9 / 22.8 ms).
#ifndef F3D_WF_REFILL
#define F3D_WF_REFILL 20
#endif
#ifndef F3D_WF_REFILL_TERRAIN
#define F3D_WF_REFILL_TERRAIN 8
#endif

template <class Wave, class Sink>
F3D_HD uint32_t trace_frames(const SceneDev &S, uint32_t first, uint32_t count, Wave wave, Sink &&sink) {
    using Lane = LaneState<Wave>;
    Lane lane(wave);
    lane.set_acc(V3{0.0f, 0.0f, 0.0f});
    lane.set_word(Lane::kFresh);  // depth 0, frame `first`, no query yet
    // what of a hit does not fit the rows (scenes with fog or hair only: the LITE kernel carries none of it)
    float hit_t = 0.0f;
    bool hit_hair = false;
    V3 hit_tangent{0.0f, 0.0f, 0.0f};
    // (Wave::kFrameStride: the lanes of a pixel take its frames in turns -- the lane's k-th frame is first + k * stride)
    auto frame_of = [&](uint32_t word) F3D_LAMBDA { return first + Wave::kFrameStride * ((word >> Lane::kFrameShift) & Lane::kFrameMask); };
    auto active = [&](uint32_t word) F3D_LAMBDA { return (word & Lane::kPending) == 0u && ((word >> Lane::kFrameShift) & Lane::kFrameMask) < count; };
    auto finish_frame = [&]() F3D_LAMBDA {  // the frame's total goes out, the next frame starts from zero
        const uint32_t word = lane.word();
        sink(frame_of(word), lane.acc());
        lane.set_acc(V3{0.0f, 0.0f, 0.0f});
        lane.set_word(((word & ~Lane::kDepthMask) + (1u << Lane::kFrameShift)) | Lane::kFresh);
    };
    // a scene whose primitives are the heightfield and spheres: every lane of the wave goes through the marches (the LITE kernel is
    // only ever launched for such a scene: no run-time test, and the other form is not in it)
#if defined(F3D_WF_STATS)
    uint32_t stat = 0u;
#endif
    const bool all_lanes = Wave::kLite || (Wave::kTerrain && S.has_terrain != 0u && S.blas_count == 0u && S.inst_count == 0u &&
                                           S.area_count == 0u && S.hair_count == 0u);
    for (;;) {
        for (;;) {  // cheap phase
            uint32_t word = lane.word();
            const bool act = active(word);
#if !defined(F3D_WF_CLOSEST_DIVERGENT)  // A/B: the round-5 form -- only the lanes with a ray enter closest()
            if (act || all_lanes) {
#else
            if (act) {
#endif
                V3 o{0.0f, 0.0f, 0.0f}, d{0.0f, 1.0f, 0.0f};
                if (!act) {
                } else if (word & Lane::kFresh) {
                    const uint32_t frame = frame_of(word);
                    const uint32_t seed_hi = splitmix32(S.seed_hi ^ frame), seed_lo = splitmix32(S.seed_lo ^ (frame * 0x00009E3Du));
                    PathState P;
                    camera_ray(S, wave.px(), wave.py(), frame, seed_hi, seed_lo, P);  // (the pixel's column and row: no division by the image width)
                    o = P.o;
                    d = P.d;
                    lane.set_dir(d);
                    lane.set_thr(P.thr);
                    lane.set_rng(P.rng_hi);
                    word = word & ~(Lane::kFresh | Lane::kDepthMask);
                } else {  // the ray the last vertex left: origin in rows A, direction in its rows
                    o = lane.a();
                    d = lane.dir();
                }
                const uint32_t depth = word & Lane::kDepthMask;
                if (act) word += 1u << Lane::kQueryShift;
                SurfaceHitWf H;
                const bool hit = closest(S, o, d, depth == 0u ? 1e-4f : 1e-3f, H, wave, depth == 0u ? wave.pixel() : kNoPixel, act);
                if (!act) {
                } else if (hit) {
                    // the hit waits for the expensive phase in the rows, not in registers
                    lane.set_a(H.p);
                    lane.set_b(H.n);
                    lane.set_mat(H.mat);
                    if (!Wave::kLite) {
                        hit_t = H.t;
                        hit_hair = H.hair;
                        hit_tangent = H.tangent;
                    }
                    lane.set_word(word | Lane::kPending);
                } else {  // pt_scatter.wgsl:113-131
                    lane.set_word(word);
                    lane.add(lane.thr() * mix3(S.miss_ground, S.miss_sky, 0.5f * (lane.dir().y + 1.0f)));
                    finish_frame();
                }
            }
            if (wave.count(active(lane.word())) < (Wave::kTerrain ? (uint32_t)F3D_WF_REFILL_TERRAIN : (uint32_t)F3D_WF_REFILL)) break;
        }
        const bool pending = (lane.word() & Lane::kPending) != 0u;
        if (wave.count(pending) == 0u) {
            if (wave.count(((lane.word() >> Lane::kFrameShift) & Lane::kFrameMask) < count) == 0u) break;
            continue;
        }
        // expensive phase: the waiting hits are shaded; then the vertices' shadow rays; then the paths continue or end
#if defined(F3D_WF_STATS)  // statistics build (the image is right, the vertex count is not): how full is the expensive phase?
        {
            const uint32_t np = wave.count(pending);
            stat += F3D_WF_STATS == 1 ? 1u : F3D_WF_STATS == 2 ? np : F3D_WF_STATS == 3 ? (np <= 32u ? 1u : 0u) : (np <= 16u ? 1u : 0u);
        }
#endif
        NeeRays nee;
        bool go_on = false;
        if (pending) {
            const uint32_t word = lane.word() & ~Lane::kPending;
            lane.set_word(word);
            SurfaceHitWf H;
            H.p = lane.a();
            H.n = lane.b();
            H.mat = lane.mat();
            H.t = hit_t;
            H.hair = hit_hair;
            H.tangent = hit_tangent;
            go_on = vertex_shade(S, frame_of(word), H, lane.dir(), lane, wave, nee);
        }
#if !defined(F3D_WF_SHADOWS_DIVERGENT)  // A/B: the round-5 form -- only the lanes with a vertex enter the shadow rays' march
        // (a scene whose occluders are the heightfield and spheres: every lane of the wave goes through the shadow rays' march)
        // (the LITE kernel is only ever launched for such a scene: no run-time test, and the other form is not in it)
        if (all_lanes) {
            vertex_shadows<true>(S, nee, lane, wave);
        } else
#endif
        if (pending) vertex_shadows<false>(S, nee, lane, wave);
        if (pending) {
            if (go_on) lane.set_a(nee.so);  // the continuing ray's origin, for the next closest()
            else finish_frame();
        }
    }
#if defined(F3D_WF_STATS)
    return lane_now() == 0u ? stat : 0u;
#endif
    return lane.word() >> Lane::kQueryShift;
}

}  // namespace wf
}  // namespace f3d
