// forge3d_amd/csrc/f3d_cone.h
// Where the primary rays of a pixel can start.
//
// The camera does not move between the frames of a render and the jitter of a sample stays inside +-0.5 px, so all
// spp x frames camera rays of a pixel lie in one thin cone with its apex at the camera.  primary_start() marches that
// CONE once (k_gbuffer) and returns a parameter t_clear such that every ray of the pixel, at every parameter up to
// t_clear, is above the highest point of every cell it is over -- by a margin far above the rounding of the march's
// band test.  The nodes a camera ray passes before t_clear are then exactly nodes whose band test rejects the ray
// (`lo > mx`: no leaf is ever solved in them; a bilinear patch along a straight ray is a quadratic, and a ray above
// the cell's maximum has no root), so a march that starts at t_clear (f3d_march.h march_begin_at) returns what a march
// from the root returns.  The certificate only has to be conservative, not bit-exact with anything: a shorter t_clear
// costs steps, never a result.
#pragma once

#ifndef F3D_HORIZON_THETA
#define F3D_HORIZON_THETA 8.0f  // a node is taken whole when it is this many times smaller than its distance (A/B: 4, 6)
#endif

// (included by f3d_shade.h after camera_dir, which it uses)

namespace f3d {

struct PrimaryStart {
    float t_clear;   // 0: no certificate (march from the root); 3e38: no ray of the pixel ever meets terrain
    uint32_t level;  // level of the node the certificate stopped at: where to drop the rays in
};

// Largest band maximum of the 3 x 3 nodes of `level` around (nx, nz) (nodes outside the grid hold no terrain).
F3D_HD float cone_max9(const TerrainDev &T, uint32_t level, uint32_t nx, uint32_t nz) {
    float mx = -3.0e38f;
    for (int dz = -1; dz <= 1; dz++)
        for (int dx = -1; dx <= 1; dx++) {
            const uint32_t qx = nx + (uint32_t)dx, qz = nz + (uint32_t)dz;  // -1 wraps and fails the range test
            if ((qx << level) < T.cell_w && (qz << level) < T.cell_h && qx <= nx + 1u && qz <= nz + 1u)
                mx = f_max(mx, T.bands[T.band_offset[level] + (qz << T.band_shift[level]) + qx].mx);
        }
    return mx;
}

F3D_HD PrimaryStart primary_start(const FrameParams &P, uint32_t gx, uint32_t gy) {
    const TerrainDev &T = P.terrain;
    const uint32_t top = T.mip_count - 1u;
    PrimaryStart out{0.0f, top};
    // |d - d_centre| of any ray of the pixel: the jitter moves the point on the z = -1 plane by at most
    // (half_w / W, half_h / H); sin(angle) <= |delta| / |v| <= |delta|, chord <= angle, asin(x) <= x (1 + x^2)
    const float px = P.cam.half_w / (float)P.cam.width, py = P.cam.half_h / (float)P.cam.height;
    const float plane = f_sqrt(px * px + py * py);
    if (!(plane < 0.25f)) return out;
    const float delta = 1.01f * plane * (1.0f + plane * plane);
    const V3 d = camera_dir(P.cam, gx, gy, 0.0f, 0.0f);
    const RayCtx r = make_ray(T, P.cam.origin, 1e-3f, d, 1e30f, false);
    float t_in, t_out;
    march_root_interval(T, r, t_in, t_out);
    if (t_in > t_out) return out;  // the centre ray misses the footprint; its neighbours may not: no certificate
    const float cell = f_min(T.spacing_x, T.spacing_z);
    // height margin: the band test compares f32 heights of the order of the ray's and the terrain's
    const float y_scale = f_abs(r.o.y) + f_abs(T.bands[T.band_offset[top]].mx) + f_abs(T.bands[T.band_offset[top]].mn);
    auto lowest = [&](float t) F3D_LAMBDA {  // height of the lowest ray of the cone at parameter t, minus the margin
        return f_fma(t, r.d.y, r.o.y) - t * delta - (1e-4f * y_scale + 1e-5f * t + 1e-3f);
    };
    // before the centre ray enters the footprint the cone may already be over it: clear only above everything
    if (t_in > r.tmin && !(f_min(lowest(r.tmin), lowest(t_in)) > T.bands[T.band_offset[top]].mx)) return out;
    // The centre ray has left the footprint at parameter b; the far side of the cone may still be over it.  The pixel sees
    // no terrain at all only if the cone never comes down again: its lowest ray does not descend and is above everything.
    auto beyond_is_clear = [&](float b) F3D_LAMBDA {
        return r.d.y - delta >= 0.0f && lowest(b) > T.bands[T.band_offset[top]].mx;
    };
    const bool x_forward = !(r.d.x < 0.0f), z_forward = !(r.d.z < 0.0f);
    uint32_t level = top, nx = 0u, nz = 0u;
    float clear = t_in;
    for (uint32_t iter = 0u; iter < 512u; iter++) {
        uint32_t cx1 = (nx + 1u) << level, cz1 = (nz + 1u) << level;
        cx1 = cx1 < T.cell_w ? cx1 : T.cell_w;
        cz1 = cz1 < T.cell_h ? cz1 : T.cell_h;
        const float tx0 = (plane_at(T.origin_x, nx << level, T.spacing_x) - r.o.x) * r.inv_x;
        const float tx1 = (plane_at(T.origin_x, cx1, T.spacing_x) - r.o.x) * r.inv_x;
        const float tz0 = (plane_at(T.origin_z, nz << level, T.spacing_z) - r.o.z) * r.inv_z;
        const float tz1 = (plane_at(T.origin_z, cz1, T.spacing_z) - r.o.z) * r.inv_z;
        const float x_out = f_max(tx0, tx1), z_out = f_max(tz0, tz1);
        const float enter = f_max(f_min(tx0, tx1), f_min(tz0, tz1)), exit = f_min(x_out, z_out);
#if defined(F3D_CONE_DEBUG)
        fprintf(stderr, "  iter %u level %u node (%u,%u) enter %.3f exit %.3f clear %.3f\n", iter, level, nx, nz, enter, exit, clear);
#endif
        if (!(enter <= clear && clear <= exit)) break;  // lost the ray (rounding at a boundary): stop here
        const float b = f_min(exit, t_out);
        // the cone's radius over [clear, b], plus slack for the ray sitting a hair outside the node
        const float radius = b * delta + 0.02f * cell;
        uint32_t need = 0u;  // finest level whose nodes are wider than the radius: the 3 x 3 block of such nodes holds the dilated node
        while (need < top && cell * (float)(1u << need) < radius) need++;
        const uint32_t ql = need > level ? need : level;
        const bool wide_enough = cell * (float)(1u << ql) >= radius;
        const float y_low = f_min(lowest(clear), lowest(b));
#if defined(F3D_CONE_DEBUG)
        fprintf(stderr, "    b %.3f radius %.3f ql %u y_low %.3f max9 %.3f\n", b, radius, ql, y_low, wide_enough ? cone_max9(T, ql, nx >> (ql - level), nz >> (ql - level)) : -1.0f);
#endif
        if (wide_enough && y_low > cone_max9(T, ql, nx >> (ql - level), nz >> (ql - level))) {
            clear = b;
            if (!(b < t_out)) {
                if (beyond_is_clear(b)) clear = 3.0e38f;
                break;
            }
            const bool cross_x = x_out <= z_out, cross_z = z_out <= x_out;
            const uint32_t qx = nx + ((cross_x && x_forward) ? 1u : 0u) - ((cross_x && !x_forward) ? 1u : 0u);
            const uint32_t qz = nz + ((cross_z && z_forward) ? 1u : 0u) - ((cross_z && !z_forward) ? 1u : 0u);
            if ((qx << level) >= T.cell_w || (qz << level) >= T.cell_h) {
                if (beyond_is_clear(b)) clear = 3.0e38f;
                break;
            }
            const bool up = level < top && (((qx ^ nx) | (qz ^ nz)) > 1u);
            nx = up ? qx >> 1 : qx;
            nz = up ? qz >> 1 : qz;
            level = up ? level + 1u : level;
        } else if (level > 0u && need < level) {
            // a finer node can still hold the cone: down into the child the centre ray is in at `clear`
            const uint32_t cl = level - 1u;
            const uint32_t xm = (2u * nx + 1u) << cl, zm = (2u * nz + 1u) << cl;
            const float txm = (plane_at(T.origin_x, xm, T.spacing_x) - r.o.x) * r.inv_x;
            const float tzm = (plane_at(T.origin_z, zm, T.spacing_z) - r.o.z) * r.inv_z;
            uint32_t ix = (x_forward != (txm <= clear)) ? 0u : 1u;
            uint32_t iz = (z_forward != (tzm <= clear)) ? 0u : 1u;
            if (!(xm < T.cell_w)) ix = 0u;
            if (!(zm < T.cell_h)) iz = 0u;
            nx = 2u * nx + ix;
            nz = 2u * nz + iz;
            level = cl;
        } else {
            break;  // the terrain is too close to the cone here: the rays take over
        }
    }
    out.t_clear = clear > t_in ? clear : 0.0f;
    out.level = level;
    return out;
}

// ---- where the sun rays of a pixel can stop -------------------------------------------------------------------------
// The sun rays of a pixel's samples are parallel and start within a short distance of one another (the samples' hit
// points lie in the pixel's cone, at depths the caller checks against the centre ray's: sun_depth_slack), so they
// fill a CYLINDER around the centre sample's sun ray.  sun_clear_from() walks that cylinder through the footprint and
// returns the parameter after which it stays above every cell it is over: beyond it no sun ray of the pixel can meet
// terrain, so their marches may stop there (the terrain ray's tmax; the mesh test keeps the full range).  The sun
// rays' curvature policy only LIFTS a ray (c2 >= 0), so the straight centre line is a lower bound.
#ifndef F3D_SUN_SLACK
#define F3D_SUN_SLACK 1.0f
#endif
// (a render constant: evaluated once by the host, f3d_setup.h fill_uniforms -> CameraDev::cone_delta)
F3D_HD float pixel_cone_delta(const CameraDev &C) { return C.cone_delta; }
// Samples whose primary hit lies within this distance (along the ray) of the centre ray's hit use the certificate.
F3D_HD float sun_depth_slack(float centre_depth, float delta, float cell) { return F3D_SUN_SLACK * centre_depth * delta + 0.5f * cell; }

F3D_HD float sun_clear_from(const FrameParams &P, V3 origin, float centre_depth) {
    const TerrainDev &T = P.terrain;
    const uint32_t top = T.mip_count - 1u;
    const float none = 3.0e38f;
    const float delta = pixel_cone_delta(P.cam);
    const V3 d = P.light.wi;
    if (delta < 0.0f || !(d.y >= 0.0f)) return none;
    const float cell = f_min(T.spacing_x, T.spacing_z);
    // |sample origin - centre origin| <= depth difference + the cone's width there + the two offsets along the normals
    const float slack = sun_depth_slack(centre_depth, delta, cell);
    const float rho = slack + (centre_depth + slack) * delta + 4e-3f;
    const RayCtx r = make_ray(T, origin, 1e-3f, d, 1e30f, false);
    float t_in, t_out;
    march_root_interval(T, r, t_in, t_out);
    if (t_in > t_out || t_in > r.tmin) return none;  // a mesh hit beside the footprint: the usual march
    const float y_scale = f_abs(r.o.y) + f_abs(T.bands[T.band_offset[top]].mx) + f_abs(T.bands[T.band_offset[top]].mn);
    auto lowest = [&](float t) F3D_LAMBDA { return f_fma(t, r.d.y, r.o.y) - rho - (1e-4f * y_scale + 1e-5f * t + 1e-3f); };
    const float radius = rho + 0.02f * cell;
    uint32_t need = 0u;
    while (need < top && cell * (float)(1u << need) < radius) need++;
    if (cell * (float)(1u << need) < radius) return none;
    const bool x_forward = !(r.d.x < 0.0f), z_forward = !(r.d.z < 0.0f);
    uint32_t level = top, nx = 0u, nz = 0u;
    float at = t_in, clear_from = t_in;  // the cylinder is known to be clear on [clear_from, at]
    float last_width = 0.0f;
    bool exit_x = false, exit_z = false;
    for (uint32_t iter = 0u; iter < 1024u; iter++) {
        uint32_t cx1 = (nx + 1u) << level, cz1 = (nz + 1u) << level;
        cx1 = cx1 < T.cell_w ? cx1 : T.cell_w;
        cz1 = cz1 < T.cell_h ? cz1 : T.cell_h;
        const float tx0 = (plane_at(T.origin_x, nx << level, T.spacing_x) - r.o.x) * r.inv_x;
        const float tx1 = (plane_at(T.origin_x, cx1, T.spacing_x) - r.o.x) * r.inv_x;
        const float tz0 = (plane_at(T.origin_z, nz << level, T.spacing_z) - r.o.z) * r.inv_z;
        const float tz1 = (plane_at(T.origin_z, cz1, T.spacing_z) - r.o.z) * r.inv_z;
        const float x_out = f_max(tx0, tx1), z_out = f_max(tz0, tz1);
        const float enter = f_max(f_min(tx0, tx1), f_min(tz0, tz1)), exit = f_min(x_out, z_out);
        if (!(enter <= at && at <= exit)) return none;  // lost the ray: no certificate
        const float b = f_min(exit, t_out);
        const uint32_t ql = need > level ? need : level;
        const bool pass = f_min(lowest(at), lowest(b)) > cone_max9(T, ql, nx >> (ql - level), nz >> (ql - level));
#if defined(F3D_CONE_DEBUG)
        fprintf(stderr, "  sun iter %u level %u node (%u,%u) [%.3f, %.3f] at %.3f b %.3f lowest %.3f max9 %.3f pass %d need %u\n", iter, level, nx, nz, enter, exit, at, b,
                f_min(lowest(at), lowest(b)), cone_max9(T, ql, nx >> (ql - level), nz >> (ql - level)), (int)pass, need);
#endif
        if (!pass && level > 0u && need < level) {  // a finer node may still be under the cylinder: down
            const uint32_t cl = level - 1u;
            const uint32_t xm = (2u * nx + 1u) << cl, zm = (2u * nz + 1u) << cl;
            const float txm = (plane_at(T.origin_x, xm, T.spacing_x) - r.o.x) * r.inv_x;
            const float tzm = (plane_at(T.origin_z, zm, T.spacing_z) - r.o.z) * r.inv_z;
            uint32_t ix = (x_forward != (txm <= at)) ? 0u : 1u;
            uint32_t iz = (z_forward != (tzm <= at)) ? 0u : 1u;
            if (!(xm < T.cell_w)) ix = 0u;
            if (!(zm < T.cell_h)) iz = 0u;
            nx = 2u * nx + ix;
            nz = 2u * nz + iz;
            level = cl;
            continue;
        }
        if (!pass) clear_from = b;  // terrain reaches the cylinder in this node: whatever is clear starts after it
        at = b;
        const bool cross_x = x_out <= z_out, cross_z = z_out <= x_out;
        last_width = cell * (float)(1u << ql);
        exit_x = cross_x;
        exit_z = cross_z;
        if (!(b < t_out)) break;
        const uint32_t qx = nx + ((cross_x && x_forward) ? 1u : 0u) - ((cross_x && !x_forward) ? 1u : 0u);
        const uint32_t qz = nz + ((cross_z && z_forward) ? 1u : 0u) - ((cross_z && !z_forward) ? 1u : 0u);
        if ((qx << level) >= T.cell_w || (qz << level) >= T.cell_h) break;
        const bool up = level < top && (((qx ^ nx) | (qz ^ nz)) > 1u);
        nx = up ? qx >> 1 : qx;
        nz = up ? qz >> 1 : qz;
        level = up ? level + 1u : level;
    }
    if (!(clear_from < at)) return none;  // the last node was not clear
    // The centre line has left the footprint, but one side of the cylinder may run on over it for a while: a line
    // leaving through one edge is `radius` away from it after radius / |d_perp|, having moved radius |d_par| / |d_perp|
    // along it.  If that stretch (plus the radius) is shorter than the nodes of the last test, every cell the cylinder
    // is still over lay in that test's 3 x 3 block, and the rays only rise (d.y >= 0).  Else -- a shallow exit, or
    // through a corner -- certified only if the cylinder is above EVERYTHING from here on.
    bool edge_ok = false;
    if (exit_x != exit_z) {
        const float perp = f_abs(exit_x ? r.d.x : r.d.z), par = f_abs(exit_x ? r.d.z : r.d.x);
        edge_ok = perp > 0.0f && radius * (par / perp) + radius <= last_width;
    }
    if (!edge_ok && !(lowest(at) > T.bands[T.band_offset[top]].mx)) return none;
    return clear_from;
}

// ---- where the IBL rays can stop: a far-horizon table of the DEM ---------------------------------------------------
// An IBL ray has a random direction and starts on the terrain.  For every BLOCK of 2^B x 2^B cells and each of 8 azimuth
// sectors the table holds the steepest slope under which anything of the terrain FARTHER than `near` (horizontally, from
// the block's centre) is seen from ANY point of the block's surface: a ray that starts on the block and climbs more
// steeply is above every cell beyond `near`, so its march may stop once it has left the near cells behind
// (f3d_march.h t_stop, the SLICED rule).  Nodes of the pyramid are taken whole as soon as they are small against their
// distance; the bound of a node (its maximum over its nearest point) bounds all its cells, so coarse nodes only make the
// horizon more cautious.  A certificate only has to be conservative, never bit-exact with anything.
// Round 2 computed such horizons per PIXEL in the G-buffer pass: 26 ms per render at 1080p (a quadtree walk per pixel) for
// 0.13 ms saved per frame.  The table depends on the DEM and its spacing only -- not on camera, sun or image -- so it is
// built once per scene (k_horizon_build, ~one walk per block), lives in the scene cache next to the band tables, and at
// the headline camera its blocks (rho = 28 m) are tighter than the pixel cones' footprints there (rho ~ 55 m).
F3D_HD uint32_t ibl_sector(float dx, float dz) {
    return (dx < 0.0f ? 1u : 0u) | (dz < 0.0f ? 2u : 0u) | (f_abs(dz) > f_abs(dx) ? 4u : 0u);
}
// cells nearer than this are the march's; beyond it a cell (dilated by rho) is seen under less than ~35 degrees, so its
// four corners tell which of the 45-degree sectors it is part of
F3D_HD float ibl_near(float rho, float cell) { return 2.5f * (cell + 2.0f * rho); }
F3D_HD float ibl_stop_distance(float rho, float cell) { return ibl_near(rho, cell) + 2.0f * cell + 2.0f * rho; }

// Blocks: the nodes of pyramid level B, B the smallest level with at most 2^18 of them (one walk each at build time).
constexpr uint32_t kHorizonMaxBlocks = 1u << 18;
F3D_HD uint32_t horizon_block_level(uint32_t cell_w, uint32_t cell_h) {
    uint32_t b = 0u;
    while ((uint64_t)((cell_w + (1u << b) - 1u) >> b) * (uint64_t)((cell_h + (1u << b) - 1u) >> b) > kHorizonMaxBlocks) b++;
    return b;
}
// horizontal radius around a block's centre that holds every IBL-ray origin on the block: half its diagonal, the 1e-3
// lift off the surface, and slack for a hit point rounded across the block's border
F3D_HD float horizon_block_rho(float spacing_x, float spacing_z, uint32_t level) {
    const float w = spacing_x * (float)(1u << level), d = spacing_z * (float)(1u << level);
    return 0.5f * f_sqrt(w * w + d * d) + 0.02f * f_max(spacing_x, spacing_z) + 4e-3f;
}
F3D_HD float horizon_block_rho(const TerrainDev &T, uint32_t level) { return horizon_block_rho(T.spacing_x, T.spacing_z, level); }
// how far below the block's lowest corner an IBL-ray origin may lie (solver rounding + the lift along a normal)
F3D_HD float horizon_y_margin(const TerrainDev &T) {
    const uint32_t top = T.mip_count - 1u;
    return 1e-4f * (f_abs(T.bands[T.band_offset[top]].mx) + f_abs(T.bands[T.band_offset[top]].mn)) + 4e-3f;
}

// out[8]: the far horizon's slope per sector, for origins within `rho` (horizontally) of (ox, oz) and not below y_lo.
F3D_HD void far_horizon_from(const TerrainDev &T, float ox, float oz, float rho, float y_lo, float *out) {
    const uint32_t top = T.mip_count - 1u;
    const float cell_min = f_min(T.spacing_x, T.spacing_z), cell_max = f_max(T.spacing_x, T.spacing_z);
    const float near = ibl_near(rho, cell_max);
    float best[kIblSectors];
    for (uint32_t s = 0u; s < kIblSectors; s++) best[s] = -3.0e38f;
    for (uint32_t s = 0u; s < kIblSectors; s++) out[s] = 3.0e38f;
    uint32_t stack[64];
    uint32_t sp = 0u;
    stack[sp++] = top << 26;
    uint32_t visited = 0u;
    while (sp != 0u) {
        if (++visited > 20000u) return;  // (never seen; a pathological DEM simply gets no certificate)
        const uint32_t e = stack[--sp];
        const uint32_t l = e >> 26, nz = (e >> 13) & 0x1FFFu, nx = e & 0x1FFFu;
        if ((nx << l) >= T.cell_w || (nz << l) >= T.cell_h) continue;
        uint32_t cx1 = (nx + 1u) << l, cz1 = (nz + 1u) << l;
        cx1 = cx1 < T.cell_w ? cx1 : T.cell_w;
        cz1 = cz1 < T.cell_h ? cz1 : T.cell_h;
        const float x0 = plane_at(T.origin_x, nx << l, T.spacing_x) - ox, x1 = plane_at(T.origin_x, cx1, T.spacing_x) - ox;
        const float z0 = plane_at(T.origin_z, nz << l, T.spacing_z) - oz, z1 = plane_at(T.origin_z, cz1, T.spacing_z) - oz;
        // nearest and farthest point of the rectangle from the origin (in the plane)
        const float nxp = f_max(f_max(x0, -x1), 0.0f), nzp = f_max(f_max(z0, -z1), 0.0f);
        const float fxp = f_max(f_abs(x0), f_abs(x1)), fzp = f_max(f_abs(z0), f_abs(z1));
        const float dmin = f_sqrt(nxp * nxp + nzp * nzp), dmax = f_sqrt(fxp * fxp + fzp * fzp);
        if (!(dmax > near)) continue;  // a near node: the march sees it
        const float mx = T.bands[T.band_offset[l] + (nz << T.band_shift[l]) + nx].mx;
        const float den = f_max(dmin, near) - rho - 0.02f * cell_min;  // > 0: near > 2 rho + cell
        const float bound = (mx - y_lo) / den;
        const float size = f_max(x1 - x0, z1 - z0) + 2.0f * rho;
        const bool small = size * 2.5f <= dmin;  // under ~35 degrees (diagonal) as seen from the origin: its corners tell its sectors
        uint32_t touched = 0xFFu;
        if (small) {
            const float ax0 = x0 - rho, ax1 = x1 + rho, az0 = z0 - rho, az1 = z1 + rho;
            touched = (1u << ibl_sector(ax0, az0)) | (1u << ibl_sector(ax1, az0)) | (1u << ibl_sector(ax0, az1)) | (1u << ibl_sector(ax1, az1));
        }
        float need = 3.0e38f;
        for (uint32_t s = 0u; s < kIblSectors; s++)
            if (touched & (1u << s)) need = f_min(need, best[s]);
        if (!(bound > need)) continue;  // cannot raise any horizon it is part of
        const bool take = small && (l == 0u || size * F3D_HORIZON_THETA <= dmin) && dmin >= near;
        if (take) {
            for (uint32_t s = 0u; s < kIblSectors; s++)
                if (touched & (1u << s)) best[s] = f_max(best[s], bound);
        } else if (l > 0u) {
            if (sp + 4u > 64u) return;  // (cannot happen: depth-first, at most 3 siblings wait per level)
            const uint32_t c = ((l - 1u) << 26) | ((2u * nz) << 13) | (2u * nx);
            stack[sp++] = c;
            stack[sp++] = c + 1u;
            stack[sp++] = c + (1u << 13);
            stack[sp++] = c + (1u << 13) + 1u;
        } else if (dmin >= near) {
            // a level-0 cell beyond `near` that is not small cannot exist (near >= 2.5 (cell + 2 rho)); be safe
            for (uint32_t s = 0u; s < kIblSectors; s++) best[s] = f_max(best[s], bound);
        }
        // (a level-0 cell with dmin < near <= dmax is a near cell: the march's)
    }
    for (uint32_t s = 0u; s < kIblSectors; s++) out[s] = best[s];
}

// Centre of block (bx, bz) of level B in the plane, and the height no origin on it lies below.
F3D_HD void horizon_block_frame(const TerrainDev &T, uint32_t level, uint32_t bx, uint32_t bz, float &cx, float &cz, float &y_lo) {
    cx = plane_at(T.origin_x, bx << level, T.spacing_x) + 0.5f * T.spacing_x * (float)(1u << level);
    cz = plane_at(T.origin_z, bz << level, T.spacing_z) + 0.5f * T.spacing_z * (float)(1u << level);
    y_lo = T.bands[T.band_offset[level] + (bz << T.band_shift[level]) + bx].mn - horizon_y_margin(T);
}
// One table record (k_horizon_build; the host emulator builds the same table with the same function).
F3D_HD void horizon_block_build(const TerrainDev &T, uint32_t level, uint32_t bx, uint32_t bz, float *out) {
    float cx, cz, y_lo;
    horizon_block_frame(T, level, bx, bz, cx, cz, y_lo);
    far_horizon_from(T, cx, cz, horizon_block_rho(T, level), y_lo - 1e-3f, out);
}

// The parameter after which the IBL ray (origin `o` on the terrain near surface point p, unit direction d) meets no
// terrain, or 3e38: no table, the origin is not where its block says it is (verified, not assumed), or the ray's slope
// does not clear the far horizon of its sector.
F3D_HD float ibl_stop(const TerrainDev &T, V3 o, V3 d) {
    if (!T.horizon) return 3.0e38f;
    uint32_t level = T.horizon_level;
    float spacing_x = T.spacing_x, spacing_z = T.spacing_z;
    F3D_OPAQUE_UNIFORM(level);  // (the block constants are formed in here, not hoisted in front of the sample loop: f3d_math.h)
    F3D_OPAQUE_UNIFORM(spacing_x);
    F3D_OPAQUE_UNIFORM(spacing_z);
    const float hlen = f_sqrt(d.x * d.x + d.z * d.z);
    if (!(hlen > 1e-6f)) return 3.0e38f;
    const float slope = d.y / hlen;
    if (!(slope >= 0.0f)) return 3.0e38f;  // (a descending ray is lowest at the FAR edge of a cell: not what the horizon bounds)
    uint32_t cx = sat_u32(f_floor((o.x - T.origin_x) * T.inv_spacing_x)), cz = sat_u32(f_floor((o.z - T.origin_z) * T.inv_spacing_z));
    cx = cx < T.cell_w - 1u ? cx : T.cell_w - 1u;
    cz = cz < T.cell_h - 1u ? cz : T.cell_h - 1u;
    const uint32_t bx = cx >> level, bz = cz >> level;
    float mx, mz, y_lo;
    horizon_block_frame(T, level, bx, bz, mx, mz, y_lo);
    const float rho = horizon_block_rho(spacing_x, spacing_z, level);
    const float ex = o.x - mx, ez = o.z - mz;
    if (!(ex * ex + ez * ez <= rho * rho) || !(o.y >= y_lo)) return 3.0e38f;  // the certificate's preconditions
#if defined(F3D_HORIZON_LAZY)  // host emulator: records are built when first read (NaN = not yet; racing threads write equal values)
    {
        float *rec = const_cast<float *>(T.horizon) + ((size_t)bz * T.horizon_bx + bx) * kIblSectors;
        if (rec[0] != rec[0]) {
            float tmp[kIblSectors];
            horizon_block_build(T, level, bx, bz, tmp);
            for (uint32_t k = kIblSectors; k-- > 0u;) rec[k] = tmp[k];  // rec[0] last: it is the "built" flag
        }
    }
#endif
    const float horizon = T.horizon[((size_t)bz * T.horizon_bx + bx) * kIblSectors + ibl_sector(d.x, d.z)];
    if (!(horizon < 1e30f)) return 3.0e38f;
    if (!(slope > horizon + 1e-4f * f_abs(horizon) + 1e-5f)) return 3.0e38f;
    return ibl_stop_distance(rho, f_max(spacing_x, spacing_z)) / hlen;
}

}  // namespace f3d
