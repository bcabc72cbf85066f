"""ViewerHandle-shaped facade over the terrain path tracer.

forge3d's ``ViewerHandle`` (reference python/forge3d/viewer.py:181-1270) remote-controls an
interactive RASTER viewer; its ``snapshot()`` is not path traced, so nothing in the reference can pin
this class -- parity is UNPINNED and stated as such in DESIGN.md.  What it offers is the same small
command surface (``load_terrain``, ``set_orbit_camera``, ``set_camera_lookat``, ``set_fov``, ``set_sun``,
``set_z_scale``, ``snapshot(path, width, height)``) routed to ``hybrid_render_terrain_reference`` with the
orbit-camera mapping of the reference's terrain viewer (src/viewer/terrain/scene.rs:110-139:
eye = target + r (sin(theta) cos(phi), cos(theta), sin(theta) sin(phi)), theta from the vertical).
World units are the DEM's own (metres when ``spacing`` is metres); the terrain is centred on the origin.
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np

from . import io as _io
from .datasets import orbit_camera
from .path_tracing import hybrid_render_terrain_reference


class OfflineTerrainViewer:
    def __init__(self, width: int = 1920, height: int = 1080, *, spp: int = 8, max_frames: int = 512,
                 min_frames: int = 32, variance_threshold: float = 1e-3, seed: int = 7):
        self.width, self.height = int(width), int(height)
        self._render = dict(spp=int(spp), max_frames=int(max_frames), min_frames=int(min_frames),
                            variance_threshold=float(variance_threshold), seed=int(seed))
        self._dem: Optional[np.ndarray] = None
        self._spacing = (1.0, 1.0)
        self._z_scale = 1.0
        self._camera = None
        self._fov = 45.0
        self._sun = (315.0, 45.0)
        self.last_result = None

    # -- scene ------------------------------------------------------------------------
    def load_terrain(self, terrain: Union[str, Path, np.ndarray], spacing: Union[float, Tuple[float, float]] = 1.0) -> None:
        dem = terrain if isinstance(terrain, np.ndarray) else _io.load_heightmap(terrain)
        if dem.ndim != 2:
            raise ValueError("heightmap must be a 2-D array")
        self._dem = np.ascontiguousarray(dem, np.float32)
        self._spacing = (float(spacing), float(spacing)) if np.isscalar(spacing) else (float(spacing[0]), float(spacing[1]))

    def set_z_scale(self, value: float) -> None:
        self._z_scale = float(value)

    def set_sun(self, azimuth_deg: float, elevation_deg: float) -> None:
        self._sun = (float(azimuth_deg), float(elevation_deg))

    # -- camera -----------------------------------------------------------------------
    def set_orbit_camera(self, phi_deg: float, theta_deg: float, radius: float, fov_deg: Optional[float] = None,
                         target: Optional[Tuple[float, float, float]] = None) -> None:
        if fov_deg is not None:
            self._fov = float(fov_deg)
        if target is None:
            top = float(self._dem.max()) * self._z_scale if self._dem is not None else 0.0
            target = (0.0, 0.5 * top, 0.0)
        self._camera = orbit_camera(tuple(float(t) for t in target), float(radius), float(phi_deg), float(theta_deg), self._fov)

    def set_camera_lookat(self, eye, target, up=(0.0, 1.0, 0.0)) -> None:
        self._camera = {"origin": tuple(map(float, eye)), "look_at": tuple(map(float, target)),
                        "up": tuple(map(float, up)), "fov_y": self._fov, "exposure": 1.0}

    def set_fov(self, deg: float) -> None:
        self._fov = float(deg)
        if self._camera is not None:
            self._camera = {**self._camera, "fov_y": self._fov}

    # -- output -----------------------------------------------------------------------
    def render(self, width: Optional[int] = None, height: Optional[int] = None) -> dict:
        if self._dem is None:
            raise RuntimeError("no terrain loaded: call load_terrain() first")
        if self._camera is None:
            span = (self._dem.shape[1] - 1) * self._spacing[0]
            self.set_orbit_camera(28.0, 49.0, 1.25 * span)
        self.last_result = hybrid_render_terrain_reference(
            self._dem, int(width or self.width), int(height or self.height), dict(self._camera, fov_y=self._fov),
            spacing=self._spacing, exaggeration=self._z_scale, sun_azimuth_deg=self._sun[0],
            sun_elevation_deg=self._sun[1], **{k: v for k, v in self._render.items() if v is not None})
        return self.last_result

    def snapshot(self, path: Union[str, Path], width: Optional[int] = None, height: Optional[int] = None) -> None:
        """Path trace the current scene and write the RGBA8 result as a PNG."""
        _io.numpy_to_png(path, self.render(width, height)["rgba"])


def render_terrain_gi(heightmap, width: int, height: int, camera=None, *, spacing=(1.0, 1.0), exaggeration: float = 1.0,
                      albedo=(0.6, 0.6, 0.6), roughness: float = 0.9, sun_azimuth_deg: float = 315.0, sun_elevation_deg: float = 45.0,
                      sun_intensity: float = 2.5, sun_color=(1.0, 0.97, 0.92), sky_color=(0.35, 0.45, 0.70), ambient_color=(0.40, 0.48, 0.62),
                      spp: int = 256, atmosphere=None, seed: int = 7, extra_spheres=(), memory_budget_bytes: int = 0) -> dict:
    """BASELINE.json configs[2]: a DEM rendered with MULTI-BOUNCE light transport ("GI") and, optionally, the AETHER
    aerial-perspective post.  Neither half is new arithmetic: the radiance is the PBR path tracer's (forge3d_amd.wavefront,
    the reference's wavefront tracer re-designed for MI355X) with the DEM as its heightfield primitive -- one sun, a sky
    environment, Lambert/GGX terrain material -- and the post is the terrain tracer's own (prometheus_aerial.wgsl through
    f3d_session_resolve) applied to that radiance over the terrain tracer's depth and hit mask for the same camera.
    The reference has no such combination (its wavefront tracer has no terrain hook); parity for it is
    the composition of the two oracles (tests/test_offline_gi.py).

    Returns dict(rgba u8 (H,W,4), hdr f32 (H,W,4) mean radiance, depth, normal, albedo, frames, gi_seconds)."""
    from .session import TerrainSession
    from .wavefront import DirectionalLight, Sphere, Terrain, WavefrontScene, render_scene

    dem = np.ascontiguousarray(heightmap, np.float32)
    cam = dict(camera or {})
    origin, look_at = cam.get("origin", (0.0, 50.0, 120.0)), cam.get("look_at", (0.0, 0.0, 0.0))
    up, fov, exposure = cam.get("up", (0.0, 1.0, 0.0)), float(cam.get("fov_y", 45.0)), float(cam.get("exposure", 1.0))
    az, el = np.deg2rad(np.float32(sun_azimuth_deg)), np.deg2rad(np.float32(sun_elevation_deg))
    to_sun = (float(np.cos(az) * np.cos(el)), float(np.sin(el)), float(np.sin(az) * np.cos(el)))  # render_terrain.rs:639-642
    spheres = [Sphere(**s) if isinstance(s, dict) else s for s in extra_spheres]
    spheres.append(Sphere(center=(0.0, -1.0e9, 0.0), radius=0.0, albedo=tuple(albedo), metallic=0.0, roughness=float(roughness)))  # the terrain's material
    scene = WavefrontScene(
        terrain=Terrain(heights=dem, spacing=tuple(spacing), exaggeration=float(exaggeration), material_id=len(spheres) - 1),
        spheres=spheres, dir_lights=[DirectionalLight(tuple(-c for c in to_sun), float(sun_intensity), tuple(sun_color), 1.0)],
        object_importance=[1.0] * len(spheres), env_ground=tuple(ambient_color), env_sky=tuple(ambient_color), miss_ground=tuple(sky_color),
        miss_sky=tuple(sky_color), cam_origin=tuple(origin), cam_look_at=tuple(look_at), cam_up=tuple(up), fov_y_deg=fov, exposure=exposure,
        seed_hi=(0x9E3779B9 ^ int(seed)) & 0xFFFFFFFF, seed_lo=0x85EBCA6B)
    # the terrain tracer's session for the same camera: depth / hit mask / AOVs of the centre rays, its resolve and post -- and
    # (round 5) its primary-ray certificates, which the PBR tracer's camera rays start from (same camera, same jitter range:
    # f3d_cone.h; F3D_GI_NO_PRIMARY_START=1 switches the hand-over off -- same image)
    import os

    kw = dict(spacing=tuple(spacing), exaggeration=float(exaggeration), albedo=tuple(albedo), sun_azimuth_deg=float(sun_azimuth_deg),
              sun_elevation_deg=float(sun_elevation_deg), sun_intensity=float(sun_intensity), sun_color=tuple(sun_color), spp=1,
              max_frames=2, min_frames=2, variance_threshold=1e30, seed=int(seed), atmosphere=atmosphere)
    with TerrainSession(dem, int(width), int(height), {"origin": origin, "look_at": look_at, "up": up, "fov_y": fov, "exposure": exposure},
                        memory_budget_bytes=int(memory_budget_bytes), **kw) as s:
        scene_dict = scene.as_dict()
        if not os.environ.get("F3D_GI_NO_PRIMARY_START"):
            scene_dict["primary_start"] = s.primary_start_ptr()
        gi = render_scene(scene_dict, int(width), int(height), int(spp))
        s.enqueue_frames(0, 2)  # (reservoirs for the resolve's validity pass; their radiance is replaced below)
        s.set_accumulation(gi["accum"])
        out = s.resolve(int(spp))
    out.update(hdr=gi["hdr"], frames=int(spp), gi_seconds=gi["loop_seconds"], path_vertices=gi["path_vertices"])
    return out
