"""``ViewerHandle`` / ``open_viewer`` / ``Renderer`` names of the reference, routed to the terrain path tracer.

forge3d's ``ViewerHandle`` (reference python/forge3d/viewer.py:181-1383) remote-controls an interactive RASTER
viewer process over IPC and ``Renderer`` (python/forge3d/__init__.py:347-421) is a CPU triangle stub; neither is
path traced, so no golden of the reference can pin what ``snapshot()`` draws here -- parity for this facade is
UNPINNED.  What carries over is the calling surface a script written against forge3d uses for an
offline render: ``open_viewer_async(width, height, terrain_path=..., fov_deg=...)`` -> ``ViewerHandle`` with
``load_terrain``, ``set_orbit_camera``, ``set_camera_lookat``, ``set_fov``, ``set_sun``, ``set_sun_time``, ``set_ibl``,
``set_z_scale``, ``snapshot(path, width, height)``, ``render_animation``, ``get_stats``, ``close`` and the context
manager.  There is no subprocess and no window: every snapshot is a converged path-traced frame on the MI355X.
Commands of the raster viewer that have no meaning offline (labels, picking, overlays, point clouds) raise
``ViewerError`` instead of being ignored.
"""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple, Union

import numpy as np

from . import io as _io
from .offline import OfflineTerrainViewer


class ViewerError(Exception):
    """reference python/forge3d/viewer.py:160-162"""


_RASTER_ONLY = ("load_obj", "load_gltf", "load_bundle", "load_overlay", "load_point_cloud", "set_point_cloud_params",
                "set_transform", "add_label", "add_labels", "add_line_label", "add_curved_label", "add_callout",
                "add_vector_overlay", "set_labels_enabled", "clear_labels", "remove_label", "set_label_typography",
                "set_declutter_algorithm", "poll_pick_events", "pick_at", "update_labels", "load_label_atlas",
                "set_terrain_scatter", "clear_terrain_scatter", "apply_scene_variant", "set_review_layer_visible")


class ViewerHandle(OfflineTerrainViewer):
    """Offline stand-in for the reference's ViewerHandle: same method names and argument order."""

    def __init__(self, width: int = 1280, height: int = 720, *, fov_deg: float = 60.0, spp: int = 8, **render):
        super().__init__(width, height, spp=spp, **render)
        self._fov = float(fov_deg)
        self._env = None
        self._env_intensity = 0.35
        self._sun_intensity = 2.5
        self._revision = 0
        self._open = True

    # -- scene --------------------------------------------------------------------------------------------
    def load_terrain(self, path: Union[str, Path, np.ndarray], spacing: Union[float, Tuple[float, float], None] = None) -> None:
        """reference viewer.py:978-986 takes a DEM path; arrays are accepted too.  Spacing defaults to the
        GeoTIFF pixel scale when the file carries one, else 1."""
        if spacing is None and not isinstance(path, np.ndarray) and str(path).lower().endswith((".tif", ".tiff")):
            scale = _io.read_geotiff(path)[1]["pixel_scale"]
            spacing = (float(scale[0]), float(scale[1])) if scale else 1.0
        super().load_terrain(path, 1.0 if spacing is None else spacing)
        self._revision += 1

    def set_sun(self, azimuth_deg: float, elevation_deg: float) -> None:
        super().set_sun(azimuth_deg, elevation_deg)
        self._revision += 1

    def set_sun_time(self, solar_time: Any, intensity: float = 1.0) -> None:
        """reference viewer.py:1128-1145: the apparent position of a forge3d.geo.SolarTime-like object."""
        from .geo import resolve_solar_time

        position = resolve_solar_time(solar_time)
        super().set_sun(float(position["azimuth_deg"]), float(position["apparent_elevation_deg"]))
        self._sun_intensity = 2.5 * float(intensity)
        self._revision += 1

    def set_ibl(self, path: Union[str, Path, np.ndarray], intensity: float = 1.0) -> None:
        """Environment map: a Radiance .hdr / .rgbe file (reference viewer.py:1147, loader src/formats/hdr.rs), an
        (H, W, 3) float32 array, or a .npy file of one."""
        if isinstance(path, np.ndarray):
            env = path
        elif Path(path).suffix.lower() in (".hdr", ".rgbe"):
            try:
                env = _io.read_hdr(path)
            except (OSError, _io.HdrError) as exc:
                raise ViewerError(str(exc)) from exc
        elif Path(path).suffix.lower() == ".npy":
            env = np.load(Path(path))
        else:
            raise ViewerError(f"Unsupported environment map format '{Path(path).suffix}' for '{path}': expected .hdr, .rgbe or .npy")
        env = np.ascontiguousarray(env, np.float32)
        if env.ndim != 3 or env.shape[2] != 3:
            raise ViewerError(f"environment map must be (H, W, 3) float32, got {env.shape}")
        self._env, self._env_intensity = env, float(intensity)
        self._revision += 1

    # -- output -------------------------------------------------------------------------------------------
    def render(self, width: Optional[int] = None, height: Optional[int] = None) -> dict:
        if not self._open:
            raise ViewerError("viewer is closed")
        self._render.update(env_map=self._env, env_intensity=self._env_intensity, sun_intensity=self._sun_intensity)
        return super().render(width, height)

    def render_animation(self, animation: Sequence[Mapping[str, Any]], output_dir: Union[str, Path], fps: int = 30,
                         width: Optional[int] = None, height: Optional[int] = None, progress_callback=None) -> None:
        """A sequence of camera keyframes ({"phi_deg", "theta_deg", "radius"[, "fov_deg", "target"]} per frame),
        one PNG per frame named frame_0000.png ... like the reference's exporter (viewer.py:1270-1334); the DEM's
        acceleration tables are built once (the library's scene cache)."""
        out = Path(output_dir)
        out.mkdir(parents=True, exist_ok=True)
        for i, key in enumerate(animation):
            self.set_orbit_camera(key["phi_deg"], key["theta_deg"], key["radius"], key.get("fov_deg"), key.get("target"))
            self.snapshot(out / f"frame_{i:04d}.png", width, height)
            if progress_callback:
                progress_callback(i, len(animation))

    def get_stats(self) -> Dict[str, Any]:
        last = self.last_result or {}
        return {"applied_command_revision": self._revision, "rendered_revision": self._revision, "frames": last.get("frames"),
                "variance": last.get("variance"), "gpu_resource_bytes": last.get("gpu_resource_bytes"), "backend": "hip-gfx950"}

    def send_ipc(self, cmd: Dict[str, Any]) -> Dict[str, Any]:
        raise ViewerError("the offline path tracer has no IPC channel; call the methods directly")

    def close(self) -> None:
        self._open = False

    def __enter__(self) -> "ViewerHandle":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    @property
    def is_running(self) -> bool:
        return self._open

    def __getattr__(self, name):
        if name in _RASTER_ONLY:
            def refuse(*_a, **_k):
                raise ViewerError(f"ViewerHandle.{name} belongs to the interactive raster viewer; the offline path "
                                  "tracer has no counterpart")
            return refuse
        raise AttributeError(name)


def open_viewer_async(width: int = 1280, height: int = 720, title: str = "forge3d Interactive Viewer", obj_path=None,
                      gltf_path=None, terrain_path=None, fov_deg: float = 60.0, timeout: float = 30.0,
                      ipc_host: str = "127.0.0.1", ipc_port: int = 0) -> ViewerHandle:
    """reference viewer.py:1390-1517 -- returns at once with a handle; here there is nothing to launch."""
    _ = title, timeout, ipc_host, ipc_port
    if obj_path is not None or gltf_path is not None:
        raise ViewerError("the offline path tracer renders terrain (and meshes passed to hybrid_render_terrain_reference); "
                          "OBJ / glTF scenes belong to the raster viewer")
    handle = ViewerHandle(width, height, fov_deg=fov_deg)
    if terrain_path is not None:
        handle.load_terrain(terrain_path)
    return handle


def open_viewer(*args, **kwargs) -> ViewerHandle:
    """reference viewer.py:1519-: the blocking variant; offline there is nothing to block on."""
    return open_viewer_async(*args, **kwargs)


class Renderer:
    """``forge3d.Renderer(width, height)`` (reference python/forge3d/__init__.py:347-421): the reference's class is a
    deterministic CPU stub (`render_triangle_rgba`); its constructor / ``get_config`` shape is kept and
    ``render_terrain`` routes a DEM to the path tracer."""

    def __init__(self, width: int, height: int, *, config: "Mapping[str, Any] | None" = None, **kwargs: Any) -> None:
        self.width, self.height = int(width), int(height)
        allowed = {"exposure", "spp", "max_frames", "min_frames", "variance_threshold", "seed"}
        unexpected = sorted(k for k in kwargs if k not in allowed)
        if unexpected:
            raise TypeError(f"Unexpected arguments: {', '.join(unexpected)}")
        self._config = {"backend": "hip-gfx950", "lighting": {"exposure": float(kwargs.pop("exposure", 1.0))},
                        "path_tracing": {"spp": 8, "max_frames": 512, "min_frames": 32, "variance_threshold": 1e-3, "seed": 7}}
        if config:
            self._config.update({k: v for k, v in dict(config).items()})
        self._config["path_tracing"].update(kwargs)

    def get_config(self) -> dict:
        return {k: (dict(v) if isinstance(v, dict) else v) for k, v in self._config.items()}

    def render_triangle_rgba(self, *, certificate=False, cache=None) -> np.ndarray:
        """The reference's deterministic test pattern (__init__.py:382-407), vectorised."""
        y, x = np.mgrid[0:self.height, 0:self.width]
        cx, cy, size = self.width // 2, self.height // 2, min(self.width, self.height) // 4
        inside = (np.abs(x - cx) + np.abs(y - cy) < size) & (y > cy - size // 2)
        img = np.empty((self.height, self.width, 4), np.uint8)
        img[...] = (16, 16, 24, 255)
        img[inside] = (128, 64, 32, 255)
        return img

    def render_triangle_png(self, path, *, certificate=False, cache=None) -> None:
        _io.numpy_to_png(path, self.render_triangle_rgba())

    def render_terrain(self, heightmap, camera=None, **kwargs) -> dict:
        from .path_tracing import hybrid_render_terrain_reference

        cam = dict(camera or {})
        cam.setdefault("exposure", self._config["lighting"]["exposure"])
        return hybrid_render_terrain_reference(heightmap, self.width, self.height, cam,
                                               **{**self._config["path_tracing"], **kwargs})
