"""forge3d_amd -- MI355X-native (gfx950 / HIP) drop-in for ONE path of forge3d:
the PROMETHEUS terrain path tracer behind ``forge3d.hybrid_render_terrain_reference``.

    import forge3d_amd as f3d
    out = f3d.hybrid_render_terrain_reference(dem, 1920, 1080, camera, spp=8, ...)   # dict of numpy arrays
    clean = f3d.denoise.atrous_denoise(out["rgba"][..., :3] / 255.0, albedo=out["albedo"], normal=out["normal"])
    f3d.numpy_to_png("frame.png", out["rgba"])

    plume = f3d.smoke.domain_from_density(density_zyx); rgba = plume.render_rgba(1920, 1080, eye, target)  # smoke ray-marcher

    pt_rgba, meta = f3d.render_adjudication_pt(512, 512, 4096)   # multi-bounce PBR tracer (the adjudication gate's PT half)

    viewer = f3d.open_viewer_async(1920, 1080, terrain_path="dem.tif")   # ViewerHandle names, offline (parity unpinned)
    viewer.set_orbit_camera(28, 49, 25_000); viewer.set_sun(302, 24); viewer.snapshot("a.png")

Everything else forge3d offers (raster viewer, cartography, GIS, ...) is out of scope
(SURVEY.md section 8).  There is no CPU fallback: the HIP library must be built
(``__graft_entry__.build()``) and a gfx950 device must be present.
"""
from . import atmosphere, denoise, io, offline, smoke, viewer, wavefront  # noqa: F401
from .io import numpy_to_png, png_to_numpy
from .atmosphere import hybrid_render_aether_spectral_reference
from .path_tracing import hybrid_render_terrain_reference
from .viewer import Renderer, ViewerHandle, open_viewer, open_viewer_async
from .wavefront import render_adjudication_pt

__all__ = ["hybrid_render_terrain_reference", "hybrid_render_aether_spectral_reference", "atmosphere", "denoise", "io", "offline", "smoke", "viewer", "numpy_to_png",
           "png_to_numpy", "wavefront", "render_adjudication_pt", "Renderer", "ViewerHandle", "open_viewer", "open_viewer_async"]
__version__ = "0.1.0"
