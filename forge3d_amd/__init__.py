"""forge3d_amd -- MI355X-native (gfx950 / HIP) drop-in for ONE path of forge3d:
the PROMETHEUS terrain path tracer behind ``forge3d.hybrid_render_terrain_reference``.

    import forge3d_amd as f3d
    out = f3d.hybrid_render_terrain_reference(dem, 1920, 1080, camera, spp=8, ...)

Everything else forge3d offers (raster viewer, cartography, GIS, ...) is out of scope
(SURVEY.md section 8).  There is no CPU fallback: the HIP library must be built
(``__graft_entry__.build()``) and a gfx950 device must be present.
"""
from .path_tracing import hybrid_render_terrain_reference

__all__ = ["hybrid_render_terrain_reference"]
__version__ = "0.1.0"
