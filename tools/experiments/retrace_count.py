import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import scenes, numpy as np
from forge3d_amd.session import TerrainSession
dem = scenes.golden_dem()
for az, el in [(302.0, 24.0), (135.0, 12.0), (17.0, 61.0), (250.0, 3.0)]:
    kw = scenes.fixed_frames(dict(scenes.scene_kwargs(dem), sun_azimuth_deg=az, sun_elevation_deg=el), 40, spp=4)
    with TerrainSession(dem, 240, 180, scenes.CAM, frames_in_flight=16, **kw) as s:
        s.enqueue_frames(0, 40, True)
        s.window_stats()
        print(az, el, "retraced", s.retraced_pixels(), "of", 240*180*40)
