#!/usr/bin/env python
"""BASELINE.md input S4 (stand-in for BASELINE.json config 4): rainier-proxy DEM + 50 000 extruded
boxes (600 000 triangles) through the mesh BVH; loop-only Msamples/s at 1080p and 4096^2, 8 spp."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.session import TerrainSession  # noqa: E402

dem, cam, kw = datasets.rainier_proxy_scene(2048)
t0 = time.perf_counter()
v, i = datasets.proxy_buildings(dem, 10.0)
t_gen = time.perf_counter() - t0
VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for (w, h, frames) in ((1920, 1080, 16), (4096, 4096, 8)):
    k = dict(kw, spp=8, kernel_variant=VARIANT, max_frames=frames + 2, min_frames=frames + 2, variance_threshold=1e30,
             mesh_vertices=v, mesh_indices=i, memory_budget_bytes=16 << 30)
    t0 = time.perf_counter()
    with TerrainSession(dem, w, h, cam, **k) as s:
        t_setup = time.perf_counter() - t0
        s.enqueue_frames(0, 2)
        s.window_stats()
        t0 = time.perf_counter()
        s.enqueue_frames(2, frames, True)
        s.window_stats()
        dt = time.perf_counter() - t0
        out = s.resolve(frames + 2)
        mesh_px = float((out["albedo"][..., 2] > 0.65).mean())
        print(json.dumps({"case": f"S4 {w}x{h} 8spp x {frames}", "triangles": int(i.shape[0]), "setup_s": t_setup,
                          "Msamples_per_s": w * h * 8 * frames / dt / 1e6, "ms_per_frame": dt / frames * 1e3,
                          "mesh_pixel_fraction": mesh_px, "sample_lanes": s.sample_lanes(),
                          "gpu_resource_bytes": s.info()["gpu_resource_bytes"]}))
