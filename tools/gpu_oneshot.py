#!/usr/bin/env python
"""PCIe-inclusive timing of the one-shot C ABI on the headline configuration."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import forge3d_amd as f3d  # noqa: E402
from forge3d_amd import _native, datasets  # noqa: E402

dem, cam, kw = datasets.rainier_proxy_scene(2048)
k = dict(kw, spp=8, max_frames=32, min_frames=32, variance_threshold=1e30)
orig = _native.lib().f3d_terrain_ref_render
for attempt in range(4):
    t0 = time.perf_counter()
    try:
        out = f3d.hybrid_render_terrain_reference(dem, 1920, 1080, cam, **k)
    except RuntimeError as exc:  # default 512 MiB budget of the reference
        print("default budget:", str(exc)[:160])
        break
    wall = time.perf_counter() - t0
    n = 1920 * 1080 * 8 * 32
    print(json.dumps({"attempt": attempt, "wall_s": wall, "loop_s": out["loop_seconds"], "setup_s": out["setup_seconds"],
                      "readback_s": out["readback_seconds"], "Msamples_per_s_loop": n / out["loop_seconds"] / 1e6,
                      "Msamples_per_s_pcie_inclusive": n / wall / 1e6, "gpu_resource_bytes": out["gpu_resource_bytes"],
                      "minmax_pyramid_bytes": out["minmax_pyramid_bytes"]}))
