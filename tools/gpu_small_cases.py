#!/usr/bin/env python
"""Loop-only GPU rates of BASELINE.md's small inputs S0 (golden scene, 1 spp until converged)
and S1 (512x512, 16 spp x 2 frames and 1 spp x 16 frames) through the one-shot C ABI."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import forge3d_amd as f3d  # noqa: E402
import scenes  # noqa: E402

dem = scenes.golden_dem()
kw = scenes.scene_kwargs(dem)
cases = [
    ("S0 256x256 1spp until converged", 256, 256, dict(kw)),
    ("S1 512x512 16spp x 2", 512, 512, scenes.fixed_frames(kw, 2, spp=16)),
    ("S1 512x512 1spp x 16", 512, 512, scenes.fixed_frames(kw, 16, spp=1)),
]
for name, w, h, k in cases:
    best = None
    for _ in range(3):
        out = f3d.hybrid_render_terrain_reference(dem, w, h, scenes.CAM, **k)
        n = w * h * int(k.get("spp", 1)) * int(out["frames"])
        rate = n / out["loop_seconds"] / 1e6
        best = rate if best is None else max(best, rate)
    print(json.dumps({"case": name, "frames": int(out["frames"]), "Msamples": n / 1e6,
                      "loop_ms": out["loop_seconds"] * 1e3, "Msamples_per_s_loop_best_of_3": best}))
