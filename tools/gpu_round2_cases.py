#!/usr/bin/env python
"""Round-2 rates of the non-headline configurations for BASELINE.md: C3 (headline + AETHER post), C5 (smoke frame),
the adjudication gate of the PBR tracer, the LUT bake, C1 with frames in flight."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from forge3d_amd import atmosphere, datasets, smoke, wavefront  # noqa: E402
from forge3d_amd.session import TerrainSession  # noqa: E402

dem, cam, kw = datasets.rainier_proxy_scene(2048)
# C3: the headline render + aerial perspective (the post runs inside the resolve)
handle = atmosphere.atmosphere_bake_luts(turbidity=2.0)
print(json.dumps({"case": "AETHER LUT bake, default dimensions, 4 orders", "device_ms": handle.bake_seconds * 1e3}))
for atm in (None, handle):
    k = dict(kw, spp=8, max_frames=40, min_frames=40, variance_threshold=1e30, memory_budget_bytes=8 << 30)
    with TerrainSession(dem, 1920, 1080, cam, atmosphere=atm, **k) as s:
        s.enqueue_frames(0, 4)
        s.window_stats()
        t0 = time.perf_counter()
        s.enqueue_frames(4, 32, True)
        s.window_stats()
        loop = time.perf_counter() - t0
        t0 = time.perf_counter()
        s.resolve(36)
        res = time.perf_counter() - t0
    print(json.dumps({"case": "C3 1080p 8 spp x 32" + (" + AETHER post" if atm else " (no atmosphere)"), "loop_ms": loop * 1e3,
                      "Msamples_per_s": 1920 * 1080 * 8 * 32 / loop / 1e6, "resolve_and_readback_ms": res * 1e3}))
# C5: smoke frame at 1080p (the plume of tests/test_smoke.py::test_config5...)
import test_smoke as ts  # noqa: E402

fields = ts.plume(seed=9, dims=(96, 64, 128))
dom = ts._domain(fields)
view = dict(camera_pos=(64.0, 70.0, -120.0), target=(64.0, 28.0, 48.0), up=(0.0, 1.0, 0.0), fovy_deg=40.0)
dom.render_rgba(1920, 1080, **view)
t0 = time.perf_counter()
dom.render_rgba(1920, 1080, **view)
print(json.dumps({"case": "C5 smoke 1080p, 96x64x128 plume", "wall_ms_incl_upload_and_readback": (time.perf_counter() - t0) * 1e3,
                  "kernel_ms": dom.last_kernel_seconds * 1e3}))
# PBR tracer gate
best = min(wavefront.render_scene(wavefront.adjudication_scene(), 512, 512, 4096)["loop_seconds"] for _ in range(3))
print(json.dumps({"case": "PBR tracer adjudication gate 512x512 x 4096 frames", "kernel_ms": best * 1e3, "Gpaths_per_s": 512 * 512 * 4096 / best / 1e9}))
