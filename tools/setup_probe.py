#!/usr/bin/env python
"""Session set-up by phase (f3d_session_setup_ms) for the headline scene: a fresh DEM, then the same DEM from the scene
cache, then another fresh DEM -- the first session of a process also pays module load and allocator start-up.
    python tools/setup_probe.py"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.session import TerrainSession  # noqa: E402

dem, cam, kw = datasets.rainier_proxy_scene(2048)
kw = dict(kw, spp=8, variance_threshold=1e30, max_frames=32, min_frames=32, memory_budget_bytes=8 << 30)
n_warm = int(sys.argv[1]) if len(sys.argv) > 1 else 64
warm = np.zeros((n_warm, n_warm), np.float32)
warm[::3, ::5] = 1.0
with TerrainSession(warm, 64, 64, cam, **dict(kw, max_frames=2, min_frames=2)) as ws:
    ws.enqueue_frames(0, 2)
torch.cuda.synchronize()
for label, d in (("fresh 2048^2 DEM", dem), ("same DEM (scene cache)", dem), ("fresh DEM again", dem + np.float32(1.0)), ("cached again", dem + np.float32(1.0))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = TerrainSession(d, 1920, 1080, cam, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ms = s.setup_ms()
    s.enqueue_frames(0, 4)
    torch.cuda.synchronize()
    s.close()
    print("%-26s python wall %.2f ms (+ %.2f ms until the passes have run) | " % (label, (t1 - t0) * 1e3, (t2 - t1) * 1e3)
          + "  ".join("%s %.2f" % (k, v) for k, v in ms.items()))
