#!/usr/bin/env python
"""One thin strip of the headline frame (rows 470-563 of 1080) with 16 frames in flight, 96 frames: wall time per frame, to be
run under rocprofv3 --kernel-trace --stats for the split between the trace launches and the per-frame ordered chain."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.session import TerrainSession  # noqa: E402

dem, cam, kw = datasets.rainier_proxy_scene(2048)
kw = dict(kw, spp=8, max_frames=200, min_frames=200, variance_threshold=1e30, memory_budget_bytes=8 << 30)
fd = int(sys.argv[1]) if len(sys.argv) > 1 else 16
with TerrainSession(dem, 1920, 1080, cam, row_begin=470, row_end=563, frames_in_flight=fd, **kw) as s:
    s.enqueue_frames(0, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.enqueue_frames(32, 96)
    torch.cuda.synchronize()
    print(f"fd {s.frames_in_flight()}: {(time.perf_counter() - t0) / 96 * 1e3:.4f} ms per strip-frame (93 rows)")
