#!/usr/bin/env python
"""The bench's timed region on a thin strip (frames 4..35 after 4 warm-up frames) under different batch ramps of the frames
in flight (F3D_FD_FULL_FROM): ms per strip-frame and the pixel-frames k_fix had to trace again."""
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
for rows in ((470, 563), (0, 345)):
    best = 1e9
    for _ in range(3):
        with TerrainSession(dem, 1920, 1080, cam, row_begin=rows[0], row_end=rows[1], frames_in_flight=16, **kw) as s:
            s.enqueue_frames(0, 4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.enqueue_frames(4, 32)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 32 * 1e3)
            retraced = s.retraced_pixels() if hasattr(s, "retraced_pixels") else -1
    print(f"rows {rows}: {best:.4f} ms per strip-frame over frames 4..35, {retraced} pixel-frames traced again")
