"""Frames in flight on the headline scene: ms per frame and how many pixel-frames had to be traced again.
tools/fd_probe.py [frames_in_flight] [row_begin row_end] [variant]"""
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.session import TerrainSession  # noqa: E402

fd = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rows = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 1080)
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dem, cam, kw = datasets.rainier_proxy_scene(2048)
kw = dict(kw, spp=8, max_frames=256, min_frames=256, variance_threshold=1e30)
with TerrainSession(dem, 1920, 1080, cam, row_begin=rows[0], row_end=rows[1], frames_in_flight=fd, kernel_variant=variant,
                    memory_budget_bytes=40 << 30, **kw) as s:
    s.enqueue_frames(0, 2)
    torch.cuda.synchronize()
    early = s.retraced_pixels() if fd else 0
    windows = ((2, 32), (34, 32), (66, 64)) if len(sys.argv) <= 5 else tuple((2 + 16 * k, 16) for k in range(int(sys.argv[5])))
    for first, n in windows:
        t0 = time.perf_counter()
        s.enqueue_frames(first, n)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / n
        total = s.retraced_pixels() if fd else 0
        print(f"rows {rows} fd {s.frames_in_flight()} lanes {s.sample_lanes()} frames [{first}, {first + n}): {ms:.3f} ms/frame, "
              f"retraced pixel-frames so far {total} (first two frames: {early})", flush=True)
