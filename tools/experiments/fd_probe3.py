"""Frames in flight on small images (BASELINE configs[0]: 512 x 512, 16 spp): tools/fd_probe3.py"""
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.session import TerrainSession  # noqa: E402

dem, cam, kw = datasets.rainier_proxy_scene(2048)
for (w, h, spp) in ((512, 512, 16), (256, 256, 8), (1024, 768, 8)):
    k = dict(kw, spp=spp, max_frames=256, min_frames=256, variance_threshold=1e30)
    for fd in (0, 3, 8, 16):
        with TerrainSession(dem, w, h, cam, frames_in_flight=fd, memory_budget_bytes=16 << 30, **k) as s:
            s.enqueue_frames(0, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.enqueue_frames(1, 32)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / 32
            print(f"{w}x{h} spp {spp} frames_in_flight {s.frames_in_flight()} lanes {s.sample_lanes()}: {ms:.3f} ms/frame = {w * h * spp / ms / 1e3:.0f} Msamples/s", flush=True)
