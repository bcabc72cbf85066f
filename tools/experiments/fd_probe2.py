import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from forge3d_amd import datasets
from forge3d_amd.distributed import HipBackend
dem, cam, kw = datasets.rainier_proxy_scene(2048)
kw = dict(kw, spp=8, max_frames=64, min_frames=64, variance_threshold=1e30, memory_budget_bytes=8 << 30)
b = HipBackend(0)
for fd in (0, 16):
    for rows in ((575, 653), (624, 703)):
        for frames in (32, 64):
            ms = min(b.probe(dem, 1920, 1080, cam, rows[0], rows[1], dict(kw, frames_in_flight=fd), frames=frames) for _ in range(2))
            print(f"backend.probe fd {fd} rows {rows} frames {frames}: {ms:.3f} ms/frame", flush=True)
