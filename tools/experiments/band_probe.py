#!/usr/bin/env python
"""ms per frame of the whole headline frame and of one thin strip for a few (bands, streams) settings."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.distributed import HipBackend  # noqa: E402

W, H = 1920, 1080
dem, cam, kw = datasets.rainier_proxy_scene(2048)
kw = dict(kw, spp=8, max_frames=64, min_frames=64, variance_threshold=1e30, memory_budget_bytes=8 << 30)
backend = HipBackend(0)
for rows in ((0, H), (624, 703)):
    for bands, streams in ((1, 0), (2, 2), (3, 3), (4, 4)):
        ms = min(backend.probe(dem, W, H, cam, rows[0], rows[1], dict(kw, bands=bands, band_streams=streams), frames=16)
                 for _ in range(2))
        print(json.dumps({"rows": rows, "bands": bands, "streams": streams, "ms": round(ms, 3)}), flush=True)
