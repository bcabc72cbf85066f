#!/usr/bin/env python
"""Single-GPU rehearsal of the multi-GPU strip balancing (forge3d_amd/distributed.py).

For N = 2, 4, 8 strips of the headline frame: time every strip of the equal partition and of
each re-balanced partition on THIS GPU (one after the other), exactly as the ranks of a real
job would, and report the compute-only bound on strong scaling T(full frame) / max_i T(strip i).
Communication (92 KB halo per neighbour and frame) is not included.
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.distributed import HALO_ROWS, HipBackend, partition_rows, rebalance, strip_rows  # noqa: E402

W, H, SPP = 1920, 1080, 8
dem, cam, kw = datasets.rainier_proxy_scene(2048)
kw = dict(kw, spp=SPP, max_frames=64, min_frames=64, variance_threshold=1e30, memory_budget_bytes=8 << 30)
if len(sys.argv) > 1:  # force a kernel variant for the strips (e.g. 4000000 = 4 sample lanes)
    kw["kernel_variant"] = int(sys.argv[1])
backend = HipBackend(0)
full = min(backend.probe(dem, W, H, cam, 0, H, kw, frames=4) for _ in range(2))
full_1lane = min(backend.probe(dem, W, H, cam, 0, H, dict(kw, kernel_variant=1000000), frames=4) for _ in range(2))
print(json.dumps({"full_frame_ms": full, "full_frame_ms_1_lane_kernel": full_1lane}))
for world in (2, 4, 8):
    bounds = [strip_rows(H, world, r)[0] for r in range(world)] + [H]
    density = np.ones(H)
    for it in range(4):
        times = [backend.probe(dem, W, H, cam, bounds[r], bounds[r + 1], kw, frames=4) for r in range(world)]
        print(json.dumps({"world": world, "round": it, "bounds": bounds, "ms": [round(t, 3) for t in times],
                          "imbalance": max(times) / (sum(times) / world),
                          "compute_bound_speedup": full / max(times)}))
        density = rebalance(density, bounds, times)
        new_bounds = partition_rows(density, world, HALO_ROWS)
        if new_bounds == bounds:
            break
        bounds = new_bounds
