#!/usr/bin/env python
"""When does every wave of the frame kernel run?  (needs a library built with -DF3D_WAVE_TIMES)

    OPT="-O3 -DF3D_WAVE_TIMES" ... build, then: python tools/wave_times.py [row_begin row_end]

Renders 3 frames of the headline scene (whole frame, or the given strip), takes the {start, end} clocks of the
last frame's workgroups and prints the duration distribution, the kernel span, what the chip's wave slots could
have done, and a longest-first replay (list scheduling on 6144 slots) -- is the tail worth attacking?  The raw
times go to gpurun_out/wave_times_<tag>.npy."""
import heapq
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.session import TerrainSession  # noqa: E402

W, H = 1920, 1080
rb, re = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, H)
dem, cam, kw = datasets.rainier_proxy_scene(2048)
kw = dict(kw, spp=8, max_frames=8, min_frames=8, variance_threshold=1e30, memory_budget_bytes=8 << 30)
buf = torch.zeros(2 * 400000, dtype=torch.int64, device="cuda")
with TerrainSession(dem, W, H, cam, row_begin=rb, row_end=re, bands=1, **kw) as s:
    assert s._lib.f3d_session_debug_wave_times(s._handle, buf.data_ptr()) == 0, "build with -DF3D_WAVE_TIMES"
    s.enqueue_frames(0, 3)
    torch.cuda.synchronize()
    lanes = s.sample_lanes()
t = buf.cpu().numpy().reshape(-1, 2)
t = t[t[:, 1] > 0].astype(np.float64)
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0  # microseconds (100 MHz clock)
dur = end - start
tag = f"{rb}_{re}"
np.save(ROOT / "gpurun_out" / f"wave_times_{tag}.npy", np.stack([start, end], 1))
slots = 256 * 4 * 6
work = dur.sum()
print(f"rows [{rb},{re}) lanes {lanes}: {len(dur)} waves, kernel span {end.max():.1f} us, sum of wave durations "
      f"{work / 1e3:.1f} ms -> {work / slots:.1f} us on {slots} slots if perfectly packed")
print("wave duration us: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(dur, [10, 50, 90, 99, 100])))
print("last start %.1f us; waves still running at 50/75/90%% of the span: %s" % (
    start.max(), [int(((start <= f * end.max()) & (end > f * end.max())).sum()) for f in (0.5, 0.75, 0.9)]))
for name, order in (("as dispatched", np.argsort(start, kind="stable")), ("longest first", np.argsort(-dur, kind="stable"))):
    heap = [0.0] * slots
    heapq.heapify(heap)
    finish = 0.0
    for d in dur[order]:
        s0 = heapq.heappop(heap)
        heapq.heappush(heap, s0 + d)
        finish = max(finish, s0 + d)
    print(f"list scheduling, {name}: {finish:.1f} us (durations taken as measured, i.e. at the measured occupancy)")
