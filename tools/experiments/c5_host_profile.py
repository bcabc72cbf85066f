#!/usr/bin/env python
"""Where a frame of the resident smoke sequence spends its HOST time: perf_counter around each library call and torch
operation of SmokeSequence.frames (untimed, asynchronous calls), averaged over 120 frames."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from forge3d_amd import smoke  # noqa: E402

W, H = 1920, 1080
dom = smoke.SmokeDomain((96, 64, 128))
emitters = [smoke.SmokeEmitter(center=(48.0, 6.0, 40.0), radius=7.0, density_rate=9.0, temperature_rate=6.0, soot_rate=0.5,
                               emission_rate=2.0, velocity=(0.0, 2.0, 0.6))]
settings = smoke.SmokeStepSettings(dt=0.2, turbulence_strength=0.5, turbulence_seed=7, wind=(0.3, 0.0, 1.0), buoyancy=1.1)
view = dict(camera_pos=(48.0, 70.0, -120.0), target=(48.0, 28.0, 64.0), up=(0.0, 1.0, 0.0), fovy_deg=40.0)
yy, xx = np.mgrid[0:H, 0:W]
terrain = np.stack([(xx * 255 // (W - 1)), (yy * 255 // (H - 1)), np.full_like(xx, 96), np.full_like(xx, 255)], axis=-1).astype(np.uint8)
seq = smoke.SmokeSequence(dom, terrain, **view)
for _ in seq.frames(40, settings, emitters):
    pass
torch.cuda.synchronize()
acc = {"step": 0.0, "render+composite": 0.0, "copy enqueue": 0.0, "wait previous copy": 0.0}
N = 120
t_all = time.perf_counter()
pending = None
for _ in range(N):
    t0 = time.perf_counter()
    seq.step(settings, emitters, steps=1)
    t1 = time.perf_counter()
    image = seq.render_to_device()
    t2 = time.perf_counter()
    turn = seq._turn
    with torch.cuda.stream(seq.copy_stream):
        seq.copy_stream.wait_event(seq.rendered[turn])
        seq.pinned[turn].copy_(image, non_blocking=True)
        seq.copied[turn].record()
    seq._turn ^= 1
    t3 = time.perf_counter()
    if pending is not None:
        seq.copied[pending].synchronize()
    pending = turn
    t4 = time.perf_counter()
    acc["step"] += t1 - t0
    acc["render+composite"] += t2 - t1
    acc["copy enqueue"] += t3 - t2
    acc["wait previous copy"] += t4 - t3
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / N * 1e3
print("wall ms per frame %.3f; host ms per frame: %s" % (wall, {k: round(v / N * 1e3, 3) for k, v in acc.items()}))
