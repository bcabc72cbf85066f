#!/usr/bin/env python
"""BASELINE.json configs[4] stand-in exactly as bench.py times it (120 frames of the resident smoke sequence at 1080p after 40
untimed ones), alone: ms per frame on the wall and by kernel.  A/B: F3D_HIP_LIBRARY=build_ab/libf3dhip_<name>.so,
F3D_SMOKE_SOLVER=launches.

    python tools/c5_time.py [frames=120]
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from forge3d_amd import smoke  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120
W, H = 1920, 1080
dom = smoke.SmokeDomain((96, 64, 128))
emitters = [smoke.SmokeEmitter(center=(48.0, 6.0, 40.0), radius=7.0, density_rate=9.0, temperature_rate=6.0, soot_rate=0.5,
                               emission_rate=2.0, velocity=(0.0, 2.0, 0.6))]
settings = smoke.SmokeStepSettings(dt=0.2, turbulence_strength=0.5, turbulence_seed=7, wind=(0.3, 0.0, 1.0), buoyancy=1.1)
view = dict(camera_pos=(48.0, 70.0, -120.0), target=(48.0, 28.0, 64.0), up=(0.0, 1.0, 0.0), fovy_deg=40.0)
yy, xx = np.mgrid[0:H, 0:W]
terrain = np.stack([(xx * 255 // (W - 1)), (yy * 255 // (H - 1)), np.full_like(xx, 96), np.full_like(xx, 255)], axis=-1).astype(np.uint8)
seq = smoke.SmokeSequence(dom, terrain, **view)
overlap = not os.environ.get("F3D_C5_NO_OVERLAP")  # A/B: everything on the null stream
for _ in seq.frames(40, settings, emitters, overlap=overlap):
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
for frame in seq.frames(frames, settings, emitters, overlap=overlap):
    last = frame
wall = (time.perf_counter() - t0) * 1e3 / frames
import hashlib  # noqa: E402

last = np.array(last)
kernel = {"solver_step": 0.0, "march": 0.0, "composite": 0.0}
for _ in seq.frames(16, settings, emitters, timing=True):
    for key in kernel:
        kernel[key] += seq.kernel_seconds[key]
frames_k = 16

print("C5 ms per frame %.3f  kernels: solver %.3f march %.3f composite %.3f  last frame sha %s  smoke pixels %d" % (
    wall, kernel["solver_step"] * 1e3 / frames_k, kernel["march"] * 1e3 / frames_k, kernel["composite"] * 1e3 / frames_k,
    hashlib.sha256(np.ascontiguousarray(last).tobytes()).hexdigest()[:12], int(np.count_nonzero(np.any(last[..., :3] != terrain[..., :3], axis=-1)))))
