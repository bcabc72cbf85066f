#!/usr/bin/env python
"""How long the HOST needs to enqueue a frame of the resident smoke sequence (perf_counter around untimed, asynchronous
library calls, no synchronisation inside the loop) against how long the device needs to run it."""
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
terrain = np.zeros((H, W, 4), np.uint8)
seq = smoke.SmokeSequence(dom, terrain, **view)
for _ in seq.frames(100, settings, emitters):
    pass
N = 100
for what, call in (("step", lambda: seq.step(settings, emitters, steps=1)), ("render + composite", seq.render_to_device)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        call()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-20s host enqueue %.3f ms a call, with the device's tail %.3f ms a call" % (what, (t1 - t0) * 1e3 / N, (t2 - t0) * 1e3 / N))
