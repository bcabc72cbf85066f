#!/usr/bin/env python
"""When did each wave of k_smoke run?  Needs a library built with -DF3D_SMOKE_TILE_CLOCK (tools/build_variant.sh tileclock
-DF3D_SMOKE_TILE_CLOCK), which leaves each wave's start time and duration (100 MHz ticks) in the first pixels of its tile.

    F3D_SMOKE_MARCH=single F3D_HIP_LIBRARY=build_ab/libf3dhip_tileclock.so python tools/experiments/smoke_tile_clock.py [steps=140]

(F3D_SMOKE_MARCH=single: the one-kernel form, whose waves are what this was written to look at -- the result is in
profiles/README.md, round 5, and is why the marcher became three launches.)
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from forge3d_amd import smoke  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 140
W, H = 1920, 1080
dom = smoke.SmokeDomain((96, 64, 128))
emitters = [smoke.SmokeEmitter(center=(48.0, 6.0, 40.0), radius=7.0, density_rate=9.0, temperature_rate=6.0, soot_rate=0.5,
                               emission_rate=2.0, velocity=(0.0, 2.0, 0.6))]
settings = smoke.SmokeStepSettings(dt=0.2, turbulence_strength=0.5, turbulence_seed=7, wind=(0.3, 0.0, 1.0), buoyancy=1.1)
dom.step(settings, emitters, steps=steps)
view = dict(camera_pos=(48.0, 70.0, -120.0), target=(48.0, 28.0, 64.0), up=(0.0, 1.0, 0.0), fovy_deg=40.0)
for _ in range(3):
    img = dom.render_rgba(W, H, **view)
words = np.ascontiguousarray(img).view(np.uint32).reshape(H, W)
start = words[0::8, 0::8].astype(np.int64).ravel()
dur = words[0::8, 1::8].astype(np.int64).ravel() * 10e-3  # us
where = words[0::8, 2::8].ravel()
start = ((start - start.min()) & 0xFFFFFFFF) * 10e-3
end = start + dur
print("kernel seconds (events) %.3f ms; waves %d; span of wave clocks %.1f us" % (dom.last_kernel_seconds * 1e3, dur.size, end.max()))
print("sum of wave durations %.1f ms = %.1f us per SIMD of 1024" % (dur.sum() * 1e-3, dur.sum() / 1024))
order = np.sort(dur)[::-1]
print("longest waves (us):", np.round(order[:8], 1), " waves > 100 us: %d, > 20 us: %d" % ((dur > 100).sum(), (dur > 20).sum()))
heavy = dur > 20
print("heavy waves start between %.1f and %.1f us; the last wave to end started at %.1f us and ran %.1f us" % (
    start[heavy].min(), start[heavy].max(), start[np.argmax(end)], dur[np.argmax(end)]))
hist, edges = np.histogram(end, bins=10, range=(0, end.max()))
print("waves ending per tenth of the span:", hist)
xcc = where >> 24
print("heavy waves per XCC:", np.bincount(xcc[heavy], minlength=8), " heavy time per XCC (ms):", np.round(np.bincount(xcc[heavy], weights=dur[heavy], minlength=8) * 1e-3, 2))
