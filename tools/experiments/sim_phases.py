#!/usr/bin/env python
"""Per-phase times of the persistent smoke solver (k_sim_step) from a -DF3D_SIM_PHASE_TIMES build: for workgroup 0, how long
each phase's own voxel loop took (arrival at barrier k minus departure from barrier k - 1) and how long it then stood at the
barrier.  F3D_HIP_LIBRARY=build_ab/libf3dhip_simphase.so python tools/experiments/sim_phases.py"""
import os
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from forge3d_amd import smoke  # noqa: E402

dom = smoke.SmokeDomain((96, 64, 128))
emitters = [smoke.SmokeEmitter(center=(48.0, 6.0, 40.0), radius=7.0, density_rate=9.0, temperature_rate=6.0, soot_rate=0.5,
                               emission_rate=2.0, velocity=(0.0, 2.0, 0.6))]
settings = smoke.SmokeStepSettings(dt=0.2, turbulence_strength=0.5, turbulence_seed=7, wind=(0.3, 0.0, 1.0), buoyancy=1.1)
dom.step(settings, emitters, steps=40)
path = tempfile.mktemp(suffix=".phases")
os.environ["F3D_SIM_PHASE_FILE"] = path
os.environ["F3D_SMOKE_SOLVER"] = "persistent"
dom.step(settings, emitters, steps=1)
print("device ms of the step:", dom.last_kernel_seconds * 1e3)
rows = [tuple(int(x) for x in line.split()) for line in open(path)]
prev = None
total_work = total_wait = 0.0
for k, arrive, leave in rows:
    work = (arrive - prev) / 100.0 if prev is not None else float("nan")
    wait = (leave - arrive) / 100.0
    print(f"barrier {k:3d}: phase before it {work:8.2f} us, at the barrier {wait:8.2f} us")
    if prev is not None:
        total_work += work
    total_wait += wait
    prev = leave
print(f"sum of phases {total_work:.1f} us, sum of waits {total_wait:.1f} us over {len(rows)} barriers")
