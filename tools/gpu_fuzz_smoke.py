#!/usr/bin/env python
"""Parity fuzz of the smoke ray-marcher on the GPU: N seeded random volumes, cameras, suns and settings through
f3d_smoke_render (perspective and projection) vs oracle/smoke_oracle.c, every byte.  Aimed at what round 5 put around the
reference's loop: the clipping against the smoke's bounding box (smoke at the grid's ends, cameras inside the volume,
axis-parallel rays, voxels much smaller than their distance from the origin), the three-instruction quotients (awkward voxel
sizes), and the chunked list of the deferred self-shadow marches (long rays, a list that runs out).

    python tools/gpu_fuzz_smoke.py [first seed] [count]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from forge3d_amd import smoke  # noqa: E402
from oracle import smoke_oracle  # noqa: E402  (checker only: this is a test tool)

AWKWARD = [1.0, 0.5, 2.0, 1.0 / 3.0, 0.3, 0.7, 1.1, 7e-3, 0.015, 123.456, float(np.nextafter(np.float32(2.0), np.float32(0.0))), 3.0]


def case(seed):
    rng = np.random.default_rng(seed)
    nx, ny, nz = (int(rng.integers(2, 40)) for _ in range(3))
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    density = np.zeros((nz, ny, nx), np.float32)
    kind = int(rng.integers(0, 6))
    for _ in range(int(rng.integers(0, 5)) if kind else 0):  # kind 0: no smoke at all
        c = rng.uniform(-1, [nx, ny, nz]) + 0.5
        r = rng.uniform(0.6, 0.5 * max(nx, ny, nz))
        d = np.sqrt((x + 0.5 - c[0]) ** 2 + (y + 0.5 - c[1]) ** 2 + (z + 0.5 - c[2]) ** 2)
        density += (rng.uniform(0.05, 3.0) * np.clip(1.0 - d / r, 0.0, 1.0)).astype(np.float32)
    if kind == 1:  # a face of the grid
        density[:, :, 0 if rng.random() < 0.5 else -1] += np.float32(rng.uniform(0.1, 1.0))
    if kind == 2:  # lone voxels
        density[:] = 0
        for _ in range(3):
            density[int(rng.integers(nz)), int(rng.integers(ny)), int(rng.integers(nx))] = np.float32(rng.uniform(0.5, 4.0))
    fields = {"density": density, "soot": (rng.uniform(0, 0.6) * density * (rng.random(density.shape) < 0.7)).astype(np.float32),
              "temperature": (rng.uniform(0, 2.0) * density).astype(np.float32), "humidity": (rng.random(density.shape) * (density > 0.02)).astype(np.float32),
              "emission_rate": (rng.uniform(0, 2.0) * density * (y < ny * 0.3)).astype(np.float32),
              "particle_age": np.where(density > 1e-5, rng.uniform(0, 25.0) * y / max(1, ny), -1.0).astype(np.float32)}
    vs = tuple(float(np.float32(AWKWARD[int(rng.integers(len(AWKWARD)))] * (1.0 if rng.random() < 0.7 else rng.uniform(0.5, 2.0)))) for _ in range(3))
    og = tuple(float(v) for v in (rng.uniform(-50, 50, 3) if rng.random() < 0.7 else rng.uniform(-1, 1, 3) * 4000.0 * max(vs)))
    ext = np.array([nx, ny, nz]) * np.array(vs)
    centre = np.array(og) + 0.5 * ext
    if rng.random() < 0.3:  # camera inside the volume
        pos = np.array(og) + rng.uniform(0.05, 0.95, 3) * ext
    else:
        direction = rng.normal(size=3)
        pos = centre + direction / np.linalg.norm(direction) * rng.uniform(0.7, 3.0) * np.linalg.norm(ext)
    target = centre + rng.uniform(-0.3, 0.3, 3) * ext
    if rng.random() < 0.15:  # an axis-parallel central ray
        axis = int(rng.integers(3))
        pos = centre.copy()
        pos[axis] -= rng.uniform(0.8, 2.0) * ext[axis]
        target = centre.copy()
    sun = rng.normal(size=3)
    if rng.random() < 0.2:
        sun = np.eye(3)[int(rng.integers(3))] * (1 if rng.random() < 0.5 else -1)
    view = rng.normal(size=3)
    if rng.random() < 0.3:
        view = np.array([0.0, -1.0, 0.0])
    up = (0.0, 1.0, 0.0) if abs((target - pos)[1]) < 0.98 * np.linalg.norm(target - pos) else (1.0, 0.0, 0.0)
    st = dict(step_size=float(rng.uniform(0.2, 1.5) * min(vs)) if rng.random() < 0.8 else 0.0,
              shadow_step_size=float(rng.uniform(0.3, 3.0) * min(vs)) if rng.random() < 0.8 else 0.0,
              shadow_steps=int(rng.integers(1, 40)), max_steps=int(rng.choice([3, 17, 64, 200, 700])), self_shadow=bool(rng.random() < 0.85),
              jitter_strength=float(rng.uniform(0, 1)), density_scale=float(rng.uniform(0.3, 3.0)), extinction=float(rng.uniform(0.2, 4.0)),
              phase_g=float(rng.uniform(-0.8, 0.8)), soot_absorption=float(rng.uniform(0, 1.0)), fire_glow=float(rng.uniform(0, 2.0)))
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 50))
    return fields, vs, og, tuple(pos), tuple(target), up, float(rng.uniform(15, 100)), tuple(sun), tuple(view), st, w, h, int(rng.integers(0, 1000))


first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
bad, t0, smoky = [], time.time(), 0
for seed in range(first, first + count):
    fields, vs, og, pos, target, up, fov, sun, view, st, w, h, frame = case(seed)
    if seed % 5 == 0:
        os.environ["F3D_SMOKE_SHADOW_SLOTS"] = "2048"  # a list with room for two chunks
    else:
        os.environ.pop("F3D_SMOKE_SHADOW_SLOTS", None)
    d = fields["density"]
    dom = smoke.SmokeDomain((d.shape[2], d.shape[1], d.shape[0]), vs, og)
    dom.set_density(d)
    dom.set_temperature(fields["temperature"]), dom.set_soot(fields["soot"]), dom.set_humidity(fields["humidity"])
    dom.set_emission(fields["emission_rate"]), dom.set_particle_age(fields["particle_age"])
    dom.frame_index = frame
    settings = smoke.SmokeRenderSettings(**st)
    got = dom.render_rgba(w, h, pos, target, up=up, fovy_deg=fov, sun_direction=sun, settings=settings)
    want = smoke_oracle.render_rgba(fields, w, h, pos, target, up=up, fovy_deg=fov, sun_direction=sun, voxel_size=vs, origin=og, frame_index=frame, **st)
    got_p = dom.render_projection_rgba(w, h, view, sun, settings=settings)
    want_p = smoke_oracle.render_projection_rgba(fields, w, h, view, sun, voxel_size=vs, origin=og, frame_index=frame, **st)
    smoky += int(want[..., 3].any()) + int(want_p[..., 3].any())
    if not np.array_equal(got, want):
        bad.append((seed, "perspective", int((got != want).any(-1).sum())))
    if not np.array_equal(got_p, want_p):
        bad.append((seed, "projection", int((got_p != want_p).any(-1).sum())))
print(f"{count} smoke scenes x 2 views from seed {first}: {len(bad)} mismatches {bad[:10]}, {smoky} of {2 * count} images show smoke, {time.time() - t0:.1f} s")
