#!/usr/bin/env python
"""Parity fuzz of the PBR tracer's HEIGHTFIELD primitive on the GPU (BASELINE.json configs[2]'s "GI": closest-hit and shadow
rays through the terrain tracer's march, shared over the wave's lanes, four lanes a pixel): N seeded random scenes -- ragged
DEMs with terraces, unequal spacings, cameras inside and outside the footprint, one or two suns, spheres on the terrain, odd
image sizes and frame counts, renders continued from an accumulation -- through f3d_wavefront_render vs
oracle/wavefront_oracle.c, bit for bit.   python tools/gpu_fuzz_wf_terrain.py [first] [count]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import scenes  # noqa: E402
from forge3d_amd import wavefront  # noqa: E402
from oracle import wavefront_oracle  # noqa: E402  (checker only: this is a test tool)


scene_of = scenes.wavefront_terrain_random_scene


first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
bad, errors, vertices, t0 = [], 0, 0, time.time()
for seed in range(first, first + count):
    try:
        scene, w, h, frames = scene_of(seed)
        d = scene.as_dict()
        want = wavefront_oracle.render(d, w, h, frames)
    except Exception as exc:  # noqa: BLE001  (a scene the oracle refuses is no scene)
        errors += 1
        continue
    got = wavefront.render_scene(d, w, h, frames)
    vertices += int(got.get("path_vertices", 0))
    ok = all(np.array_equal(got[key], want[key], equal_nan=True) for key in ("accum", "hdr", "rgba"))
    if ok and frames >= 2:  # ... and continued from the first frame's accumulation, one frame a launch
        part = wavefront.render_scene(d, w, h, 1, frames_per_launch=1)
        rest = wavefront.render_scene(d, w, h, frames - 1, first_frame=1, accum=part["accum"])
        ok = np.array_equal(rest["accum"], want["accum"], equal_nan=True)
    if not ok:
        bad.append(seed)
print(f"{count} heightfield scenes of the PBR tracer from seed {first}: {len(bad)} mismatches {bad[:10]}, {errors} refused, {vertices / 1e6:.1f} M path vertices, {time.time() - t0:.1f} s")
