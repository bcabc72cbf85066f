#!/usr/bin/env python
"""Parity fuzz of the multi-bounce PBR tracer on the GPU: N seeded random scenes (tests/scenes.wavefront_random_scene)
through f3d_wavefront_render vs oracle/wavefront_oracle.c, bit for bit.  python tools/gpu_fuzz_wavefront.py [first] [count]"""
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

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
bad, t0 = [], time.time()
for seed in range(first, first + count):
    scene, w, h, frames = scenes.wavefront_random_scene(seed)
    d = scene.as_dict()
    want = wavefront_oracle.render(d, w, h, frames)
    got = wavefront.render_scene(d, w, h, frames)
    for key in ("accum", "hdr", "rgba"):
        if not np.array_equal(got[key], want[key], equal_nan=True):
            bad.append((seed, key, int((got[key] != want[key]).sum())))
            break
print(f"{count} PBR scenes from seed {first}: {len(bad)} mismatches {bad[:10]}, {time.time() - t0:.1f} s")
