#!/usr/bin/env python
"""tools/gpu_fuzz_fd.py's random configurations (scene, size, lone strip, spp, sample lanes, batch size, frame count)
on the host emulator; meant to run under AddressSanitizer (see tools/asan_emul.sh for the build + preload), where any
out-of-bounds access of the shared kernel code stops the run.  Also checks frames in flight == fused frames.
python tools/fuzz_emul_strips.py [first_seed] [count]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import scenes  # noqa: E402
from emul import emul  # noqa: E402

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
bad, t0, done = [], time.time(), 0
for seed in range(first, first + count):
    rng = np.random.default_rng(900000 + seed)
    dem, size, cam, kw = scenes.random_scene(seed)
    frames = int(rng.integers(2, 40))
    kw = dict(kw, max_frames=frames, min_frames=frames, variance_threshold=1e30)
    h = size[1]
    rows = None
    if h >= 12 and rng.random() < 0.5:
        b = int(rng.integers(0, h - 6))
        rows = (b, int(rng.integers(b + 3, h + 1)))
    lanes = int(rng.choice([1, 1, 2, 4, 8]))
    fd = int(rng.choice([2, 3, 5, 8, 16, 32]))
    try:
        a = emul.render(dem, size[0], size[1], cam, rows=rows, sample_lanes=lanes, **kw)
        b = emul.render(dem, size[0], size[1], cam, rows=rows, frames_in_flight=fd, **kw)
    except RuntimeError:
        continue
    done += 1
    if not all(np.array_equal(a[k], b[k], equal_nan=True) for k in ("rgba", "albedo", "normal", "depth", "accum", "m2")):
        bad.append((seed, lanes, fd, frames, rows))
print(f"{done} of {count} configurations from seed {first} rendered: {len(bad)} mismatches {bad[:8]}, {time.time() - t0:.1f} s")
