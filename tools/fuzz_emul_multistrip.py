#!/usr/bin/env python
"""tools/gpu_fuzz_strips.py on the host emulator: 2-4 row strips with halo exchange (EmulBackend sessions, whole frames or
edge / interior parts) against the one-strip image, random scenes and boundaries.  Meant for tools/asan_emul.sh-style runs
(AddressSanitizer build of the emulator preloaded): the strip-session code paths of the shared kernel code under ASan.
python tools/fuzz_emul_multistrip.py [first_seed] [count]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import scenes  # noqa: E402
from emul import emul  # noqa: E402
from forge3d_amd.session import HALO_ROWS as R, reservoir_buffer_bytes  # noqa: E402

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
backend = emul.EmulBackend()
bad, done, t0 = [], 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(700000 + seed)
    dem, size, cam, kw = scenes.random_scene(seed)
    W, H = size
    world = int(rng.integers(2, 5))
    if H < world * R:
        continue
    frames = int(rng.integers(2, 10))
    kw = dict(kw, max_frames=frames, min_frames=frames, variance_threshold=1e30)
    extra = rng.multinomial(H - world * R, np.ones(world) / world)
    bounds, b = [], 0
    for i in range(world):
        bounds.append((b, b + R + int(extra[i])))
        b = bounds[-1][1]
    parts = bool(rng.integers(0, 2))
    try:
        full = emul.render(dem, W, H, cam, **kw)["rgba"].reshape(H, W, 4)
    except RuntimeError:
        continue
    sessions, bufs = [], []
    for b, e in bounds:
        res = [torch.zeros(reservoir_buffer_bytes(e - b, W), dtype=torch.uint8) for _ in range(2)]
        bufs.append(res)
        sessions.append(backend.make_session(dem, W, H, cam, b, e, res, backend.empty_i32(4), kw))
    row = W * 16
    for f in range(frames):
        last = f + 1 == frames
        for s in sessions:
            if parts:
                s.enqueue_frame_part(f, 1, last)
                s.enqueue_frame_part(f, 2, last)
            else:
                s.enqueue_frames(f, 1, last)
        for i in range(len(bounds) - 1):
            up, dn = bufs[i][f & 1], bufs[i + 1][f & 1]
            rows_up = bounds[i][1] - bounds[i][0]
            dn[0:R * row] = up[rows_up * row:(rows_up + R) * row]
            up[(rows_up + R) * row:(rows_up + 2 * R) * row] = dn[R * row:2 * R * row]
    got = np.concatenate([s.resolve(frames)["rgba"] for s in sessions], axis=0)
    for s in sessions:
        s.close()
    done += 1
    if not np.array_equal(got, full):
        bad.append((seed, bounds, parts, frames))
print(f"{done} of {count} configurations from seed {first} rendered as strips: {len(bad)} mismatches {bad[:6]}, {time.time() - t0:.1f} s")
