#!/usr/bin/env python
"""Wide parity fuzz on the GPU: N seeded random scenes (tests/scenes.random_scene) through libf3dhip vs
the oracle, bit for bit (or the same render error).  python tools/gpu_fuzz.py [first_seed] [count]
(run with OMP_NUM_THREADS=8: the oracle's OpenMP team of a 128-thread host is slower than 8 threads on
images this small)."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import forge3d_amd as f3d  # noqa: E402
import scenes  # noqa: E402
from oracle import oracle  # noqa: E402  (checker only: this is a test tool)

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 2000), (int(sys.argv[2]) if len(sys.argv) > 2 else 400)
POISON = [0x00, 0xFF, 0xA5, 0x7F] if os.environ.get("F3D_FUZZ_POISON") else None  # device buffers pre-filled: results must not care
if POISON:
    import ctypes

    from forge3d_amd import _native

    _native.lib().f3d_scene_cache_limit(ctypes.c_uint32(0))
bad, errors, t0 = [], 0, time.time()
log = open(ROOT / "gpurun_out" / "fuzz_progress.log", "a") if (ROOT / "gpurun_out").exists() else None
for seed in range(first, first + count):
    dem, size, cam, kw = scenes.random_scene(seed)
    if POISON:
        _native.debug_poison(POISON[seed % 4])
    if log:
        print(seed, round(time.time() - t0, 2), file=log, flush=True)
    try:
        want = oracle.render(dem, size[0], size[1], cam, **kw)
    except RuntimeError as exc:
        errors += 1
        try:
            f3d.hybrid_render_terrain_reference(dem, size[0], size[1], cam, **kw)
            bad.append((seed, "oracle raised, GPU did not"))
        except RuntimeError:
            pass
        continue
    try:
        got = f3d.hybrid_render_terrain_reference(dem, size[0], size[1], cam, **kw)
    except RuntimeError as exc:
        bad.append((seed, f"GPU raised: {str(exc)[:80]}"))
        continue
    if not all(np.array_equal(got[k], want[k], equal_nan=True) for k in ("rgba", "albedo", "normal", "depth")) \
            or got["frames"] != want["frames"] or np.float32(got["variance"]) != np.float32(want["variance"]):
        bad.append((seed, "image differs"))
print(f"{count} scenes from seed {first}: {len(bad)} mismatches {bad[:10]}, {errors} render errors on both sides, "
      f"{time.time() - t0:.1f} s")
