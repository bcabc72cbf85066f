#!/usr/bin/env python
"""Certificates (csrc/f3d_cone.h) on the host emulator: every random scene rendered with the camera rays starting where
the pixel's cone certificate ends and the sun rays stopping where the pixel's cylinder certificate begins, and again
with neither (F3D_EMUL_NO_PRIMARY_START, F3D_EMUL_NO_SUN_CLEAR):
every output must be the same bits.  python tools/fuzz_emul_certificates.py [first_seed] [count]"""
import os
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
bad, done, t0 = [], 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(500000 + seed)
    dem, size, cam, kw = scenes.random_scene(seed)
    frames = int(rng.integers(2, 6))
    kw = dict(kw, max_frames=frames, min_frames=frames, variance_threshold=1e30)
    lanes = int(rng.choice([1, 1, 4]))
    outs = []
    try:
        for off in ("1", None):
            for name in ("F3D_EMUL_NO_PRIMARY_START", "F3D_EMUL_NO_SUN_CLEAR", "F3D_EMUL_NO_IBL_STOP"):
                if off:
                    os.environ[name] = off
                else:
                    os.environ.pop(name, None)
            outs.append(emul.render(dem, size[0], size[1], cam, sample_lanes=lanes, **kw))
    except RuntimeError:
        continue
    finally:
        os.environ.pop("F3D_EMUL_NO_PRIMARY_START", None)
        os.environ.pop("F3D_EMUL_NO_SUN_CLEAR", None)
        os.environ.pop("F3D_EMUL_NO_IBL_STOP", None)
    done += 1
    if not all(np.array_equal(outs[0][k], outs[1][k], equal_nan=True) for k in ("rgba", "albedo", "normal", "depth", "accum", "m2", "res")):
        bad.append(seed)
print(f"{done} of {count} scenes from seed {first}: {len(bad)} mismatches {bad[:10]}, {time.time() - t0:.1f} s")
