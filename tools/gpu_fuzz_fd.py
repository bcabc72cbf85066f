#!/usr/bin/env python
"""Frames in flight against the fused frame kernel on random configurations (scene, size, strip, spp, sample lanes,
batch size, frame count): every output and the variance statistic must be the same bits.  No oracle involved: the
fused kernel is checked against it elsewhere.  python tools/gpu_fuzz_fd.py [first_seed] [count]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import scenes  # noqa: E402
from forge3d_amd import _native  # noqa: E402
from forge3d_amd.session import TerrainSession  # noqa: E402

POISON = [0x00, 0xFF, 0xA5, 0x7F] if os.environ.get("F3D_FUZZ_POISON") else None  # the two sessions of a configuration under different patterns

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
if POISON:
    import ctypes

    _native.lib().f3d_scene_cache_limit(ctypes.c_uint32(0))  # every session builds its own tables, under its own pattern
bad, retraced, t0 = [], 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(900000 + seed)
    dem, size, cam, kw = scenes.random_scene(seed)
    frames = int(rng.integers(2, 40))
    kw = dict(kw, max_frames=frames, min_frames=frames, variance_threshold=1e30)
    h = size[1]
    rows = {}
    if h >= 12 and rng.random() < 0.5:
        b = int(rng.integers(0, h - 6))
        rows = dict(row_begin=b, row_end=int(rng.integers(b + 3, h + 1)))
    variant = int(rng.choice([0, 1000000, 2000000, 4000000, 8000000]))
    fd = int(rng.choice([2, 3, 5, 8, 16, 32]))
    outs = []
    try:
        for k, in_flight in enumerate((0, fd)):
            if os.environ.get("F3D_FUZZ_WAVEFRONT"):  # the batched session traces through the wavefront kernels (terrain-only scenes)
                os.environ["F3D_WAVEFRONT"] = "1" if in_flight else "0"
                os.environ["F3D_WF_QUORUM"] = str(int(rng.choice([1, 8, 16, 32, 64])))
            if POISON:
                _native.debug_poison(POISON[(seed % 4 + k * (1 + (seed // 4) % 3)) % 4])  # two different patterns
            with TerrainSession(dem, size[0], size[1], cam, kernel_variant=variant, frames_in_flight=in_flight,
                                memory_budget_bytes=8 << 30, **rows, **kw) as s:
                s.enqueue_frames(0, frames, True)
                m2, flag = s.window_stats()
                out = s.resolve(frames)
                if in_flight:
                    retraced += s.retraced_pixels()
                outs.append((out, m2, flag))
    except (RuntimeError, ValueError) as exc:
        if len(outs) == 1:
            bad.append((seed, f"only the batched session raised: {str(exc)[:80]}"))
        continue
    (a, m2a, fa), (b, m2b, fb) = outs
    if np.float32(m2a).tobytes() != np.float32(m2b).tobytes() or fa != fb:
        bad.append((seed, "variance statistic"))
    for key in ("rgba", "albedo", "normal", "depth"):
        if not np.array_equal(a[key], b[key], equal_nan=True):
            bad.append((seed, key, variant, fd, frames, rows))
            break
print(f"{count} configurations from seed {first}: {len(bad)} mismatches {bad[:8]}, {retraced} re-traced pixel-frames, {time.time() - t0:.1f} s")
