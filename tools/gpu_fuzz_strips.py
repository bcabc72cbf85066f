#!/usr/bin/env python
"""Row strips with halo exchange against the one-strip image on random configurations (scene, size, 2-4 strips with
random boundaries of >= HALO_ROWS rows, spp, sample lanes, fused frames / frames in flight / two-part frames, frame
count): every output must be the same bits.  The exchange is a device-to-device copy on one GPU (what RCCL moves
between ranks).  python tools/gpu_fuzz_strips.py [first_seed] [count]"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import scenes  # noqa: E402
from forge3d_amd import _native  # noqa: E402
from forge3d_amd.session import HALO_ROWS as R, TerrainSession, reservoir_buffer_bytes  # noqa: E402

POISON = bool(os.environ.get("F3D_FUZZ_POISON"))  # the one-strip render and the strips under different fill patterns

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
if POISON:
    import ctypes

    _native.lib().f3d_scene_cache_limit(ctypes.c_uint32(0))  # every session builds its own tables, under its own pattern
dev = torch.device("cuda", 0)
bad, done, t0 = [], 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(700000 + seed)
    dem, size, cam, kw = scenes.random_scene(seed)
    W, H = size
    world = int(rng.integers(2, 5))
    if H < world * R:
        continue
    frames = int(rng.integers(2, 24))
    kw = dict(kw, max_frames=frames, min_frames=frames, variance_threshold=1e30)
    extra = rng.multinomial(H - world * R, np.ones(world) / world)  # every strip: R rows plus its share of the rest
    bounds, b = [], 0
    for i in range(world):
        bounds.append((b, b + R + int(extra[i])))
        b = bounds[-1][1]
    assert b == H
    variant = int(rng.choice([0, 1000000, 2000000, 4000000, 8000000]))
    mode = str(rng.choice(["fused", "in_flight", "parts"]))
    fd = int(rng.choice([2, 3, 8, 16])) if mode == "in_flight" else 0
    if POISON:
        _native.debug_poison((0x00, 0xA5)[seed % 2])
    try:
        with TerrainSession(dem, W, H, cam, kernel_variant=variant, memory_budget_bytes=8 << 30, **kw) as s:
            s.enqueue_frames(0, frames, True)
            m2_full, flag_full = s.window_stats()
            full = s.resolve(frames)
    except (RuntimeError, ValueError):
        continue
    if POISON:
        _native.debug_poison((0xFF, 0x7F)[(seed // 2) % 2])
    sessions, bufs = [], []
    for b, e in bounds:
        res = [torch.zeros(reservoir_buffer_bytes(e - b, W), dtype=torch.uint8, device=dev) for _ in range(2)]
        bufs.append(res)
        sessions.append(TerrainSession(dem, W, H, cam, row_begin=b, row_end=e, frames_in_flight=fd, kernel_variant=variant,
                                       memory_budget_bytes=8 << 30, ext_reservoirs=(res[0].data_ptr(), res[1].data_ptr()), **kw))
    row = W * 16

    def exchange(which):
        torch.cuda.synchronize()
        if os.environ.get("F3D_FUZZ_NO_EXCHANGE"):  # negative control: the harness must notice
            return
        for i in range(len(bounds) - 1):
            up, dn = bufs[i][which], bufs[i + 1][which]
            rows_up = bounds[i][1] - bounds[i][0]
            dn[0:R * row] = up[rows_up * row:(rows_up + R) * row]
            up[(rows_up + R) * row:(rows_up + 2 * R) * row] = dn[R * row:2 * R * row]
        torch.cuda.synchronize()

    f = 0
    while f < frames:
        last = f + 1 == frames
        if mode == "in_flight":
            n = sessions[0].trace_batch(f, frames - f)
            for s in sessions:
                s.enqueue_trace(f, n)
            for g in range(f, f + n):
                for s in sessions:
                    s.enqueue_merge(g, g + 1 == frames)
                exchange(g & 1)
            f += n
            continue
        if mode == "parts":
            for s in sessions:
                s.enqueue_frame_part(f, 1, last)
            for s in sessions:
                s.enqueue_frame_part(f, 2, last)
        else:
            for s in sessions:
                s.enqueue_frames(f, 1, last)
        exchange(f & 1)
        f += 1
    stats = [s.window_stats() for s in sessions]
    parts = [s.resolve(frames) for s in sessions]
    for s in sessions:
        s.close()
    done += 1
    m2 = max(st[0] for st in stats)
    if np.float32(m2).tobytes() != np.float32(m2_full).tobytes() or any(st[1] for st in stats) != flag_full:
        bad.append((seed, "variance statistic", mode, bounds))
    for key in ("rgba", "albedo", "normal", "depth"):
        if not np.array_equal(np.concatenate([p[key] for p in parts], axis=0), full[key], equal_nan=True):
            bad.append((seed, key, mode, fd, variant, frames, bounds))
            break
print(f"{done} of {count} configurations from seed {first} rendered as strips: {len(bad)} mismatches {bad[:6]}, {time.time() - t0:.1f} s")
