#!/usr/bin/env python
"""Schedule model of the sample-lane frame kernel (4 x 4 pixels x 4 sample lanes a wave, two rounds at 8 spp) from the CPU
emulator's per-ray step logs: where the idle lanes of k_frame are, phase by phase, and what between-phase compaction of a
tile's 8 samples per pixel could buy (round-4 verdict item 3).  Cost unit = one march iteration of the wave.

    python tools/phase_model.py [rows=400:656] [log=/tmp/model/band.raylog]

  shipped      per round: primaries in lockstep, sun rays in lockstep, IBL rays in lockstep with the tail (<= 16 marching
               lanes) dealt over the wave (ray sharing, modelled as perfect packing x SHARE_EFF + SHARE_ROUND per deal)
  phase-major  primaries of both rounds, then the sun rays of BOTH rounds ballot-compacted into as few 64-lane passes as
               they fill, then the IBL rays likewise
  stream       phase-major, and inside an occlusion phase a lane that finishes takes the next waiting ray (refill when
               REFILL_Q lanes wait, REFILL_COST iterations a refill)
"""
import os
import struct
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

SHARE_BELOW, SHARE_EFF, SHARE_ROUND = 16, 0.7, 3.0
REFILL_Q, REFILL_COST = 16, 3.0


def load_log(path):
    data = open(path, "rb").read()
    W, rows, spp, _ = struct.unpack_from("<4I", data, 0)
    off = 16
    rec = np.dtype([("kind", "<u4"), ("steps", "<u4"), ("mask", "<u8")])
    pixels = []
    for _ in range(W * rows):
        (n,) = struct.unpack_from("<I", data, off)
        off += 4
        pixels.append(np.frombuffer(data, rec, n, off))
        off += 16 * n
    return W, rows, spp, pixels


def make_log(rows, path):
    from emul import emul
    from forge3d_amd import datasets

    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    os.environ["F3D_EMUL_RAYLOG"] = path
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    del os.environ["F3D_EMUL_RAYLOG"]


def lockstep(steps):
    """(wave iterations, lane-steps) of rays marched in lockstep without sharing."""
    steps = np.asarray(steps, float)
    return (steps.max() if steps.size else 0.0), steps.sum()


def lockstep_shared(steps):
    """lockstep until <= SHARE_BELOW lanes still march, then the rest dealt over the wave."""
    s = np.sort(np.asarray(steps, float))[::-1]
    if s.size == 0:
        return 0.0
    if s.size <= SHARE_BELOW:
        t = 0.0
        rest = s.sum()
    else:
        t = s[SHARE_BELOW]  # the moment the (SHARE_BELOW+1)-th longest ray finishes
        rest = np.clip(s[:SHARE_BELOW] - t, 0, None).sum()
    if rest <= 0:
        return t
    rounds = max(1.0, np.ceil(np.log2(max(rest / 64.0, 1.0)) + 1))
    return t + rest / (64.0 * SHARE_EFF) + SHARE_ROUND * min(rounds, 10)


def stream(steps, shared_tail=True):
    """64 lanes, a pool of rays: finished lanes wait; when REFILL_Q wait (or nobody marches) they take new rays."""
    pool = list(np.asarray(steps, float))
    pool.reverse()
    lanes = np.zeros(64)
    t = 0.0
    # first fill
    n = min(64, len(pool))
    for i in range(n):
        lanes[i] = pool.pop()
    while True:
        marching = lanes > 0
        if not pool:
            # drain: the tail
            live = lanes[marching]
            if live.size == 0:
                break
            t += lockstep_shared(live) if shared_tail else live.max()
            break
        idle = (~marching).sum()
        if idle >= REFILL_Q or not marching.any():
            t += REFILL_COST
            for i in np.nonzero(~marching)[0]:
                if not pool:
                    break
                lanes[i] = pool.pop()
            continue
        live = np.sort(lanes[marching])
        # advance until REFILL_Q lanes are idle
        need = REFILL_Q - idle
        step = live[need - 1]
        t += step
        lanes[marching] -= step
        lanes = np.clip(lanes, 0, None)
    return t


def main():
    rows = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "400:656").split(":"))
    path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/model/band_%d_%d.raylog" % rows
    if not os.path.exists(path):
        make_log(rows, path)
    W, R, spp, pixels = load_log(path)
    S, TW, TH = 4, 4, 4
    tot = {k: 0.0 for k in ("shipped", "phase-major", "pm+stream-ibl", "pm+stream-both", "stream-in-round")}
    phase = {k: [0.0, 0.0] for k in ("primary", "sun", "ibl")}  # [lane-steps, 64 x wave iterations] in the shipped schedule
    pm_phase = {k: [0.0, 0.0] for k in ("primary", "sun", "ibl")}
    counts = dict(samples=0, hits=0, sun=0, ibl=0, waves=0, waves_mixed=0)
    for ty in range(0, R, TH):
        for tx in range(0, W, TW):
            px = [pixels[y * W + x] for y in range(ty, min(ty + TH, R)) for x in range(tx, min(tx + TW, W))]
            P = np.zeros((len(px), spp, 3))
            for li, rays in enumerate(px):
                s = -1
                for kind, steps, mask in rays:
                    kind = int(kind) & 0xFF
                    if kind == 2:
                        s += 1
                        P[li, s, 0] = max(steps, 1)
                    elif kind == 7:
                        P[li, s, 1] = max(steps, 1)
                    else:
                        P[li, s, 2] = max(steps, 1)
            counts["waves"] += 1
            counts["samples"] += P.shape[0] * spp
            hit = (P[:, :, 1] > 0) | (P[:, :, 2] > 0)
            counts["hits"] += hit.sum()
            counts["sun"] += (P[:, :, 1] > 0).sum()
            counts["ibl"] += (P[:, :, 2] > 0).sum()
            if 0 < hit.sum() < hit.size:
                counts["waves_mixed"] += 1
            ship = 0.0
            for r0 in range(0, spp, S):
                blk = P[:, r0:r0 + S, :].reshape(-1, 3)
                for j, name in enumerate(("primary", "sun", "ibl")):
                    rays = blk[:, j][blk[:, j] > 0]
                    c = lockstep_shared(rays) if name == "ibl" else (rays.max() if rays.size else 0.0)
                    ship += c
                    phase[name][0] += rays.sum()
                    phase[name][1] += 64.0 * c
            tot["shipped"] += ship
            # phase-major
            prim = sum(P[:, r0:r0 + S, 0].max() for r0 in range(0, spp, S))
            pm_phase["primary"][0] += P[:, :, 0].sum()
            pm_phase["primary"][1] += 64.0 * prim
            pm = prim
            ps_ibl = prim
            ps_both = prim
            for j, name in ((1, "sun"), (2, "ibl")):
                rays = P[:, :, j].T.reshape(-1)  # sample-major: round A's rays first
                rays = rays[rays > 0]
                c = 0.0
                for p0 in range(0, rays.size, 64):
                    part = rays[p0:p0 + 64]
                    c += lockstep_shared(part) if name == "ibl" else part.max()
                pm += c
                pm_phase[name][0] += rays.sum()
                pm_phase[name][1] += 64.0 * c
                st = stream(rays, shared_tail=(name == "ibl")) if rays.size else 0.0
                ps_ibl += st if name == "ibl" else c
                ps_both += st
            tot["phase-major"] += pm
            tot["pm+stream-ibl"] += ps_ibl
            tot["pm+stream-both"] += ps_both
    print(f"rows {rows}: {counts['waves']} waves ({counts['waves_mixed']} with hits AND misses), {counts['samples']} samples, "
          f"hit {counts['hits'] / counts['samples']:.3f}, sun ray {counts['sun'] / max(1, counts['hits']):.3f} of hits, "
          f"IBL ray {counts['ibl'] / max(1, counts['hits']):.3f} of hits")
    for name in ("primary", "sun", "ibl"):
        u, c = phase[name]
        u2, c2 = pm_phase[name]
        print(f"  {name:8s}: shipped lane-steps {u:.4g}, wave iterations {c / 64:.4g}, lockstep lane use {u / max(c, 1):.3f}"
              f"   | phase-major iterations {c2 / 64:.4g} ({u2 / max(c2, 1):.3f})")
    for k, v in tot.items():
        if v:
            print(f"  {k:16s} wave iterations {v:.5g}  ({tot['shipped'] / v:.3f}x)")


if __name__ == "__main__":
    main()
