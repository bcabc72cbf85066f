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
  pair         per round ONE march loop for both occlusion rays of a sample: a lane runs its sun ray, then its IBL ray
               ("pair ideal": switching at once and for free; "pair q<N>": the lanes that are through with their sun ray
               wait until N of them wait -- or nobody else will -- and switch together at REFILL_COST, the tail shared)

    python tools/phase_model.py 400:656 [log] [--pair]     (--pair: every 5th tile, the pair schedules are slow to simulate)
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


def pair_stream(sun, ibl, quorum, refill_cost, share=True):
    """One loop for both occlusion rays of every lane (see the module docstring)."""
    q = [[x for x in (s, i) if x > 0] for s, i in zip(sun, ibl)]
    lanes = np.zeros(len(q))
    pending = np.array([len(x) for x in q])
    for li in range(len(q)):
        if q[li]:
            lanes[li] = q[li].pop(0)
            pending[li] -= 1
    t = refill_cost
    for _ in range(10000):
        marching = lanes > 0
        waiting = (~marching) & (pending > 0)
        if not marching.any() and not waiting.any():
            break
        if pending.sum() == 0:
            live = lanes[marching]
            t += lockstep_shared(live) if share else live.max()
            break
        if waiting.sum() >= quorum or not marching.any() or (waiting.any() and not (marching & (pending > 0)).any()):
            t += refill_cost
            for li in np.nonzero(waiting)[0]:
                lanes[li] = q[li].pop(0)
                pending[li] -= 1
            continue
        cand = np.sort(lanes[marching & (pending > 0)])
        need = quorum - waiting.sum()
        step = cand[need - 1] if len(cand) >= need else (cand[-1] if len(cand) else lanes[marching].max())
        t += step
        lanes[marching] -= step
        lanes = np.clip(lanes, 0, None)
    return t


def pair_table(W, R, spp, pixels):
    names = ("shipped", "pair ideal", "pair q16", "pair q32", "pair q16, switch cost 1", "pair q16, tail not shared")
    tot = dict.fromkeys(names, 0.0)
    prim = 0.0
    tiles = [(ty, tx) for ty in range(0, R, 4) for tx in range(0, W, 4)][::5]
    for ty, tx in tiles:
        px = [pixels[y * W + x] for y in range(ty, min(ty + 4, R)) for x in range(tx, min(tx + 4, W))]
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
        for r0 in range(0, spp, 4):
            blk = P[:, r0:r0 + 4, :].reshape(-1, 3)
            prim += blk[:, 0].max()
            sun, ibl = blk[:, 1], blk[:, 2]
            if not ((sun > 0) | (ibl > 0)).any():
                continue
            s_, i_ = sun[sun > 0], ibl[ibl > 0]
            tot["shipped"] += (s_.max() if s_.size else 0.0) + (lockstep_shared(i_) if i_.size else 0.0)
            both = sun + ibl
            tot["pair ideal"] += lockstep_shared(both[both > 0])
            tot["pair q16"] += pair_stream(sun, ibl, 16, REFILL_COST)
            tot["pair q32"] += pair_stream(sun, ibl, 32, REFILL_COST)
            tot["pair q16, switch cost 1"] += pair_stream(sun, ibl, 16, 1.0)
            tot["pair q16, tail not shared"] += pair_stream(sun, ibl, 16, REFILL_COST, False)
    print(f"pair schedules on {len(tiles)} tiles: primary iterations {prim:.0f}")
    for k, v in tot.items():
        print(f"  {k:28s} occlusion iterations {v:9.0f}   whole frame {(prim + tot['shipped']) / (prim + v):.3f}x")


def main():
    rows = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "400:656").split(":"))
    path = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "/tmp/model/band_%d_%d.raylog" % rows
    if not os.path.exists(path):
        make_log(rows, path)
    W, R, spp, pixels = load_log(path)
    if "--pair" in sys.argv:
        pair_table(W, R, spp, pixels)
        return
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
