#!/usr/bin/env python
"""SIMT schedule model of the march kernel from the CPU emulator's per-ray step logs.

Renders a strip of the headline frame (1920 wide, full DEM) through tests/emul with
F3D_EMUL_RAYLOG and replays the per-ray march-step counts through schedules of a 64-lane
wave (8x8 pixel tile).  Cost unit = one march iteration of the wave; a schedule's lane
utilisation = total lane-steps / (64 * wave iterations).

  nested        shipped kernel: per sample, primary / shadow / IBL phases in lockstep
  sec-fused     per sample: primary, then each lane runs shadow+IBL back to back
  sec-batched   all primaries of the frame first (lockstep), then each lane runs all its
                secondary rays back to back
  continuous    each lane runs all of its rays back to back
  ideal         perfect packing
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


def main():
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "480:608").split(":"))
    w, h, spp = 1920, 1080, 8
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, w, h, cam, rows=rows, **dict(kw, spp=spp, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)

    names = ["nested", "sec-fused", "sec-batched", "continuous"]
    cost = dict.fromkeys(names, 0.0)
    lane_steps = 0.0
    by_kind = {2: [0.0, 0.0], 7: [0.0, 0.0], 3: [0.0, 0.0]}
    leafs = 0
    hist = []
    for ty in range(0, R, 8):
        for tx in range(0, W, 8):
            lanes = [pixels[y * W + x] for y in range(ty, min(ty + 8, R)) for x in range(tx, min(tx + 8, W))]
            # per lane: samples[s] = (primary, shadow, ibl) step counts
            P = np.zeros((len(lanes), spp, 3))
            for li, rays in enumerate(lanes):
                s = -1
                for kind, steps, mask in rays:
                    kind = int(kind) & 0xFF
                    if kind == 2:
                        s += 1
                        P[li, s, 0] = steps
                    elif kind == 7:
                        P[li, s, 1] = steps
                    else:
                        P[li, s, 2] = steps
                    leafs += bin(int(mask)).count("1")
            lane_steps += P.sum()
            mx = P.max(axis=0)  # (spp, 3)
            cost["nested"] += mx.sum()
            cost["sec-fused"] += mx[:, 0].sum() + (P[:, :, 1] + P[:, :, 2]).max(axis=0).sum()
            cost["sec-batched"] += mx[:, 0].sum() + (P[:, :, 1] + P[:, :, 2]).sum(axis=1).max()
            cost["continuous"] += P.sum(axis=(1, 2)).max()
            for j, k in enumerate((2, 7, 3)):
                by_kind[k][0] += P[:, :, j].sum()
                by_kind[k][1] += 64 * mx[:, j].sum()
            hist.append(P.reshape(-1))
    n_rays = sum(len(p) for p in pixels)
    print(f"rows {rows}: {n_rays} rays, {lane_steps / n_rays:.1f} steps/ray, {leafs / n_rays:.2f} queued leaves/ray")
    for k, name in ((2, "primary"), (7, "shadow"), (3, "ibl")):
        u, c = by_kind[k]
        print(f"  {name:8s}: lane-steps {u:.3g} ({u / lane_steps:.1%} of all), lockstep utilisation {u / c:.3f}")
    for name in names:
        print(f"  {name:12s}: utilisation {lane_steps / (64 * cost[name]):.3f}  wave iterations {cost[name]:.0f}")
    print(f"  ideal       : wave iterations {lane_steps / 64:.0f}")
    allp = np.concatenate(hist)
    allp = allp[allp > 0]
    print("  steps/ray percentiles 10/50/90/99/max:", [float(np.percentile(allp, q)) for q in (10, 50, 90, 99, 100)])




def simulate_batched(P, batch, quorum, switch_cost):
    """Phase-B simulator: P (lanes, spp, 3) step counts; per batch of `batch` samples the lanes run
    their secondary rays back to back; finished lanes wait until `quorum` lanes wait (or nobody
    marches), then all of them switch (cost `switch_cost` iterations)."""
    lanes, spp, _ = P.shape
    total = 0.0
    for b0 in range(0, spp, batch):
        total += P[:, b0:b0 + batch, 0].max(axis=0).sum()  # primaries in lockstep
        queues = []
        for li in range(lanes):
            q = [x for s in range(b0, min(b0 + batch, spp)) for x in (P[li, s, 1], P[li, s, 2]) if x > 0]
            queues.append(q[::-1])
        remaining = np.zeros(lanes)
        pending = np.array([len(q) for q in queues])
        while True:
            waiting = (remaining <= 0) & (pending > 0)
            marching = remaining > 0
            if not marching.any() and not waiting.any():
                break
            if waiting.sum() >= quorum or not marching.any():
                total += switch_cost
                for li in np.nonzero(waiting)[0]:
                    remaining[li] = queues[li].pop()
                    pending[li] -= 1
                marching = remaining > 0
            # advance to the next event in one go
            if waiting.sum() >= quorum:
                continue
            live = remaining[marching]
            # steps until the number of waiting lanes can change
            step = live.min()
            total += step
            remaining[marching] -= step
    return total


def batched_table(rows="480:608"):
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    w, h, spp = 1920, 1080, 8
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, w, h, cam, rows=rows, **dict(kw, spp=spp, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    waves = []
    for ty in range(0, R, 8):
        for tx in range(0, W, 8 * 4):  # every 4th tile: the simulator is slow
            lanes = [pixels[y * W + x] for y in range(ty, min(ty + 8, R)) for x in range(tx, min(tx + 8, W))]
            P = np.zeros((len(lanes), spp, 3))
            for li, rays in enumerate(lanes):
                s = -1
                for kind, steps, mask in rays:
                    kind = int(kind) & 0xFF
                    if kind == 2:
                        s += 1
                    P[li, s, {2: 0, 7: 1}.get(int(kind), 2)] = steps
            waves.append(P)
    lane_steps = sum(P.sum() for P in waves)
    nested = sum(P.max(axis=0).sum() for P in waves)
    print(f"{len(waves)} waves; nested iterations {nested:.0f} (utilisation {lane_steps / 64 / nested:.3f})")
    for batch in (1, 2, 4, 8):
        for quorum in (1, 8, 16, 32):
            for sc in (1.5, 3.0):
                c = sum(simulate_batched(P, batch, quorum, sc) for P in waves)
                print(f"  batch {batch} quorum {quorum:2d} switch {sc}: iterations {c:.0f}  = {nested / c:.2f}x fewer")


def ibl_batched_bound(rows="480:608"):
    """Upper bound (free switches) for deferring only the IBL rays of B consecutive samples."""
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    nested = 0.0
    alt = {2: 0.0, 4: 0.0, 8: 0.0}
    alt_si = {2: 0.0, 4: 0.0, 8: 0.0}
    for ty in range(0, R, 8):
        for tx in range(0, W, 8):
            lanes = [pixels[y * W + x] for y in range(ty, min(ty + 8, R)) for x in range(tx, min(tx + 8, W))]
            P = np.zeros((len(lanes), spp, 3))
            for li, rays in enumerate(lanes):
                s = -1
                for kind, steps, mask in rays:
                    kind = int(kind) & 0xFF
                    if kind == 2:
                        s += 1
                    P[li, s, {2: 0, 7: 1}.get(int(kind), 2)] = steps
            mx = P.max(axis=0)
            nested += mx.sum()
            for B in alt:
                ib = P[:, :, 2].reshape(len(lanes), spp // B, B).sum(axis=2).max(axis=0).sum()
                alt[B] += mx[:, 0].sum() + mx[:, 1].sum() + ib
                sb = P[:, :, 1].reshape(len(lanes), spp // B, B).sum(axis=2).max(axis=0).sum()
                alt_si[B] += mx[:, 0].sum() + sb + ib
    for B in alt:
        print(f"  IBL rays of {B} samples back to back: {nested / alt[B]:.3f}x fewer iterations;"
              f" + shadow rays likewise (separately): {nested / alt_si[B]:.3f}x")


def lanes_model(rows="480:608", S=8):
    """Lockstep utilisation of the sample-lane kernel: a wave = (64 / S) pixels x S samples, one
    phase per ray type."""
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    tw, th = {1: (8, 8), 2: (8, 4), 4: (4, 4), 8: (4, 2)}[S]
    tot = np.zeros(3)
    cost = np.zeros(3)
    for ty in range(0, R, th):
        for tx in range(0, W, tw):
            lanes = [pixels[y * W + x] for y in range(ty, min(ty + th, R)) for x in range(tx, min(tx + tw, W))]
            P = np.zeros((len(lanes), spp, 3))
            for li, rays in enumerate(lanes):
                s = -1
                for kind, steps, mask in rays:
                    kind = int(kind) & 0xFF
                    if kind == 2:
                        s += 1
                    P[li, s, {2: 0, 7: 1}.get(int(kind), 2)] = steps
            for r0 in range(0, spp, S):
                blk = P[:, r0:r0 + S, :].reshape(-1, 3)
                tot += blk.sum(axis=0)
                cost += 64 * blk.max(axis=0)
    names = ("primary", "shadow", "ibl")
    print(f"S={S}: overall utilisation {tot.sum() / cost.sum():.3f}; wave iterations {cost.sum() / 64:.0f}")
    for k in range(3):
        print(f"  {names[k]:8s}: {tot[k] / cost[k]:.3f} ({cost[k] / cost.sum():.1%} of the iterations)")


def split_model(rows="480:608", S=8):
    """What if the last few marching any-hit rays of a phase were split into segments handed to
    the idle lanes (one rebalance per phase, when <= thr lanes still march)?"""
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    tw, th = {1: (8, 8), 2: (8, 4), 4: (4, 4), 8: (4, 2)}[S]
    base = np.zeros(3)
    alt = {(thr, ov): np.zeros(3) for thr in (4, 8, 16, 32) for ov in (3, 8)}
    for ty in range(0, R, th):
        for tx in range(0, W, tw):
            lanes = [pixels[y * W + x] for y in range(ty, min(ty + th, R)) for x in range(tx, min(tx + tw, W))]
            P = np.zeros((len(lanes), spp, 3))
            for li, rays in enumerate(lanes):
                s = -1
                for kind, steps, mask in rays:
                    kind = int(kind) & 0xFF
                    if kind == 2:
                        s += 1
                    P[li, s, {2: 0, 7: 1}.get(int(kind), 2)] = steps
            for r0 in range(0, spp, S):
                blk = P[:, r0:r0 + S, :].reshape(-1, 3)
                base += blk.max(axis=0)
                for (thr, ov), acc in alt.items():
                    for k in range(3):
                        st = np.sort(blk[:, k])[::-1]  # descending
                        if k == 0 or st[0] == 0:
                            acc[k] += st[0]
                            continue
                        n = len(st)
                        t_reb = st[thr] if thr < n else 0.0  # when only thr rays are left marching
                        live = st[:thr] - t_reb
                        live = live[live > 0]
                        if len(live) == 0:
                            acc[k] += st[0]
                            continue
                        m = max(1, 64 // len(live))
                        after = (live / m + ov).max() if m > 1 else live.max()
                        acc[k] += min(st[0], t_reb + 2 + after)
    print(f"S={S}: baseline iterations primary/shadow/ibl = {base.astype(int)}  total {base.sum():.0f}")
    for (thr, ov), acc in alt.items():
        print(f"  rebalance at <= {thr:2d} marching, segment overhead {ov} steps: shadow {base[1] / acc[1]:.2f}x ibl {base[2] / acc[2]:.2f}x"
              f" total {base.sum() / acc.sum():.3f}x fewer iterations")


def persistent_model(rows="480:608"):
    """Kernel-B model: all IBL (or shadow) rays of the strip in generation order, traced by 64-lane
    waves of PERSISTENT lanes that refill from a queue when >= Q lanes are idle (refill costs R
    iterations for the wave)."""
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    for kind, name in ((3, "ibl"), (7, "shadow")):
        steps = []
        for ty in range(0, R, 2):  # 4x2-pixel tiles, sample-major inside the pixel: the order kernel A would emit
            for tx in range(0, W, 4):
                for y in range(ty, min(ty + 2, R)):
                    for x in range(tx, min(tx + 4, W)):
                        steps.extend(int(s) for k, s, _ in pixels[y * W + x] if k == kind)
        steps = np.asarray(steps[:400000], np.float64)
        lockstep = sum(steps[i:i + 64].max() for i in range(0, len(steps), 64))
        print(f"{name}: {len(steps)} rays, mean {steps.mean():.1f} steps; lockstep waves {lockstep:.0f} iterations"
              f" (utilisation {steps.sum() / 64 / lockstep:.3f})")
        for Q in (8, 16, 32):
            for Rc in (2.0, 4.0):
                # one long-lived wave per 4096 rays (many waves in flight on the chip)
                total = 0.0
                for c0 in range(0, len(steps), 4096):
                    chunk = steps[c0:c0 + 4096]
                    nxt = 64
                    rem = chunk[:64].copy()
                    if len(rem) < 64:
                        total += rem.max() if len(rem) else 0.0
                        continue
                    while True:
                        idle = rem <= 0
                        n_idle = int(idle.sum())
                        if nxt < len(chunk) and (n_idle >= Q or n_idle == 64):
                            take = min(n_idle, len(chunk) - nxt)
                            idx = np.nonzero(idle)[0][:take]
                            rem[idx] = chunk[nxt:nxt + take]
                            nxt += take
                            total += Rc
                            continue
                        live = rem[rem > 0]
                        if len(live) == 0:
                            break
                        if nxt >= len(chunk):
                            total += live.max()
                            break
                        # run until enough lanes are idle to refill
                        order = np.sort(live)
                        need = max(0, Q - n_idle)
                        step = order[min(need, len(order)) - 1] if need > 0 else order[0]
                        total += step
                        rem -= step
                print(f"   persistent, refill at {Q:2d} idle, refill cost {Rc}: {total:.0f} iterations = {lockstep / total:.2f}x fewer"
                      f" (utilisation {steps.sum() / 64 / total:.3f})")


def walkout_model(rows="480:608"):
    """How much of an occlusion ray's march is pure walk-out (steps after the LAST node whose band
    test passed)?  An oracle 'certificate' that ended a ray right there bounds what horizon tables
    could buy; reported per phase as lockstep wave iterations with S = 4 sample lanes."""
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    S, tw, th = 4, 4, 4
    base = np.zeros(3)
    cut = np.zeros(3)
    lane_steps = np.zeros(3)
    lane_cut = np.zeros(3)
    for ty in range(0, R, th):
        for tx in range(0, W, tw):
            lanes = [pixels[y * W + x] for y in range(ty, min(ty + th, R)) for x in range(tx, min(tx + tw, W))]
            P = np.zeros((len(lanes), spp, 3))
            Q = np.zeros((len(lanes), spp, 3))
            for li, rays in enumerate(lanes):
                s = -1
                for kind, steps, mask in rays:
                    k = int(kind) & 0xFF
                    last_pass = int(kind) >> 8
                    if k == 2:
                        s += 1
                    j = {2: 0, 7: 1}.get(k, 2)
                    P[li, s, j] = steps
                    Q[li, s, j] = min(int(steps), last_pass + 2)  # + the step that notices the clearance
            for r0 in range(0, spp, S):
                b, q = P[:, r0:r0 + S, :].reshape(-1, 3), Q[:, r0:r0 + S, :].reshape(-1, 3)
                base += b.max(axis=0)
                cut += np.stack([b[:, 0], q[:, 1], q[:, 2]], 1).max(axis=0)  # primaries untouched
                lane_steps += b.sum(axis=0)
                lane_cut += q.sum(axis=0)
    for j, name in enumerate(("primary", "shadow", "ibl")):
        print(f"  {name:8s}: lane-steps {lane_steps[j]:.3g} -> {lane_cut[j]:.3g} ({lane_cut[j] / max(lane_steps[j], 1):.2f}); "
              f"lockstep wave iterations {base[j]:.0f} -> {cut[j] if j else base[j]:.0f}")
    print(f"  total wave iterations {base.sum():.0f} -> {cut.sum():.0f} = {base.sum() / cut.sum():.3f}x fewer with a perfect certificate")


def sorted_groups_model(rows="480:608"):
    """Upper bound for re-dealing the IBL (and sun) rays of a WORKGROUP of G waves by predicted length
    before tracing them: a perfect predictor = sort by the actual step count, then cut into waves."""
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    S, tw, th = 4, 4, 4
    for G in (1, 2, 4, 8, 16):
        tot = np.zeros(3)
        srt = np.zeros(3)
        for ty in range(0, R, th):
            for tx0 in range(0, W, tw * G):
                per_round = [[[] for _ in range(3)] for _ in range(spp // S)]
                for g in range(G):
                    tx = tx0 + g * tw
                    lanes = [pixels[y * W + x] for y in range(ty, min(ty + th, R)) for x in range(tx, min(tx + tw, W))]
                    P = np.zeros((len(lanes), spp, 3))
                    for li, rays in enumerate(lanes):
                        s = -1
                        for kind, steps, mask in rays:
                            kind = int(kind) & 0xFF
                            if kind == 2:
                                s += 1
                            P[li, s, {2: 0, 7: 1}.get(kind, 2)] = steps
                    for r in range(spp // S):
                        blk = P[:, r * S:(r + 1) * S, :].reshape(-1, 3)
                        tot += blk.max(axis=0)
                        for j in range(3):
                            per_round[r][j].append(blk[:, j])
                for r in range(spp // S):
                    for j in range(3):
                        allr = np.sort(np.concatenate(per_round[r][j]))[::-1]
                        srt[j] += sum(allr[i] for i in range(0, len(allr), 64))
        print(f"  workgroup of {G:2d} waves: shadow {tot[1] / srt[1]:.2f}x ibl {tot[2] / srt[2]:.2f}x fewer iterations; "
              f"total {tot.sum() / (tot[0] + srt[1] + srt[2]):.3f}x (perfect length predictor)")


def predictor_model(rows="480:608", want_kind=3):
    """Like sorted_groups_model, but the rays of a workgroup are sorted by a feature known BEFORE tracing
    (the direction's elevation d.y, logged by the emulator) instead of by their true length."""
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    S, tw, th = 4, 4, 4
    feats, lens = [], []
    for G in (4, 8, 16):
        tot = srt = per = 0.0
        for ty in range(0, R, th):
            for tx0 in range(0, W, tw * G):
                for r in range(spp // S):
                    steps, feat = [], []
                    for g in range(G):
                        tx = tx0 + g * tw
                        wave_steps = []
                        for y in range(ty, min(ty + th, R)):
                            for x in range(tx, min(tx + tw, W)):
                                s = -1
                                for kind, st, mask in pixels[y * W + x]:
                                    k = int(kind) & 0xFF
                                    if k == 2:
                                        s += 1
                                    elif k == want_kind and r * S <= s < (r + 1) * S:
                                        wave_steps.append(float(st))
                                        feat.append(float(np.uint32(int(mask) >> 32).view(np.float32)))
                        if wave_steps:
                            tot += max(wave_steps)
                            steps.extend(wave_steps)
                    if not steps:
                        continue
                    steps, feat = np.asarray(steps), np.asarray(feat)
                    if G == 4:
                        feats.append(feat)
                        lens.append(steps)
                    by_len = np.sort(steps)[::-1]
                    by_feat = steps[np.argsort(feat)]  # ascending elevation: grazing rays first
                    per += sum(by_len[i] for i in range(0, len(by_len), 64))
                    srt += sum(by_feat[i:i + 64].max() for i in range(0, len(by_feat), 64))
        print(f"  kind-{want_kind} rays, workgroup of {G:2d} waves: sorted by the feature {tot / srt:.2f}x fewer iterations (perfect predictor {tot / per:.2f}x)")
    f, l = np.concatenate(feats), np.concatenate(lens)
    print("  rank correlation of steps with d.y:", float(np.corrcoef(np.argsort(np.argsort(f)), np.argsort(np.argsort(l)))[0, 1]))


def pairing_model(rows="480:608"):
    """Static pairing inside a wave: the G most grazing IBL rays (lowest cos(normal, ray), known before
    tracing) are cut in two, the far half goes to the lane with the G-th steepest ray, which traces it
    after its own.  Assumes the cut halves the steps (+ `ov` steps per segment).  Needs an emulator built
    with -DF3D_MODEL_HINT="dot(n,ei)"."""
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    S, tw, th = 4, 4, 4
    base = 0.0
    alt = {(G, ov, parts): 0.0 for G in (4, 8, 16, 24) for ov in (3, 6) for parts in (2, 3)}
    for ty in range(0, R, th):
        for tx in range(0, W, tw):
            for r in range(spp // S):
                steps, key = [], []
                for y in range(ty, min(ty + th, R)):
                    for x in range(tx, min(tx + tw, W)):
                        s = -1
                        for kind, st, mask in pixels[y * W + x]:
                            k = int(kind) & 0xFF
                            if k == 2:
                                s += 1
                            elif k == 3 and r * S <= s < (r + 1) * S:
                                steps.append(float(st))
                                key.append(float(np.uint32(int(mask) >> 32).view(np.float32)))
                if not steps:
                    continue
                steps, key = np.asarray(steps), np.asarray(key)
                base += steps.max()
                order = np.argsort(key)  # grazing first
                n = len(steps)
                for (G, ov, parts), _ in alt.items():
                    g = min(G, n // (parts if parts > 1 else 2))
                    t = steps.copy()
                    for i in range(g):
                        donor = order[i]
                        piece = steps[donor] / parts + ov
                        t[donor] = piece
                        for p in range(1, parts):
                            helper = order[n - 1 - (i * (parts - 1) + (p - 1))]
                            t[helper] = t[helper] + piece
                    alt[(G, ov, parts)] += t.max()
    for (G, ov, parts), v in alt.items():
        print(f"  {G:2d} donors cut in {parts}, overhead {ov}: IBL wave iterations {base / v:.2f}x fewer")


def tail_occupancy(rows="480:608"):
    """Share of a phase's wave iterations that run with at most k lanes still marching (S = 4 tiles)."""
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    S, tw, th = 4, 4, 4
    ks = (1, 2, 4, 8, 16, 32)
    tot = np.zeros(3)
    low = np.zeros((3, len(ks)))
    for ty in range(0, R, th):
        for tx in range(0, W, tw):
            lanes = [pixels[y * W + x] for y in range(ty, min(ty + th, R)) for x in range(tx, min(tx + tw, W))]
            P = np.zeros((len(lanes), spp, 3))
            for li, rays in enumerate(lanes):
                s = -1
                for kind, steps, mask in rays:
                    k = int(kind) & 0xFF
                    if k == 2:
                        s += 1
                    P[li, s, {2: 0, 7: 1}.get(k, 2)] = steps
            for r0 in range(0, spp, S):
                blk = P[:, r0:r0 + S, :].reshape(-1, 3)
                for j in range(3):
                    st = np.sort(blk[:, j])[::-1]
                    tot[j] += st[0]
                    for i, k in enumerate(ks):
                        low[j, i] += st[0] - (st[k] if k < len(st) else 0.0)  # iterations after only k lanes are left
    for j, name in enumerate(("primary", "shadow", "ibl")):
        print(f"  {name:8s}: " + ", ".join(f"<= {k} lanes: {low[j, i] / tot[j]:.0%}" for i, k in enumerate(ks)))


def drain_model(rows="480:608"):
    """How well are the deferred leaf solves packed?  Replays the leaf positions of the first 32 steps of
    every ray (leaf_mask) through the wave's FIFO rule (drain when a FIFO holds 4 or nobody marches) and
    reports drain iterations (each = one leaf solve per lane with something queued) against the ideal."""
    from emul import emul
    from forge3d_amd import datasets

    rows = tuple(int(x) for x in rows.split(":"))
    dem, cam, kw = datasets.rainier_proxy_scene(2048)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, 1920, 1080, cam, rows=rows, **dict(kw, spp=8, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, R, spp, pixels = load_log(log)
    os.unlink(log)
    S, tw, th = 4, 4, 4
    march_it = np.zeros(3)
    drain_it = np.zeros(3)
    solves = np.zeros(3)
    for ty in range(0, R, th):
        for tx in range(0, W, tw * 3):  # every third tile: python loop
            lanes = [pixels[y * W + x] for y in range(ty, min(ty + th, R)) for x in range(tx, min(tx + tw, W))]
            per = [[[] for _ in range(3)] for _ in range(spp // S)]
            for rays in lanes:
                s = -1
                for kind, steps, mask in rays:
                    k = int(kind) & 0xFF
                    if k == 2:
                        s += 1
                    per[s // S][{2: 0, 7: 1}.get(k, 2)].append((int(steps), int(mask) & 0xFFFFFFFF))
            for r in range(spp // S):
                for j in range(3):
                    rays = per[r][j]
                    if not rays:
                        continue
                    n = max(st for st, _ in rays)
                    march_it[j] += n
                    q = [0] * len(rays)
                    for it in range(n):
                        for i, (st, mask) in enumerate(rays):
                            if it < st and it < 32 and (mask >> it) & 1:
                                q[i] += 1
                        marching = any(it + 1 < st for st, _ in rays)
                        if max(q) >= 4 or (not marching and max(q) > 0):
                            drain_it[j] += max(q)
                            solves[j] += sum(q)
                            q = [0] * len(rays)
    for j, name in enumerate(("primary", "shadow", "ibl")):
        print(f"  {name:8s}: march iterations {march_it[j]:.0f}, drain iterations {drain_it[j]:.0f} "
              f"(x ~3 march steps each), lane utilisation of the drains {solves[j] / max(1.0, 64 * drain_it[j]):.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "batched":
        batched_table(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] == "ibl":
        ibl_batched_bound(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2].startswith("lanes"):
        lanes_model(sys.argv[1], int(sys.argv[2][5:] or 8))
    elif len(sys.argv) > 2 and sys.argv[2] == "split":
        split_model(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] == "persistent":
        persistent_model(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] == "walkout":
        walkout_model(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] == "sorted":
        sorted_groups_model(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] == "predictor":
        predictor_model(sys.argv[1], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
    elif len(sys.argv) > 2 and sys.argv[2] == "pairing":
        pairing_model(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] == "tail":
        tail_occupancy(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] == "drain":
        drain_model(sys.argv[1])
    else:
        main()
