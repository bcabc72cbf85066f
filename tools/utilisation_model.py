#!/usr/bin/env python
"""SIMT lane-utilisation model of the frame kernel, from the CPU emulator's per-ray step logs.

Runs the headline scene (reduced resolution) through tests/emul with F3D_EMUL_RAYLOG, then
replays the logged per-ray step sequences (inner-node step / fat-leaf step) through models of
how a 64-lane wave would schedule them:

  nested      the shipped kernel: per sample, three phases (primary / shadow / IBL); in each
              phase iteration k runs the inner body if ANY lane's k-th step is an inner step
              and the leaf body if any lane's is a leaf step
  continuous  each lane runs its rays back to back (state machine), same per-iteration rule
  ideal       perfect packing: total lane-steps / 64

Costs: inner step = 1, leaf step = LEAF_COST.  Prints useful-lane fraction per model.
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

LEAF_COST = 2.0


def load_log(path):
    data = open(path, "rb").read()
    W, rows, spp, _ = struct.unpack_from("<4I", data, 0)
    off = 16
    pixels = []
    for _ in range(W * rows):
        (n,) = struct.unpack_from("<I", data, off)
        off += 4
        rays = []
        for _ in range(n):
            kind, steps, mask = struct.unpack_from("<IIQ", data, off)
            off += 16
            rays.append((kind, steps, mask))
        pixels.append(rays)
    return W, rows, spp, pixels


def seq(steps, mask):
    return [(mask >> k) & 1 for k in range(min(steps, 64))] + [0] * max(0, steps - 64)


def wave_cost(lane_seqs):
    """lockstep cost of a set of per-lane step sequences + the useful lane-cost."""
    n = max((len(s) for s in lane_seqs), default=0)
    cost = useful = 0.0
    for k in range(n):
        kinds = [s[k] for s in lane_seqs if k < len(s)]
        inner = sum(1 for x in kinds if x == 0)
        leaf = len(kinds) - inner
        cost += (1.0 if inner else 0.0) + (LEAF_COST if leaf else 0.0)
        useful += inner * 1.0 + leaf * LEAF_COST
    return cost, useful


def main():
    from emul import emul
    from forge3d_amd import datasets

    w, h, spp = 320, 184, 8
    dem, cam, kw = datasets.rainier_proxy_scene(1024)
    log = tempfile.mktemp(suffix=".raylog")
    os.environ["F3D_EMUL_RAYLOG"] = log
    emul.render(dem, w, h, cam, **dict(kw, spp=spp, max_frames=2, min_frames=2, variance_threshold=1e30))
    W, rows, spp, pixels = load_log(log)
    os.unlink(log)

    tot = {"nested": [0.0, 0.0], "continuous": [0.0, 0.0]}
    lane_steps = 0.0
    n_waves = 0
    for ty in range(0, rows, 8):
        for tx in range(0, W, 8):
            lanes = [pixels[y * W + x] for y in range(ty, min(ty + 8, rows)) for x in range(tx, min(tx + 8, W))]
            n_waves += 1
            # split each lane's rays into samples: a sample starts at each primary ray (kind 2)
            per_lane = []
            for rays in lanes:
                samples, cur = [], None
                for kind, steps, mask in rays:
                    if kind == 2:
                        cur = {"p": seq(steps, mask), "s": [], "i": []}
                        samples.append(cur)
                    elif kind == 7 or (kind == 3 and False):
                        cur["s"] = seq(steps, mask)
                    else:
                        cur["i"] = seq(steps, mask)
                per_lane.append(samples)
            for s in range(spp):
                for ph in ("p", "s", "i"):
                    c, u = wave_cost([l[s][ph] for l in per_lane if s < len(l)])
                    tot["nested"][0] += c * 64
                    tot["nested"][1] += u
            flat = [[x for smp in l for ph in ("p", "s", "i") for x in smp[ph]] for l in per_lane]
            c, u = wave_cost(flat)
            tot["continuous"][0] += c * 64
            tot["continuous"][1] += u
            lane_steps += u
    print(f"{n_waves} waves, leaf cost {LEAF_COST}")
    for k, (c, u) in tot.items():
        print(f"  {k:11s}: lane utilisation {u / c:.3f}   (wave cost {c / 64:.0f})")
    print(f"  ideal      : wave cost {lane_steps / 64:.0f}")


if __name__ == "__main__":
    main()
