#!/usr/bin/env python
"""Single-GPU rehearsal of the multi-GPU strip balancing (forge3d_amd/distributed.py).

For N strips of the headline frame: time every strip of the equal partition and of each re-balanced
partition on THIS GPU (one after the other), exactly as the ranks of a real job would, and report the
compute-only bound on strong scaling T(full frame) / max_i T(strip i).  Communication (123 KB halo per
neighbour and frame) is not included.

    python tools/strip_balance.py [--worlds 2,4,8] [--bands B] [--streams S] [--variant V] [--frames K]
    python tools/strip_balance.py --sweep      # bands x streams grid on the 8-strip partition
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.distributed import HALO_ROWS, ROW_COST_FLOOR, HipBackend, partition_rows, rebalance, strip_rows  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--worlds", default="2,4,8")
ap.add_argument("--bands", type=int, default=0)
ap.add_argument("--streams", type=int, default=0)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--fd", type=int, default=0, help="frames in flight of the strip sessions (f3d_session_opts.frames_in_flight)")
ap.add_argument("--config", default="c2", help="c2: the 1080p headline frame; c4: BASELINE.json configs[3] stand-in, 4096^2 with the 600 000-triangle mesh")
ap.add_argument("--costmap", action="store_true", help="cut the strips from ONE row-cost map of the full frame (round 5: StripRenderer's default) "
                                                       "instead of iterating measured probes; --rounds N then adds N measured refinement rounds")
args = ap.parse_args()

W, H, SPP = 1920, 1080, 8
LOOP_FRAMES = 32
dem, cam, kw = datasets.rainier_proxy_scene(2048)
kw = dict(kw, spp=SPP, max_frames=64, min_frames=64, variance_threshold=1e30, memory_budget_bytes=8 << 30,
          kernel_variant=args.variant)
if args.config == "c4":
    W, H, LOOP_FRAMES = 4096, 4096, 16  # 128 spp
    v, i = datasets.proxy_buildings(dem, kw["spacing"][0])
    kw.update(mesh_vertices=v, mesh_indices=i, memory_budget_bytes=24 << 30)
backend = HipBackend(0)


def probe(b0, b1, **extra):
    loop = bool(extra.get("frames_in_flight"))  # as StripRenderer._balance: strips with frames in flight are balanced on the 32-frame loop
    return min(backend.probe(dem, W, H, cam, b0, b1, dict(kw, **extra), frames=LOOP_FRAMES if loop else args.frames, whole_loop=loop) for _ in range(2))


full = probe(0, H)
print(json.dumps({"full_frame_ms": full}), flush=True)
if args.fd:
    print(json.dumps({"full_frame_ms_frames_in_flight": probe(0, H, frames_in_flight=min(args.fd, 12)), "fd": min(args.fd, 12)}), flush=True)
if args.sweep:
    # the balanced 8-strip partition of profiles/r01_strip_balance.log
    bounds = [0, 394, 483, 552, 624, 703, 801, 917, 1080]
    for bands, streams in ((1, 0), (3, 2), (4, 2), (4, 4), (6, 3), (6, 4), (8, 4), (8, 8), (12, 4)):
        times = [probe(bounds[r], bounds[r + 1], bands=bands, band_streams=streams) for r in range(8)]
        print(json.dumps({"world": 8, "bands": bands, "streams": streams, "ms": [round(t, 3) for t in times],
                          "compute_bound_speedup": full / max(times)}), flush=True)
    for bands, streams in ((2, 2), (4, 4), (8, 4)):  # does the whole frame gain from overlapping its own tail?
        print(json.dumps({"world": 1, "bands": bands, "streams": streams,
                          "ms": probe(0, H, bands=bands, band_streams=streams)}), flush=True)
    sys.exit(0)
for world in [int(x) for x in args.worlds.split(",")]:
    bounds = [strip_rows(H, world, r)[0] for r in range(world)] + [H]
    density = np.ones(H)
    if args.costmap:
        import time as _time

        # round 6: every rank probes its EQUAL share of the rows (here one after the other; in a job at the same time, so the
        # set-up pays the longest share, not the sum)
        parts, share_ms = [], []
        for r in range(world):
            t0 = _time.perf_counter()
            parts.append(np.asarray(backend.row_costs(dem, W, H, cam, kw, row_begin=strip_rows(H, world, r)[0], row_end=strip_rows(H, world, r)[1]), np.float64))
            share_ms.append((_time.perf_counter() - t0) * 1e3)
        density = np.concatenate(parts)
        bounds = partition_rows(density + ROW_COST_FLOOR * density.mean(), world, HALO_ROWS)
        print(json.dumps({"world": world, "cost_map_ms_longest_share": round(max(share_ms), 2), "cost_map_ms_all_shares_in_turn": round(sum(share_ms), 2), "bounds": bounds}), flush=True)
    for it in range(0 if (args.costmap and args.rounds == 4) else args.rounds):
        times = [probe(bounds[r], bounds[r + 1], bands=args.bands, band_streams=args.streams, frames_in_flight=args.fd if world >= 4 else 0) for r in range(world)]
        print(json.dumps({"world": world, "round": it, "bounds": bounds, "ms": [round(t, 3) for t in times],
                          "imbalance": max(times) / (sum(times) / world),
                          "compute_bound_speedup": full / max(times)}), flush=True)
        density = rebalance(density, bounds, times)
        new_bounds = partition_rows(density, world, HALO_ROWS)
        if new_bounds == bounds:
            break
        bounds = new_bounds
    # the partition as a 256-spp render runs it: the WHOLE accumulation loop (frames 0..31 of a fresh session, first frame and
    # short first batches included), per strip, against the same loop of the full frame
    import time

    import torch
    from forge3d_amd.session import TerrainSession

    def loop_ms(b0, b1, **extra):
        best = 1e9
        for _ in range(2):
            with TerrainSession(dem, W, H, cam, row_begin=b0, row_end=b1, **dict(kw, **extra)) as s:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s.enqueue_frames(0, LOOP_FRAMES)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) * 1e3)
        return best

    full_loop = loop_ms(0, H)
    fd = args.fd if world >= 4 else 0  # (as the strip driver: fat strips keep the fused kernel)
    loops = [loop_ms(bounds[r], bounds[r + 1], frames_in_flight=fd) for r in range(world)]
    if args.costmap and max(loops) / (sum(loops) / world) > 1.05:
        # bench.py's one re-cut from what the first frames cost each rank on these strips (recut_after_warmup)
        recut = partition_rows(rebalance(density, bounds, loops), world, HALO_ROWS)
        if recut != bounds:
            print(json.dumps({"world": world, "recut": "max / mean of the strips' loops %.3f > 1.05" % (max(loops) / (sum(loops) / world)),
                              "before": {"bounds": bounds, "strips_ms": [round(t, 3) for t in loops], "compute_bound_speedup_256spp": full_loop / max(loops)}}), flush=True)
            bounds = recut
            loops = [loop_ms(bounds[r], bounds[r + 1], frames_in_flight=fd) for r in range(world)]
    print(json.dumps({"config": args.config, "cut": "cost map" if args.costmap else "measured rounds", "world": world, "render_256spp_loop_ms": {"full_frame": round(full_loop, 3), "strips": [round(t, 3) for t in loops]},
                      "ms_per_strip_frame": round(max(loops) / LOOP_FRAMES, 4), "compute_bound_speedup_256spp": full_loop / max(loops), "bounds": bounds}), flush=True)
