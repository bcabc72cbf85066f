#!/usr/bin/env python
"""A short window of one BASELINE configuration (or of a multi-GPU strip), for rocprofv3 (tools/gpu_profile_config.sh):

    python tools/config_workload.py C4          4096 x 4096, proxy DEM + 600 000 triangles, 8 spp x 6 frames  (k_frame<0,6,4,true>)
    python tools/config_workload.py C3_gi       1080p, proxy DEM in the PBR path tracer, 32 paths a pixel       (k_wf_paths<true>)
    python tools/config_workload.py gate        the adjudication scene of the PBR tracer, 512 x 512 x 4096 frames              (k_wf_paths<false>)
    python tools/config_workload.py C5          16 frames of the resident smoke sequence at 1080p: solver phases + march + composite
    python tools/config_workload.py strip       the heaviest eighth of the 1080p headline frame, 16 frames in flight, 48 frames
    python tools/config_workload.py strip_fused the same strip with the fused kernel (k_frame<0,6,8,false>)
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.session import TerrainSession  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "C4"
dem, cam, kw = datasets.rainier_proxy_scene(2048)
kw = dict(kw, spp=8, variance_threshold=1e30)
if len(sys.argv) > 2:  # A/B: kernel_variant of the session (e.g. 4000000 = 4 sample lanes)
    kw["kernel_variant"] = int(sys.argv[2])
t0 = time.perf_counter()
if what == "C4":
    v, i = datasets.proxy_buildings(dem, kw["spacing"][0])
    with TerrainSession(dem, 4096, 4096, cam, memory_budget_bytes=16 << 30, mesh_vertices=v, mesh_indices=i, **dict(kw, max_frames=6, min_frames=6)) as s:
        s.enqueue_frames(0, 2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s.enqueue_frames(2, 4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 4
        print("C4 ms per frame %.3f = %.0f Msamples/s (mesh walk: %s)" % (dt * 1e3, 4096 * 4096 * 8 / dt / 1e6, __import__("os").environ.get("F3D_MESH_BVH", "4-wide")))
elif what == "C3_gi":
    from forge3d_amd import offline

    k = dict(spacing=kw["spacing"], exaggeration=kw["exaggeration"], sun_azimuth_deg=kw["sun_azimuth_deg"], sun_elevation_deg=kw["sun_elevation_deg"],
             sun_intensity=kw["sun_intensity"], memory_budget_bytes=8 << 30)
    gi = offline.render_terrain_gi(dem, 1920, 1080, cam, spp=32, **k)
    print("gi loop ms", gi["gi_seconds"] * 1e3)
elif what == "gate":  # the adjudication gate's path-traced half (no terrain primitive: k_wf_paths<false>), 512^2 x 4096 frames
    from forge3d_amd import wavefront as w

    best = min((w.render_scene(w.adjudication_scene(), 512, 512, 4096) for _ in range(3)), key=lambda o: o["loop_seconds"])
    print("gate gi loop ms", best["loop_seconds"] * 1e3)
elif what == "C5":
    from forge3d_amd import smoke

    dom = smoke.SmokeDomain((96, 64, 128))
    emitters = [smoke.SmokeEmitter(center=(48.0, 6.0, 40.0), radius=7.0, density_rate=9.0, temperature_rate=6.0, soot_rate=0.5,
                                   emission_rate=2.0, velocity=(0.0, 2.0, 0.6))]
    settings = smoke.SmokeStepSettings(dt=0.2, turbulence_strength=0.5, turbulence_seed=7, wind=(0.3, 0.0, 1.0), buoyancy=1.1)
    dom.step(settings, emitters, steps=40)
    view = dict(camera_pos=(48.0, 70.0, -120.0), target=(48.0, 28.0, 64.0), up=(0.0, 1.0, 0.0), fovy_deg=40.0)
    yy, xx = np.mgrid[0:1080, 0:1920]
    terrain = np.stack([(xx * 255 // 1919), (yy * 255 // 1079), np.full_like(xx, 96), np.full_like(xx, 255)], axis=-1).astype(np.uint8)
    seq = smoke.SmokeSequence(dom, terrain, **view)  # the resident sequence bench.py times: solver phases, pack, march, composite per frame
    for _ in seq.frames(16, settings, emitters, overlap=False):  # (one kernel after the other: clean durations; bench.py's sequence runs the solver beside the march)
        pass
elif what in ("strip", "strip_fused"):
    # rows of the heaviest strip of the balanced 8-strip partition (profiles/r04_strip_balance.log)
    b0, b1 = (int(x) for x in __import__("os").environ.get("F3D_STRIP_ROWS", "666,755").split(","))
    fd = 16 if what == "strip" else 0
    with TerrainSession(dem, 1920, 1080, cam, row_begin=b0, row_end=b1, memory_budget_bytes=8 << 30, frames_in_flight=fd,
                        **dict(kw, max_frames=64, min_frames=64)) as s:
        s.enqueue_frames(0, 16)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s.enqueue_frames(16, 32)
        torch.cuda.synchronize()
        print("strip rows", b0, b1, "fd", s.frames_in_flight(), "lanes", s.sample_lanes(), "ms per frame", (time.perf_counter() - t1) / 32 * 1e3)
else:
    raise SystemExit(__doc__)
print(what, "wall s", time.perf_counter() - t0)
