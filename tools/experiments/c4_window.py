"""Time BASELINE.json configs[3]'s stand-in (proxy DEM + 600 000 triangles, 4096 x 4096) alone and print a digest of the
image: python tools/experiments/c4_window.py [frames]   (library from F3D_HIP_LIBRARY for an A/B)"""
import hashlib
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from forge3d_amd import datasets  # noqa: E402
from forge3d_amd.session import TerrainSession  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dem, cam, kw = datasets.rainier_proxy_scene(2048)
v, i = datasets.proxy_buildings(dem, kw["spacing"][0])
mesh = {} if os.environ.get("C4_NO_MESH") == "1" else dict(mesh_vertices=v, mesh_indices=i)
kw = dict(kw, spp=8, max_frames=2 + 2 * frames, min_frames=2 + 2 * frames, variance_threshold=1e30)
t_setup = time.perf_counter()
with TerrainSession(dem, size, size, cam, device=0, memory_budget_bytes=16 << 30, kernel_variant=int(os.environ.get('C4_VARIANT', '0')), **mesh, **kw) as s:
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    s.enqueue_frames(0, 2)
    torch.cuda.synchronize()
    rates = []
    for w in range(2):
        t0 = time.perf_counter()
        s.enqueue_frames(2 + w * frames, frames, True)
        s.window_stats()
        dt = time.perf_counter() - t0
        rates.append(size * size * 8 * frames / dt / 1e6)
    img = s.resolve(2 + 2 * frames)
digest = hashlib.sha256(np.ascontiguousarray(img["rgba"]).tobytes()).hexdigest()[:16]
print("variant %s " % os.environ.get("C4_VARIANT", "0") + "C4 %d^2 %d tris: %s Msamples/s  image %s  setup %.0f ms" % (size, 0 if not mesh else i.shape[0], ["%.0f" % r for r in rates], digest, t_setup * 1e3))

if os.environ.get("C4_MESH_STATS") == "1":  # a -DF3D_MESH_STATS build (tools/build_variant.sh): what a wave pays for the mesh walk
    import ctypes

    lib = ctypes.CDLL(os.environ["F3D_HIP_LIBRARY"])
    out = (ctypes.c_ulonglong * 8)()
    lib.f3d_debug_mesh_stats(out, 0)
    wi, li, wl, ll, ww, lw = [int(v) for v in out[:6]]
    print("mesh walk: %d wave walks (%.1f lanes), %.1f iterations per wave walk, lanes active per iteration %.1f; "
          "leaf block in %.1f %% of the iterations with %.1f lanes" % (ww, lw / max(ww, 1), wi / max(ww, 1), li / max(wi, 1), 100.0 * wl / max(wi, 1), ll / max(wl, 1)))
