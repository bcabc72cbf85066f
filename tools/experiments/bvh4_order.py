#!/usr/bin/env python
"""Node visits of the 4-wide mesh walk with and without near-first ordering of the entered children, counted by the host
emulator on a small window of BASELINE configs[3] (no GPU needed):

    F3D_EMUL_CXXFLAGS="-DF3D_MESH_STATS_HOST" python tools/experiments/bvh4_order.py
    F3D_EMUL_CXXFLAGS="-DF3D_MESH_STATS_HOST -DF3D_BVH4_ORDERED" python tools/experiments/bvh4_order.py

slots: [0] nodes visited, [2] leaf blocks entered, [4] walks."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from emul import emul  # noqa: E402

from forge3d_amd import datasets  # noqa: E402

dem, cam, kw = datasets.rainier_proxy_scene(512)
v, i = datasets.proxy_buildings(dem, kw["spacing"][0])
size = int(sys.argv[1]) if len(sys.argv) > 1 else 192
out = (C.c_ulonglong * 8)()
emul.lib().emul_mesh_stats(out, 1)
r = emul.render(dem, size, size, cam, mesh_vertices=v, mesh_indices=i, **dict(kw, spp=2, max_frames=2, min_frames=2, variance_threshold=1e30))
emul.lib().emul_mesh_stats(out, 1)
print("triangles", len(i) // 3 if np.ndim(i) == 1 else len(i), "nodes visited", out[0], "leaf blocks", out[2], "walks", out[4],
      "nodes/walk %.2f" % (out[0] / max(1, out[4])), "checksum", int(np.asarray(r["rgba"], dtype=np.uint64).sum()))
