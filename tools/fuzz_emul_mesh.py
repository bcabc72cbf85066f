#!/usr/bin/env python
"""The mesh walk of csrc/f3d_shade.h on the host emulator against the oracle's sweep over all triangles, on the random scenes
of tests/scenes.py that carry a mesh (terrain + mesh, every output the same bits): python tools/fuzz_emul_mesh.py first_seed count [inside]"""
import sys, time
import numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import scenes
from emul import emul
from oracle import oracle
first, count = int(sys.argv[1]), int(sys.argv[2])
inside = len(sys.argv) > 3 and sys.argv[3] == "inside"  # meshes inside the DEM's footprint: the mesh-band form of the march (csrc/f3d_meshgrid.h)
bad, done, t0 = [], 0, time.time()
seed = first
while done < count and seed < first + 40 * count:
    dem, size, cam, kw = scenes.random_scene_city_inside(seed) if inside else scenes.random_scene(seed)
    seed += 1
    if kw.get("mesh_vertices") is None:
        continue
    kw = dict(kw, max_frames=3, min_frames=3, variance_threshold=1e30)
    try:
        want = oracle.render(dem, size[0], size[1], cam, **kw)
        got = emul.render(dem, size[0], size[1], cam, **kw)
    except Exception as exc:
        print("seed", seed - 1, "error", str(exc)[:100]); continue
    done += 1
    ok = all(np.array_equal(got[k], want[k]) for k in ("rgba", "albedo", "normal")) and np.array_equal(got["depth"], want["depth"], equal_nan=True)
    if not ok: bad.append(seed - 1)
print("%d mesh scenes from seed %d: mismatches %s, %.0f s" % (done, first, bad, time.time() - t0))
