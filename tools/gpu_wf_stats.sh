#!/bin/bash
# Statistics builds of the PBR tracer (-DF3D_WF_STATS=k: build_ab/libf3dhip_st<k>.so): what the "path vertices" counter holds instead --
# 1 expensive phases, 2 the sum of their pending lanes, 3 / 4 phases with <= 32 / <= 16 pending lanes (C3 GI, 1080p x 64).
cd $GRAFT_REPO_ROOT
for k in "$@"; do
F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_st$k.so python - $k <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, warnings
sys.path.insert(0, ".")
from forge3d_amd import atmosphere, datasets, offline
dem, cam, kw = datasets.rainier_proxy_scene(2048)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    handle = atmosphere.AtmosphereLutHandle.load_shipped(atmosphere.AtmosphereConfig(turbidity=2.0))
k = dict(spacing=kw["spacing"], exaggeration=kw["exaggeration"], sun_azimuth_deg=kw["sun_azimuth_deg"], sun_elevation_deg=kw["sun_elevation_deg"],
         sun_intensity=kw["sun_intensity"], atmosphere=handle, memory_budget_bytes=8 << 30)
gi = offline.render_terrain_gi(dem, 1920, 1080, cam, spp=64, **k)
print("stat %s: %d  (%.2f ms)" % (sys.argv[1], gi["path_vertices"], gi["gi_seconds"] * 1e3))
PY
done
