#!/bin/bash
# C3 GI (1080p x 64 paths a pixel) with the tree's library under environment settings:  tools/gpu_wf_env_ab.sh "NAME=value" ...   ("-" = none)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for kv in "$@"; do
  ( [ "$kv" != "-" ] && export "$kv"; python - "$kv" <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, warnings
sys.path.insert(0, ".")
from forge3d_amd import atmosphere, datasets, offline
dem, cam, kw = datasets.rainier_proxy_scene(2048)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    handle = atmosphere.AtmosphereLutHandle.load_shipped(atmosphere.AtmosphereConfig(turbidity=2.0))
k = dict(spacing=kw["spacing"], exaggeration=kw["exaggeration"], sun_azimuth_deg=kw["sun_azimuth_deg"], sun_elevation_deg=kw["sun_elevation_deg"],
         sun_intensity=kw["sun_intensity"], atmosphere=handle, memory_budget_bytes=8 << 30)
offline.render_terrain_gi(dem, 1920, 1080, cam, spp=8, **k)
gi = min(offline.render_terrain_gi(dem, 1920, 1080, cam, spp=64, **k)["gi_seconds"] for _ in range(3))
print("%-28s C3 GI 1080p x 64: %.2f ms = %.0f Mpaths/s" % (sys.argv[1], gi * 1e3, 1920 * 1080 * 64 / gi / 1e6))
PY
  )
done; done
