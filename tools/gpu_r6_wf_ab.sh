#!/bin/bash
# A/B of library builds on the PBR tracer's two timings (adjudication gate, C3 GI), same box:  tools/gpu_r6_wf_ab.sh name ...
# (name "tree" = the in-tree library; others build_ab/libf3dhip_<name>.so from tools/build_variant.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/keep; LOG=gpurun_out/keep/r06_wf_ab.log
for rep in 1 2; do for name in "$@"; do
  lib=$PWD/build_ab/libf3dhip_$name.so; [ "$name" = tree ] && lib=$PWD/forge3d_amd/libf3dhip.so
  F3D_HIP_LIBRARY=$lib python - "$name" <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $LOG
import sys, warnings
sys.path.insert(0, ".")
from forge3d_amd import atmosphere, datasets, offline, wavefront as w
name = sys.argv[1]
best = min(w.render_scene(w.adjudication_scene(), 512, 512, 4096)["loop_seconds"] for _ in range(3))
dem, cam, kw = datasets.rainier_proxy_scene(2048)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    handle = atmosphere.AtmosphereLutHandle.load_shipped(atmosphere.AtmosphereConfig(turbidity=2.0))
k = dict(spacing=kw["spacing"], exaggeration=kw["exaggeration"], sun_azimuth_deg=kw["sun_azimuth_deg"], sun_elevation_deg=kw["sun_elevation_deg"],
         sun_intensity=kw["sun_intensity"], atmosphere=handle, memory_budget_bytes=8 << 30)
offline.render_terrain_gi(dem, 1920, 1080, cam, spp=8, **k)
gi = min(offline.render_terrain_gi(dem, 1920, 1080, cam, spp=64, **k)["gi_seconds"] for _ in range(3))
print("%-12s gate 512^2 x 4096: %.1f ms   C3 GI 1080p x 64: %.2f ms = %.0f Mpaths/s" % (name, best * 1e3, gi * 1e3, 1920 * 1080 * 64 / gi / 1e6))
PY
done; done
