#!/bin/bash
# Round 6, the PBR tracer with its path state in LDS rows: device tests, fuzz, timings (one gpurun call):  tools/gpu_r6_wf.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/keep; LOG=gpurun_out/keep/r06_wf.log; : > $LOG
timeout 900 python -m pytest tests/test_wavefront.py tests/test_offline_gi.py tests/test_gpu_parity.py -m gpu -x -q -k "wavefront or gi or offline or adjudication or pbr or Wavefront" 2>&1 | tail -3 | tee -a $LOG
timeout 300 python tools/gpu_fuzz_wavefront.py 61000 ${FUZZ:-300} 2>&1 | tail -1 | tee -a $LOG
python tools/wf_time.py 512 4096 2>&1 | tail -1 | tee -a $LOG
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $LOG
import sys, time, numpy as np
sys.path.insert(0, ".")
from forge3d_amd import atmosphere, datasets, offline
import warnings
dem, cam, kw = datasets.rainier_proxy_scene(2048)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    handle = atmosphere.AtmosphereLutHandle.load_shipped(atmosphere.AtmosphereConfig(turbidity=2.0))
k = dict(spacing=kw["spacing"], exaggeration=kw["exaggeration"], sun_azimuth_deg=kw["sun_azimuth_deg"], sun_elevation_deg=kw["sun_elevation_deg"],
         sun_intensity=kw["sun_intensity"], atmosphere=handle, memory_budget_bytes=8 << 30)
offline.render_terrain_gi(dem, 1920, 1080, cam, spp=8, **k)
for rep in range(3):
    gi = offline.render_terrain_gi(dem, 1920, 1080, cam, spp=64, **k)
    print("C3_gi: %.2f ms for 64 paths/px = %.1f Mpaths/s, %.3f vertices/path" % (gi["gi_seconds"] * 1e3, 1920 * 1080 * 64 / gi["gi_seconds"] / 1e6, gi["path_vertices"] / (1920 * 1080 * 64)))
PY
