#!/bin/bash
# every configuration of the bench line with / without -mllvm -amdgpu-opt-vgpr-liverange=false
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/keep
for n in nolr final; do F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_$n.so timeout 280 python bench.py --steps 16 --no-cpu-baseline --extra-windows 1 2>/dev/null | tail -1 > gpurun_out/keep/bench_$n.json; done
python - <<PY
import json
for n in ("nolr", "final"):
    d = json.load(open("gpurun_out/keep/bench_%s.json" % n)); c = d["configs"]
    print(n, round(d["value"]), round(d["config_terrain_filling"]["value"]), {k: (round(v["value"], 1) if "value" in v else v) for k, v in c.items()}, c["C5"].get("kernel_ms"))
PY
