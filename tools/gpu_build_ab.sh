#!/bin/bash
# Build libf3dhip.so with extra -D flags ON THE GPU BOX and bench it: tools/gpu_build_ab.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
for FLAGS in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 ${OPT:--O3} -std=c++17 -fPIC -shared -ffp-contract=off ${SLP:--fno-slp-vectorize} $FLAGS \
      forge3d_amd/csrc/f3d_kernels.hip forge3d_amd/csrc/f3d_host.hip forge3d_amd/csrc/f3d_denoise.hip forge3d_amd/csrc/f3d_smoke.hip forge3d_amd/csrc/f3d_lbvh.hip forge3d_amd/csrc/f3d_wavefront.hip forge3d_amd/csrc/f3d_aether_bake.hip -o forge3d_amd/libf3dhip.so 2> gpurun_out/build_ab.err || { echo "build failed: $FLAGS"; tail -5 gpurun_out/build_ab.err; continue; }
  for v in ${VARIANTS:-0}; do
    python bench.py --steps ${STEPS:-16} --warmup 4 --variant $v --no-cpu-baseline --extra-windows 2 --no-terrain-filling 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('flags [$FLAGS] variant $v: %.1f Msamples/s (windows ms %s)' % (d['value'], d.get('windows_ms_per_step')))"
  done
done
