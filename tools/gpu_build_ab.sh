#!/bin/bash
# Build a library variant with extra -D flags ON THE GPU BOX and bench it (prefer tools/build_variant.sh + gpu_variant_ab.sh:
# they compile here and spend no box time on the compiler): tools/gpu_build_ab.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT; mkdir -p build_ab; export F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_onbox.so
for FLAGS in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 ${OPT:--O3} -std=c++17 -fPIC -shared -ffp-contract=off ${SLP:--fno-slp-vectorize} $FLAGS \
      forge3d_amd/csrc/f3d_kernels.hip forge3d_amd/csrc/f3d_host.hip forge3d_amd/csrc/f3d_denoise.hip forge3d_amd/csrc/f3d_smoke.hip forge3d_amd/csrc/f3d_lbvh.hip forge3d_amd/csrc/f3d_wavefront.hip forge3d_amd/csrc/f3d_aether_bake.hip -o build_ab/libf3dhip_onbox.so 2> gpurun_out/build_ab.err || { echo "build failed: $FLAGS"; tail -5 gpurun_out/build_ab.err; continue; }
  for v in ${VARIANTS:-0}; do
    python bench.py --steps ${STEPS:-16} --warmup 4 --variant $v --no-cpu-baseline --extra-windows 2 --no-terrain-filling 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('flags [$FLAGS] variant $v: %.1f Msamples/s (windows ms %s)' % (d['value'], d.get('windows_ms_per_step')))"
  done
done
