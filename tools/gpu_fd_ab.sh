#!/bin/bash
# Build libf3dhip.so with extra -D flags ON THE GPU BOX and run the 8-strip rehearsal with frames in flight:
#   tools/gpu_fd_ab.sh "<flags A>" "<flags B>" ...     (FD=frames in flight, default 16; ARGS=extra strip_balance args)
cd $GRAFT_REPO_ROOT
for FLAGS in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize $FLAGS \
      forge3d_amd/csrc/f3d_kernels.hip forge3d_amd/csrc/f3d_host.hip forge3d_amd/csrc/f3d_denoise.hip forge3d_amd/csrc/f3d_smoke.hip \
      forge3d_amd/csrc/f3d_lbvh.hip forge3d_amd/csrc/f3d_wavefront.hip forge3d_amd/csrc/f3d_aether_bake.hip -o forge3d_amd/libf3dhip.so 2> gpurun_out/build_ab.err || { echo "build failed: $FLAGS"; tail -5 gpurun_out/build_ab.err; continue; }
  echo "== flags [$FLAGS]"
  python tools/strip_balance.py --worlds 8 --rounds ${ROUNDS:-4} --frames ${FRAMES:-32} --fd ${FD:-16} ${ARGS:-} 2>&1 | grep -v amdgpu.ids | cut -c1-320
done
