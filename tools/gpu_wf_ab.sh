#!/bin/bash
# Build libf3dhip.so with extra -D flags ON THE GPU BOX and time the adjudication gate render (512^2 x 4096 frames):
#   tools/gpu_wf_ab.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
for FLAGS in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize $FLAGS \
      forge3d_amd/csrc/f3d_kernels.hip forge3d_amd/csrc/f3d_host.hip forge3d_amd/csrc/f3d_denoise.hip forge3d_amd/csrc/f3d_smoke.hip \
      forge3d_amd/csrc/f3d_lbvh.hip forge3d_amd/csrc/f3d_wavefront.hip forge3d_amd/csrc/f3d_aether_bake.hip -o forge3d_amd/libf3dhip.so 2> gpurun_out/build_ab.err || { echo "build failed: $FLAGS"; tail -5 gpurun_out/build_ab.err; continue; }
  python - <<PY
from forge3d_amd import wavefront as w
best = None
for _ in range(3):
    out = w.render_scene(w.adjudication_scene(), ${SIZE:-512}, ${SIZE:-512}, ${FRAMES:-4096})
    best = out if best is None or out["loop_seconds"] < best["loop_seconds"] else best
print("flags [$FLAGS]: kernel %.1f ms, %.2f Gpaths/s, %.2f Gvertices/s" % (best["loop_seconds"] * 1e3, best["paths"] / best["loop_seconds"] / 1e9, best["path_vertices"] / best["loop_seconds"] / 1e9))
PY
done
