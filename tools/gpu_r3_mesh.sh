#!/bin/bash
# A/B of the deferred triangle tests in the mesh BVH walk (f3d_shade.h mesh_bvh): headline workload, the C4 stand-in, tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/keep; L=gpurun_out/keep/mesh_ab.log; : > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mesh or config4 or bvh" 2>&1 | tail -3 | tee -a $L
for n in meshinl meshq meshinl meshq; do
  F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_$n.so timeout 300 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | sed "s/^/$n /" | tee -a $L
done
STEPS=16 bash tools/gpu_variant_ab.sh meshinl meshq meshinl meshq 2>&1 | tee -a $L
timeout 600 python tools/gpu_fuzz.py 2>&1 | tail -3 | tee -a $L
