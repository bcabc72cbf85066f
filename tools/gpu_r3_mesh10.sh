#!/bin/bash
# 16-byte quantised BVH records (walk5) against 32-byte float records (walk4) on the C4 stand-in; mesh + LBVH tests
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep; L=$R/gpurun_out/keep/mesh_ab10.log; : > $L
for n in walk4 walk5 walk4 walk5; do
  F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_$n.so timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | sed "s/^/$n /" | tee -a $L
done
F3D_MESH_BVH=lbvh F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_walk5.so timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | sed "s/^/walk5 lbvh /" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mesh or config4 or bvh" 2>&1 | tail -3 | tee -a $L
