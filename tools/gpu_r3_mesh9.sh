#!/bin/bash
# slab test as one fma per plane (walk4) against sub + mul (walk2) on the C4 stand-in; mesh tests on the in-tree library
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep; L=$R/gpurun_out/keep/mesh_ab9.log; : > $L
for n in walk2 walk4 walk2 walk4; do
  F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_$n.so timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | sed "s/^/$n /" | tee -a $L
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mesh or config4 or bvh" 2>&1 | tail -3 | tee -a $L
