#!/bin/bash
# occluded(): terrain first (new) against mesh first (the reference's order) -- C4 stand-in, headline, mesh tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/keep; L=gpurun_out/keep/mesh_ab4.log; : > $L
for n in meshfirst terrfirst meshfirst terrfirst; do
  F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_$n.so timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | sed "s/^/$n /" | tee -a $L
done
STEPS=16 bash tools/gpu_variant_ab.sh meshfirst terrfirst meshfirst terrfirst 2>&1 | tee -a $L
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mesh or config4 or bvh" 2>&1 | tail -3 | tee -a $L
