#!/bin/bash
# second A/B call: the C4 stand-in with deferred / inline triangle tests; the headline with the mesh code compiled out
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/keep; L=gpurun_out/keep/mesh_ab2.log; : > $L
for n in meshinl meshq meshinl meshq; do
  F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_$n.so timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | sed "s/^/$n /" | tee -a $L
done
STEPS=16 bash tools/gpu_variant_ab.sh nomesh meshinl nomesh 2>&1 | tee -a $L
