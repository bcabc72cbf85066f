#!/bin/bash
# the select-form mesh walk on the C4 stand-in; the terrain-only kernel with aligned loops on the headline
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep; L=$R/gpurun_out/keep/mesh_ab7.log; : > $L
for n in tmpl walk2 walk2al tmpl walk2; do
  F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_$n.so timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | sed "s/^/$n /" | tee -a $L
done
STEPS=16 bash tools/gpu_variant_ab.sh meshfirst walk2 walk2al meshfirst walk2 walk2al 2>&1 | tee -a $L
