#!/bin/bash
# mesh walk with the next record requested ahead (walk3) against the select form (walk2) on the C4 stand-in;
# headline: the terrain-only kernel against the mesh-capable one forced onto the terrain-only scene
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep; L=$R/gpurun_out/keep/mesh_ab8.log; : > $L
for n in walk2 walk3 walk2 walk3; do
  F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_$n.so timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | sed "s/^/$n /" | tee -a $L
done
STEPS=16 bash tools/gpu_variant_ab.sh meshfirst walk2 2>&1 | tee -a $L
F3D_FORCE_MESH_KERNEL=1 STEPS=16 bash tools/gpu_variant_ab.sh walk2 walk3 2>&1 | sed 's/^/forced mesh kernel: /' | tee -a $L
STEPS=16 bash tools/gpu_variant_ab.sh meshfirst walk2 2>&1 | tee -a $L
