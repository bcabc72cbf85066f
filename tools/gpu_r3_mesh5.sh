#!/bin/bash
# register budgets of the frame kernel on the C4 stand-in (the mesh walk shifts the balance?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/keep; L=gpurun_out/keep/mesh_ab5.log; : > $L
for v in 0 104 105 107 0 104 105; do
  C4_VARIANT=$v F3D_HIP_LIBRARY=$PWD/build_ab/libf3dhip_terrfirst.so timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | tee -a $L
done
