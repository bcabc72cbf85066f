#!/bin/bash
# tile maps / dispatch order on the C4 stand-in (does the mesh walk want locality more than balance?)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep; L=$R/gpurun_out/keep/mesh_ab11.log; : > $L
for v in 0 4000 1000 3000 0 4000; do
  C4_VARIANT=$v timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | tee -a $L
done
