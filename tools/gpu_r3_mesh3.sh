#!/bin/bash
# where does the C4 stand-in's time go: with and without the mesh at 4096^2, and the per-kernel split
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/keep; L=gpurun_out/keep/mesh_ab3.log; : > $L
timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | tee -a $L
C4_NO_MESH=1 timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | sed 's/^/nomesh /' | tee -a $L
timeout 200 python tools/experiments/c4_window.py 4 1024 2>&1 | tail -1 | tee -a $L
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/tools/experiments/c4_window.py 4 > /dev/null 2>&1
f=$(find /tmp/prof_c4 -name '*kernel_stats.csv' | head -1); head -8 "$f" | cut -c1-200 | tee -a $GRAFT_REPO_ROOT/$L
