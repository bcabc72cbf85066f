#!/bin/bash
# One rocprofv3 --pmc pass of the headline bench for an ad-hoc counter list: tools/gpu_pmc_once.sh TAG COUNTER...
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/p -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --extra-windows 0 --no-terrain-filling > $OUT/log.txt 2>&1
cd $R; python tools/rocpd_summary.py $OUT 2>&1 | grep -i "k_frame\|k_head" | head -20
