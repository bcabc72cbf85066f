#!/bin/bash
# Cost of the once-per-render passes (k_gbuffer incl. the ray certificates) per library variant: rocprofv3 kernel trace of a
# 2-frame bench.   tools/gpu_gbuffer_cost.sh name ...
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for name in "$@"; do
  OUT=$R/gpurun_out/gb_$name; mkdir -p $OUT
  F3D_HIP_LIBRARY=$R/build_ab/libf3dhip_$name.so timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- \
      python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --extra-windows 0 --no-terrain-filling > $OUT/log.txt 2>&1
  (cd $R; python tools/rocpd_summary.py $OUT 2>&1 | grep "k_gbuffer\|k_frame\|k_horizon" | awk -v n=$name '{print n, $0}') | tee -a $R/gpurun_out/gbuffer_cost.log
done
