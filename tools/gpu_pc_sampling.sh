#!/bin/bash
# PC sampling of the headline bench (rocprofv3 beta feature): tools/gpu_pc_sampling.sh [method] [unit] [interval]
# writes gpurun_out/pcs/* ; short run, hard timeouts (a beta feature must not hang the box)
M=${1:-stochastic}; U=${2:-cycles}; I=${3:-1048576}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pcs_$M; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I --kernel-trace --output-format csv -d $OUT/p -o bench -- \
  python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --extra-windows 0 --no-terrain-filling --no-configs > $OUT/log.txt 2>&1
echo "rc=$?"; tail -3 $OUT/log.txt; find $OUT -type f | head; for f in $(find $OUT -name "*pc_sampling*.csv" | head -2); do wc -l $f; head -3 $f; done
