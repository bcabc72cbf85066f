#!/bin/bash
# The four parity fuzzers on fresh seeds, each bounded in time:  tools/gpu_round_fuzz.sh r04 [scale of the oracle fuzzers] [scale of the other two]
#   scenes vs the oracle | frames in flight vs the fused kernel | strips with halos vs one strip | PBR tracer vs its oracle
TAG=${1:-r04}; K=${2:-1}; K2=${3:-$K}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export OMP_NUM_THREADS=8
L=gpurun_out/${TAG}_fuzz.log; : > $L
run() { local t0=$(date +%s); echo "== $*" >> $L; timeout 170 python "$@" 2>&1 | tail -1 >> $L; echo "   ($(( $(date +%s) - t0 )) s)" >> $L; }
run tools/gpu_fuzz.py 640000 $((45 * K))
run tools/gpu_fuzz_fd.py 92000 $((600 * K2))
run tools/gpu_fuzz_strips.py 72000 $((300 * K2))
run tools/gpu_fuzz_wavefront.py 34000 $((40 * K))
run tools/gpu_fuzz_smoke.py 5000 $((50 * K))
cat $L
