#!/bin/bash
# Everything profiles/ needs from one round, in one gpurun call:  tools/gpu_round_measure.sh r04
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/${TAG}_gpu_tests.log; cat gpurun_out/${TAG}_gpu_tests.log
python bench.py > gpurun_out/${TAG}_bench_1gpu.json 2> gpurun_out/${TAG}_bench_1gpu.err; tail -c 600 gpurun_out/${TAG}_bench_1gpu.json; echo
bash tools/gpu_profile.sh 0 $TAG > gpurun_out/${TAG}_profile.log 2>&1; tail -3 gpurun_out/${TAG}_profile.log
for c in strip strip_fused C4 C3_gi C5; do bash tools/gpu_profile_config.sh $TAG $c > gpurun_out/${TAG}_profile_$c.log 2>&1; done
bash tools/gpu_multi_rehearsal.sh $TAG > gpurun_out/${TAG}_rehearsal.log 2>&1; tail -4 gpurun_out/${TAG}_rehearsal.log
python tools/gpu_oneshot.py 2>/dev/null > gpurun_out/${TAG}_oneshot.log; tail -2 gpurun_out/${TAG}_oneshot.log
python tools/setup_probe.py 768 2>/dev/null > gpurun_out/${TAG}_setup_probe.log; cat gpurun_out/${TAG}_setup_probe.log
