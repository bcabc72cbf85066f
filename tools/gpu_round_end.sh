#!/bin/bash
# One gpurun call at the end of a change to the kernel sources: GPU tests, the frames-in-flight fuzz, the rocprofv3
# passes (tools/gpu_profile.sh), then bench.py with the fresh PMC traffic in place.  $1 = profile tag (r02)
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gpu_tests.log
timeout 300 python tools/gpu_fuzz_fd.py 3800 1200 > gpurun_out/fuzz_fd.log 2>&1; tail -2 gpurun_out/fuzz_fd.log
bash tools/gpu_profile.sh 0 $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -1 gpurun_out/profile_$TAG.log
cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
