#!/bin/bash
# second half of the closing run: the device suite again (after a test fix), the rocprofv3 passes, the bench line; only
# small files go to gpurun_out (it is merged back up to 64 MiB)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gpu_tests.log
bash tools/gpu_profile.sh 0 r03 > gpurun_out/profile_r03.log 2>&1; tail -1 gpurun_out/profile_r03.log
mkdir -p gpurun_out/keep; cp gpurun_out/prof_r03/pmc_traffic.json gpurun_out/prof_r03/summary.txt gpurun_out/keep/ 2>/dev/null
cp gpurun_out/prof_r03/pmc_traffic.json profiles/r03_pmc_traffic.json
timeout 600 python bench.py > gpurun_out/keep/bench_r03.json 2> gpurun_out/keep/bench_r03.err; cat gpurun_out/keep/bench_r03.json | cut -c1-300
rm -rf gpurun_out/prof_r03/*/  # the rocpd databases stay on the box
ls -la gpurun_out/keep
