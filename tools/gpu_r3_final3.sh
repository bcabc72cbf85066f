#!/bin/bash
# closing run, last part: the rocprofv3 passes of the headline bench (tools/gpu_profile.sh) and the default bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep
bash tools/gpu_profile.sh 0 r03 > gpurun_out/profile_r03.log 2>&1; tail -1 gpurun_out/profile_r03.log
cp gpurun_out/prof_r03/pmc_traffic.json gpurun_out/prof_r03/summary.txt gpurun_out/keep/ 2>/dev/null
cp gpurun_out/prof_r03/pmc_traffic.json profiles/r03_pmc_traffic.json
timeout 600 python bench.py > gpurun_out/keep/bench_r03.json 2> gpurun_out/keep/bench_r03.err; cat gpurun_out/keep/bench_r03.json | cut -c1-300
rm -rf gpurun_out/prof_r03/*/
