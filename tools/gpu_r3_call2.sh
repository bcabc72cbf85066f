#!/bin/bash
# Round 3, second GPU call: rocprofv3 summary + PMC passes of the frame kernel WITH certificates, proper register-budget /
# sample-lane variants, FIFO / sharing / numerics-split A/B builds, and the per-wave issue-rate microbenchmark.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/variant_ab.log
bash tools/gpu_variant_ab.sh all all:105 all:107 all:8000000 all:2000000 fifo3 share8 share32 contract divsqrt all
./build_ab/issue_calib 20000 > gpurun_out/issue_calib.log 2>&1; cat gpurun_out/issue_calib.log
bash tools/gpu_profile.sh 0 r03a > gpurun_out/prof_r03a.log 2>&1; tail -40 gpurun_out/prof_r03a.log
