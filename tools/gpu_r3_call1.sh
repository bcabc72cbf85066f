#!/bin/bash
# Round 3, first GPU call: device verification of the certificate branch, the certificate / register-budget / numerics-tier
# A/B on the headline workload, the VALU counter calibration, and a fuzz of the shipped library against the oracle.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/variant_ab.log
timeout 300 python -m pytest tests/test_primary_start.py tests/test_halo_reach.py -m gpu -q 2>&1 | tail -3 | tee gpurun_out/r3_call1_tests.log
bash tools/gpu_variant_ab.sh none primary primary_sun all none all all:4105 all:4107 all:8000 fast fast:4105
bash tools/gpu_valu_calib.sh > gpurun_out/valu_calib.log 2>&1; head -20 gpurun_out/valu_calib.log
OMP_NUM_THREADS=8 timeout 240 python tools/gpu_fuzz.py 130000 1500 2>&1 | tail -2 | tee gpurun_out/r3_call1_fuzz.log
