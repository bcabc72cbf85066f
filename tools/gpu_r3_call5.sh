#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/variant_ab.log gpurun_out/gbuffer_cost.log
timeout 300 python -m pytest tests/test_primary_start.py -m gpu -q 2>&1 | tail -2
bash tools/gpu_gbuffer_cost.sh all theta4 theta6
bash tools/gpu_variant_ab.sh primary_sun all theta6 theta4 all
OMP_NUM_THREADS=8 timeout 240 python tools/gpu_fuzz.py 140000 1200 2>&1 | tail -2
