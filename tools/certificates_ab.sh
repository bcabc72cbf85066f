#!/bin/bash
# First thing to run for this branch: build the four variants HERE (no box time), then on the GPU box the parity tests of
# the certificates and one timing of each variant.
#   tools/certificates_ab.sh build        (in the build container)
#   gpurun --timeout 600 -- 'bash tools/certificates_ab.sh run'
set -e
cd "$(dirname "$0")/.."
case "${1:-build}" in
build)
    bash tools/build_variant.sh none -DF3D_NO_PRIMARY_START -DF3D_NO_SUN_CLEAR -DF3D_NO_IBL_STOP &
    bash tools/build_variant.sh primary -DF3D_NO_SUN_CLEAR -DF3D_NO_IBL_STOP &
    bash tools/build_variant.sh primary_sun -DF3D_NO_IBL_STOP &
    bash tools/build_variant.sh all &
    wait
    ;;
run)
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
    timeout 200 python -m pytest tests/test_primary_start.py tests/test_halo_reach.py -m gpu -q 2>&1 | tail -2
    bash tools/gpu_variant_ab.sh none primary primary_sun all none all
    timeout 300 python tools/gpu_fuzz.py 120000 3000 2>&1 | tail -1   # the shipped library (all certificates) against the oracle
    ;;
esac
