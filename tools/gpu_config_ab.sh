#!/bin/bash
# A/B of prebuilt library variants (tools/build_variant.sh) on one configuration's workload (tools/config_workload.py):
#   tools/gpu_config_ab.sh <config> <variant> [<variant> ...]        e.g.  tools/gpu_config_ab.sh C3_gi wfbase wfdefer wfbase wfdefer
# Output appended to gpurun_out/config_ab.log.
cd "${GRAFT_REPO_ROOT:-.}"
CONFIG=$1; shift
mkdir -p gpurun_out
for V in "$@"; do
  printf "%-8s %-14s " "$CONFIG" "$V" | tee -a gpurun_out/config_ab.log
  F3D_HIP_LIBRARY=build_ab/libf3dhip_$V.so timeout 300 python tools/config_workload.py $CONFIG 2>&1 | grep -E "ms per frame|gi loop ms" | tr '\n' ' ' | tee -a gpurun_out/config_ab.log
  echo | tee -a gpurun_out/config_ab.log
done
