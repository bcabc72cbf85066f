#!/bin/bash
# Round 6, the two bounded attempts on k_frame, one gpurun call:  tools/gpu_r6_kframe_ab.sh
#   branchless : -DF3D_BRANCHLESS_STEP   the march step with no data-dependent branch but the rare corner ties
#   sunhz      : -DF3D_SUN_HORIZON       the DEM-block far horizons (F3D_IBL_HORIZON=1, built once per DEM) stop the sun rays too
# Variants are built HERE by tools/build_variant.sh; every row is the default bench window (16 frames, 3 windows), image mean printed.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/keep; LOG=gpurun_out/keep/r06_kframe_ab.log; : > $LOG
run() {  # name lib env...
  name=$1; lib=$2; shift 2
  env "$@" F3D_HIP_LIBRARY=$lib python bench.py --steps 16 --warmup 4 --no-cpu-baseline --extra-windows 2 --no-terrain-filling --no-configs 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-34s %.1f Msamples/s  windows %s  setup %.2f ms  rgb %s' % ('$name', d['value'], d.get('windows_ms_per_step'), d['config']['setup_ms_once_per_render'], d['config']['image_mean_rgb']))" | tee -a $LOG
}
T=$PWD/forge3d_amd/libf3dhip.so
for rep in 1 2; do
  run "tree" $T F3D_X=0
  run "branchless" $PWD/build_ab/libf3dhip_branchless.so F3D_X=0
  run "tree + IBL block horizons" $T F3D_IBL_HORIZON=1
  run "sunhz + IBL block horizons" $PWD/build_ab/libf3dhip_sunhz.so F3D_IBL_HORIZON=1
  run "both + IBL block horizons" $PWD/build_ab/libf3dhip_both.so F3D_IBL_HORIZON=1
done
