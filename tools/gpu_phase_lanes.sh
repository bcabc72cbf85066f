#!/bin/bash
# Where the idle lanes of k_frame are, phase by phase (round-4 verdict item 3): one PMC pass of the headline bench command
# per library -- the shipped one and the three timing builds (IBL rays not traced / sun rays not traced / neither; wrong
# images on purpose, tools/build_variant.sh <name> -DF3D_TIMING_NO_IBL ...) -- and the differences between them.
#   tools/gpu_phase_lanes.sh TAG    ->  gpurun_out/phase_lanes_TAG/{<lib>/...db, phase_lanes.json}
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/phase_lanes_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for name in tree noibl nosun neither; do
  lib=$R/build_ab/libf3dhip_$name.so; [ "$name" = tree ] && lib=$R/forge3d_amd/libf3dhip.so
  F3D_HIP_LIBRARY=$lib timeout 240 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE \
      -d $OUT/$name -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --extra-windows 0 --no-terrain-filling --no-configs > $OUT/$name.log 2>&1
done
cd $R
python tools/phase_lanes.py $OUT > $OUT/phase_lanes.json
cat $OUT/phase_lanes.json
