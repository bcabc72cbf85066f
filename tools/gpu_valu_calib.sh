#!/bin/bash
# Calibrate the SQ VALU counters on gfx950 against a pure v_fma_f32 kernel (tools/experiments/valu_calib.hip):
#   hipcc --offload-arch=gfx950 -O3 -w tools/experiments/valu_calib.hip -o build_ab/valu_calib     (build container)
#   gpurun -- 'bash tools/gpu_valu_calib.sh'
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/valu_calib; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
$R/build_ab/valu_calib 20000 > $OUT/timing.txt 2>&1
for cfg in "1 0" "4 0" "8 0" "8 32" "8 16"; do
  set -- $cfg
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY \
      -d $OUT/w$1_m$2 -o c -- $R/build_ab/valu_calib 20000 $1 $2 > $OUT/w$1_m$2.log 2>&1
done
cd $R; python tools/rocpd_summary.py $OUT > $OUT/counters.txt 2>&1
cat $OUT/timing.txt; grep -v "^$" $OUT/counters.txt | head -80
