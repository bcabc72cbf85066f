#!/bin/bash
# PMC passes of the wavefront trace kernels on the headline workload (tools/gpu_profile.sh does the fused kernel)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_wf; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export F3D_WAVEFRONT=1 F3D_WF_FRAMES=${WF_FRAMES:-2}
BENCH="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --extra-windows 0 --no-terrain-filling"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o bench -- $BENCH > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o bench -- $BENCH > $OUT/pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum WRITE_SIZE -d $OUT/pmc3 -o bench -- $BENCH > $OUT/pmc3.log 2>&1
cd $R; python tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1; grep "k_wf\|k_merge\|k_fix" $OUT/summary.txt
