#!/bin/bash
# rocprofv3 passes for the multi-bounce PBR tracer at the adjudication gate (run on the GPU box via gpurun): $1 = tag
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_wf_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/wf_time.py 512 4096"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o wf -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM -d $OUT/pmc1 -o wf -- $CMD > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $OUT/pmc2 -o wf -- $CMD > $OUT/pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d $OUT/pmc3 -o wf -- $CMD > $OUT/pmc3.log 2>&1
cd $R
python tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
grep fpl $OUT/trace.log
grep -A12 "==== trace" $OUT/summary.txt
