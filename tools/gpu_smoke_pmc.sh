#!/bin/bash
# PMC passes of the smoke marcher on the configs[4] stand-in (tools/c5_time.py, 8 frames): tools/gpu_smoke_pmc.sh TAG [env...]
TAG=${1:-smoke}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/smoke_pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
[ -f $R/gpurun_out/counters_list.txt ] || (rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1 || true)
env "$@" timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $OUT/pmc1 -o c5 -- python $R/tools/c5_time.py 8 > $OUT/pmc1.log 2>&1
env "$@" timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o c5 -- python $R/tools/c5_time.py 8 > $OUT/pmc2.log 2>&1
env "$@" timeout 200 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc3 -o c5 -- python $R/tools/c5_time.py 8 > $OUT/pmc3.log 2>&1
cd $R; python tools/rocpd_summary.py $OUT 2>&1 | grep -E "====|k_smoke_light|k_smoke_rays|k_smoke_shade" 
