#!/bin/bash
# the frame kernels compiled per scene kind (terrain only / with mesh): headline A/B, then counters of the C4 stand-in with
# and without its mesh (lane utilisation and waits of the mesh walk)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep; L=$R/gpurun_out/keep/mesh_ab6.log; : > $L
STEPS=16 bash tools/gpu_variant_ab.sh meshfirst tmpl meshfirst tmpl 2>&1 | tee -a $L
timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -1 | tee -a $L
for m in 0 1; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    OUT=$R/gpurun_out/pmc_c4; rm -rf $OUT; mkdir -p $OUT
    (cd /tmp; export TMPDIR=/tmp; C4_NO_MESH=$m timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p -o bench -- python $R/tools/experiments/c4_window.py 2 > $OUT/log.txt 2>&1)
    echo "== no_mesh=$m  $set" | tee -a $L; python tools/rocpd_summary.py $OUT 2>&1 | grep -i "k_frame" | head -4 | tee -a $L
    tail -2 $OUT/log.txt | cut -c1-300 >> $L
  done
done
rm -rf $R/gpurun_out/pmc_c4
