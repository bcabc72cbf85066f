#!/bin/bash
# A/B of library variants with their dynamic instruction mix: tools/gpu_variant_pmc.sh name...
# per variant: bench timing (3 windows), then one --pmc pass (SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
STEPS=16 bash tools/gpu_variant_ab.sh "$@"
for name in "$@"; do
  name=${name%%:*}
  OUT=$R/gpurun_out/pmc_salu_$name; mkdir -p $OUT
  (cd /tmp; export TMPDIR=/tmp; F3D_HIP_LIBRARY=$R/build_ab/libf3dhip_$name.so timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/p -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --extra-windows 0 --no-terrain-filling --no-configs > $OUT/log.txt 2>&1)
  echo "== $name"; python tools/rocpd_summary.py $OUT 2>&1 | grep -i "k_frame" | head -8
done
