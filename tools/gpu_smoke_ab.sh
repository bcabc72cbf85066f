#!/bin/bash
# Kernel-trace A/B of the smoke marcher on the configs[4] stand-in: tools/gpu_smoke_ab.sh NAME...  (NAME = build_ab variant, "shipped" = the tree's library)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for NAME in "$@"; do
  OUT=$R/gpurun_out/smoke_ab_$NAME; rm -rf $OUT; mkdir -p $OUT
  if [ $NAME = shipped ]; then unset F3D_HIP_LIBRARY; else export F3D_HIP_LIBRARY=$R/build_ab/libf3dhip_$NAME.so; fi
  F3D_C5_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o c5 -- python $R/tools/c5_time.py 60 > $OUT/run.log 2>&1
  echo "== $NAME: $(grep 'C5 ms' $OUT/run.log)"
  python $R/tools/rocpd_summary.py $OUT 2>&1 | grep -E "k_smoke|k_composite" | head -4
  find $OUT -name "*.db" -delete
done
