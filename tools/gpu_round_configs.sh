#!/bin/bash
# The rocprofv3 passes of the other configurations' dominant kernels and of the strip form, the strip rehearsals and the N-rank
# set-up rehearsal, one gpurun call:  tools/gpu_round_configs.sh TAG     (summaries -> gpurun_out/keep/)
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep; K=gpurun_out/keep
for CFG in C3_gi C4 C5 strip strip_fused; do
  bash tools/gpu_profile_config.sh $TAG $CFG > gpurun_out/profile_${TAG}_$CFG.log 2>&1
  cp gpurun_out/prof_${TAG}_$CFG/summary.txt $K/${TAG}_${CFG}_rocprofv3_summary.txt
  rm -rf gpurun_out/prof_${TAG}_$CFG
  grep -E "k_wf_paths|k_frame|k_smoke\(|k_phase<6u>|k_trace|k_merge" $K/${TAG}_${CFG}_rocprofv3_summary.txt | grep -E "^void|^\(anon|^f3d" | head -4
done
{ echo "== C2 (1080p headline), strips cut from one row-cost map, 16 frames in flight from 4 strips on"; python tools/strip_balance.py --costmap --worlds 2,4,8 --fd 16 2>/dev/null; } > $K/${TAG}_strip_balance.log
{ echo "== C4 stand-in (4096^2, 600 000 triangles), strips cut from one row-cost map"; python tools/strip_balance.py --config c4 --costmap --worlds 2,4,8 --fd 16 2>/dev/null; } > $K/${TAG}_strip_balance_C4.log
grep speedup $K/${TAG}_strip_balance.log $K/${TAG}_strip_balance_C4.log | cut -c1-260
bash tools/gpu_rank_setup.sh $TAG 2 4 8 2>&1 | tail -3
cp gpurun_out/${TAG}_bench_*ranks_one_gpu.json $K/
