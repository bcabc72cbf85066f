#!/bin/bash
# The strip rehearsals of one round on the one GPU of the box, one gpurun call:  tools/gpu_strip_rehearsal.sh TAG
#   compute-only strong-scaling bound of the strips (cost map from equal shares, one re-cut), C2 and the C4 stand-in;
#   bench.py's N-rank path with every rank on the one GPU (gloo collectives, real IPC halos): set-up trace, identity fields, re-cut
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep; K=gpurun_out/keep
{ echo "== C2 (1080p headline), strips cut from a row-cost map measured in equal shares, 16 frames in flight from 4 strips on"; python tools/strip_balance.py --costmap --worlds 2,4,8 --fd 16 2>/dev/null; } > $K/${TAG}_strip_balance.log
{ echo "== C4 stand-in (4096^2, 600 000 triangles), strips cut from a row-cost map measured in equal shares"; python tools/strip_balance.py --config c4 --costmap --worlds 2,4,8 --fd 16 2>/dev/null; } > $K/${TAG}_strip_balance_C4.log
grep -h "speedup\|cost_map" $K/${TAG}_strip_balance.log $K/${TAG}_strip_balance_C4.log | cut -c1-300
bash tools/gpu_rank_setup.sh $TAG 2 4 8 2>&1 | tail -3 | cut -c1-900
cp gpurun_out/${TAG}_bench_*ranks_one_gpu.json gpurun_out/${TAG}_bench_*ranks_one_gpu.err $K/
