#!/bin/bash
# Single-GPU rehearsal of the multi-GPU job (run on the GPU box via gpurun): the compute-only strong-scaling bound of the
# strips (tools/strip_balance.py) and bench.py's N-rank path with every rank on the one GPU (collectives over gloo,
# halos over real IPC handles) -- the second is a test of the code path and of the JSON line, its rate means nothing.
#   tools/gpu_multi_rehearsal.sh <tag>
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LOG=gpurun_out/${TAG}_strip_balance.log
: > $LOG
echo "== fd 16 (the driver's default from 5 ranks on)" >> $LOG
python tools/strip_balance.py --worlds 8 --fd 16 --rounds 4 2>/dev/null >> $LOG
echo "== fd 0 (fused kernel: the default below 5 ranks)" >> $LOG
python tools/strip_balance.py --worlds 2,4 --fd 0 --rounds 3 2>/dev/null >> $LOG
cat $LOG
for N in 2 8; do
  F3D_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 \
      bench.py --gpus $N --steps 8 --warmup 2 --extra-windows 0 --no-cpu-baseline 2> gpurun_out/${TAG}_bench_${N}ranks_one_gpu.err | tail -1 > gpurun_out/${TAG}_bench_${N}ranks_one_gpu.json
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_${N}ranks_one_gpu.json"))
c = d["config"]
print("$N ranks on one GPU:", {k: c.get(k) for k in ("parallelism", "peer_halos", "rank_ms_per_step", "halo_wait_ms_per_frame", "halo_longest_wait_ms", "strip_row_bounds", "frames_in_flight")})
PY
done
