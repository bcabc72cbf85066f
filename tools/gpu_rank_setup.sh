#!/bin/bash
# bench.py's N-rank path with every rank on the ONE GPU of the box (gloo collectives, halos over real IPC handles): a test
# of the code path and of the set-up time (config.setup_ms_once_per_render); the rates mean nothing.   tools/gpu_rank_setup.sh TAG N...
TAG=${1:-r05}; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for N in "$@"; do
  F3D_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 \
      bench.py --gpus $N --steps 8 --warmup 2 --extra-windows 0 --no-cpu-baseline 2> gpurun_out/${TAG}_bench_${N}ranks_one_gpu.err | tail -1 > gpurun_out/${TAG}_bench_${N}ranks_one_gpu.json
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_${N}ranks_one_gpu.json"))
c = d["config"]
print("$N ranks on one GPU:", {k: c.get(k) for k in ("setup_ms_once_per_render", "setup_trace_ms_rank0", "peer_halos", "rank_ms_per_step", "halo_wait_ms_per_frame", "strip_row_bounds", "frames_in_flight")})
PY
done
