#!/bin/bash
# Round-3 closing run on the GPU box (one gpurun call): the whole device suite, the fuzzers on fresh seeds, the rocprofv3
# passes of the headline bench (tools/gpu_profile.sh -> profiles/r03_pmc_traffic.json), the default bench line, the
# multi-rank bench rehearsed on the one GPU (gloo, peer halos between processes), smoke().
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export OMP_NUM_THREADS=8
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gpu_tests.log
timeout 200 python tools/gpu_fuzz.py 500000 2500 > gpurun_out/fuzz_r03.log 2>&1; tail -1 gpurun_out/fuzz_r03.log
timeout 200 python tools/gpu_fuzz_fd.py 91000 2500 > gpurun_out/fuzz_fd_r03.log 2>&1; tail -1 gpurun_out/fuzz_fd_r03.log
timeout 200 python tools/gpu_fuzz_strips.py 71000 800 > gpurun_out/fuzz_strips_r03.log 2>&1; tail -1 gpurun_out/fuzz_strips_r03.log
timeout 200 python tools/gpu_fuzz_wavefront.py 33000 800 > gpurun_out/fuzz_wf_r03.log 2>&1; tail -1 gpurun_out/fuzz_wf_r03.log
unset OMP_NUM_THREADS
bash tools/gpu_profile.sh 0 r03 > gpurun_out/profile_r03.log 2>&1; tail -1 gpurun_out/profile_r03.log
cp gpurun_out/prof_r03/pmc_traffic.json profiles/r03_pmc_traffic.json
timeout 600 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; cat gpurun_out/bench_r03.json
for n in 2 4; do
  F3D_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) \
    bench.py --gpus $n --steps 8 --warmup 2 --no-cpu-baseline --extra-windows 0 > gpurun_out/bench_r03_rehearsal_$n.json 2> gpurun_out/bench_r03_rehearsal_$n.err
  echo "rehearsal $n ranks rc=$?"; tail -1 gpurun_out/bench_r03_rehearsal_$n.json | cut -c1-400
done
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
