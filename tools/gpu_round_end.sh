#!/bin/bash
# The closing run of a change to the kernel sources, one gpurun call:  tools/gpu_round_end.sh TAG [quick|full]
#   quick (default): the device suite, a slice of each fuzz, the rocprofv3 passes of the headline bench
#                    (tools/gpu_profile.sh -> profiles/TAG_pmc_traffic.json, whose source hash bench.py checks), the default bench line
#   full:            + the fuzzers on thousands of fresh seeds, the 2- and 4-rank bench rehearsed on the one GPU
#                    (gloo, peer halos between processes) and smoke()
# Small results go to gpurun_out/keep/ (gpurun merges gpurun_out back only while it stays under 64 MiB).
TAG=${1:-r03}; MODE=${2:-quick}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep; K=gpurun_out/keep
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $K/gpu_tests_$TAG.log; tail -2 $K/gpu_tests_$TAG.log
if [ $MODE = full ]; then N1=2500; N2=2500; N3=800; N4=800; else N1=40; N2=200; N3=40; N4=40; fi
export OMP_NUM_THREADS=8
timeout 300 python tools/gpu_fuzz.py 500000 $N1 2>&1 | tail -1 | tee $K/fuzz_$TAG.log
timeout 300 python tools/gpu_fuzz_fd.py 91000 $N2 2>&1 | tail -1 | tee -a $K/fuzz_$TAG.log
timeout 300 python tools/gpu_fuzz_strips.py 71000 $N3 2>&1 | tail -1 | tee -a $K/fuzz_$TAG.log
timeout 300 python tools/gpu_fuzz_wavefront.py 33000 $N4 2>&1 | tail -1 | tee -a $K/fuzz_$TAG.log
unset OMP_NUM_THREADS
bash tools/gpu_profile.sh 0 $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -1 gpurun_out/profile_$TAG.log | cut -c1-400
cp gpurun_out/prof_$TAG/pmc_traffic.json gpurun_out/prof_$TAG/summary.txt $K/ 2>/dev/null
cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout 600 python bench.py > $K/bench_$TAG.json 2> $K/bench_$TAG.err; cut -c1-300 $K/bench_$TAG.json
rm -rf gpurun_out/prof_$TAG/*/
if [ $MODE = full ]; then
  for n in 2 4; do
    F3D_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) \
      bench.py --gpus $n --steps 8 --warmup 2 --no-cpu-baseline --extra-windows 0 > $K/bench_${TAG}_rehearsal_$n.json 2> $K/bench_${TAG}_rehearsal_$n.err
    echo "rehearsal $n ranks rc=$?"; tail -1 $K/bench_${TAG}_rehearsal_$n.json | cut -c1-400
  done
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
