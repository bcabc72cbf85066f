#!/bin/bash
# L2 behaviour of the frame kernel under the tile -> XCD maps: default (rows dealt round-robin) vs contiguous bands (variant 3000)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in 0 3000; do
  OUT=$R/gpurun_out/l2_$v; mkdir -p $OUT
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/pmc -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --extra-windows 0 --no-terrain-filling --no-configs --variant $v > $OUT/log.txt 2>&1
  (cd $R; python tools/rocpd_summary.py $OUT 2>&1 | grep "k_frame" | awk -v n=$v '{print "variant", n, $0}')
  python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --extra-windows 0 --no-terrain-filling --no-configs --variant $v 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('variant $v: %.1f Msamples/s' % d['value'])"
done
