#!/bin/bash
# A/B the frame-kernel variants on the headline workload (run on the GPU box via gpurun).
mkdir -p gpurun_out
for v in "$@"; do
  timeout 200 python bench.py --steps 8 --warmup 2 --variant $v --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: print('variant $v:', line.strip()[:300]); continue
    print('variant $v: %.1f Msamples/s  %.3f ms/step' % (d['value'], d['ms_per_step']))
" | tee -a gpurun_out/ab.log
done
