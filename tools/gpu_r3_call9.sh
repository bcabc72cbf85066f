#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
F3D_FUZZ_WAVEFRONT=1 timeout 300 python tools/gpu_fuzz_fd.py 510000 400 2>&1 | tail -2
for q in 16 32; do for n in 2 4; do
F3D_WAVEFRONT=1 F3D_WF_FRAMES=$n F3D_WF_QUORUM=$q timeout 120 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --extra-windows 1 --no-terrain-filling 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('wavefront frames $n quorum $q: %.1f Msamples/s  windows %s fd %s rgb %s' % (d['value'], d.get('windows_ms_per_step'), d['config']['frames_in_flight'], d['config']['image_mean_rgb']))"
done; done
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/wf2
F3D_WAVEFRONT=1 F3D_WF_FRAMES=2 timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wf2/trace -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --extra-windows 0 --no-terrain-filling > $R/gpurun_out/wf2/log.txt 2>&1
cd $R; python tools/rocpd_summary.py gpurun_out/wf2 2>&1 | head -9
