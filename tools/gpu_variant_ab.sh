#!/bin/bash
# Time locally built library variants (tools/build_variant.sh) on the headline workload:
#   tools/gpu_variant_ab.sh name[:kernel_variant] ...   e.g. base raw raw:8000000   (name "tree" = the in-tree library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; v=0; [[ "$spec" == *:* ]] && v=${spec##*:}
  lib=$PWD/build_ab/libf3dhip_$name.so; [ "$name" = tree ] && lib=$PWD/forge3d_amd/libf3dhip.so
  F3D_HIP_LIBRARY=$lib python bench.py --steps ${STEPS:-16} --warmup 4 --variant $v --no-cpu-baseline --extra-windows 2 --no-terrain-filling --no-configs 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-10s variant %-8s %.1f Msamples/s  windows %s  rgb %s' % ('$name', '$v', d['value'], d.get('windows_ms_per_step'), d['config']['image_mean_rgb']))" | tee -a gpurun_out/variant_ab.log
done
