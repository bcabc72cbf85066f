#!/bin/bash
# Mesh-walk A/B on the GPU box (BASELINE.json configs[3] stand-in: 2048^2 DEM + 600 000 triangles, 4096^2, 8 spp per frame):
#   tools/gpu_mesh_ab.sh [--headline] [--tests] [--pmc] name[:kernel_variant] ...
# name = a library built with tools/build_variant.sh (build_ab/libf3dhip_<name>.so), or "tree" for the in-tree library;
# kernel_variant as f3d_session_opts (104 / 105: waves per SIMD, 4000: image-order dispatch, ...).  Every row prints a digest of
# the image.  --headline adds the 1080p headline workload per name, --tests the mesh / BVH device tests on the in-tree library,
# --pmc three counter passes with and without the mesh (lane utilisation and waits of the walk).  A -DF3D_MESH_STATS build
# prints the wave-level statistics of the walk (C4_MESH_STATS=1).  Round-3 results: profiles/r03_mesh_walk_ab.log.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/keep; L=$R/gpurun_out/keep/mesh_ab.log; : > $L
HEAD=0; TESTS=0; PMC=0; NAMES=()
for a in "$@"; do case $a in --headline) HEAD=1;; --tests) TESTS=1;; --pmc) PMC=1;; *) NAMES+=("$a");; esac; done
libof() { [ "$1" = tree ] && echo $R/forge3d_amd/libf3dhip.so || echo $R/build_ab/libf3dhip_$1.so; }
for spec in "${NAMES[@]}"; do
  n=${spec%%:*}; v=0; [[ "$spec" == *:* ]] && v=${spec##*:}
  C4_VARIANT=$v F3D_HIP_LIBRARY=$(libof $n) timeout 200 python tools/experiments/c4_window.py 4 2>&1 | tail -2 | sed "s/^/$n /" | tee -a $L
done
if [ $HEAD = 1 ]; then STEPS=16 bash tools/gpu_variant_ab.sh "${NAMES[@]}" 2>&1 | grep "Msamples" | tee -a $L; fi
if [ $TESTS = 1 ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mesh or config4 or bvh" 2>&1 | tail -3 | tee -a $L; fi
if [ $PMC = 1 ]; then
  for m in 0 1; do
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
      OUT=$R/gpurun_out/pmc_c4; rm -rf $OUT; mkdir -p $OUT
      (cd /tmp; export TMPDIR=/tmp; C4_NO_MESH=$m timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p -o bench -- python $R/tools/experiments/c4_window.py 2 > $OUT/log.txt 2>&1)
      echo "== no_mesh=$m  $set" | tee -a $L; python tools/rocpd_summary.py $OUT 2>&1 | grep -i "k_frame" | head -8 | tee -a $L
    done
  done
  rm -rf $R/gpurun_out/pmc_c4
fi
