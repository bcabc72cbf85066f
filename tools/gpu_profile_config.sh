#!/bin/bash
# rocprofv3 passes for one configuration's workload (tools/config_workload.py), run on the GPU box via gpurun:
#   tools/gpu_profile_config.sh <tag> <config> [kernel name pattern for the PMC summary]
# One --kernel-trace --stats pass + separate --pmc passes (never combined with other trace domains); the text summary lands
# in gpurun_out/prof_<tag>_<config>/summary.txt (copy to profiles/<tag>_<config>_rocprofv3_summary.txt).
TAG=${1:-r04}; CFG=${2:-C4}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_${TAG}_${CFG}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/config_workload.py $CFG"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o w -- $CMD > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o w -- $CMD > $OUT/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $OUT/pmc2 -o w -- $CMD > $OUT/pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc3 -o w -- $CMD > $OUT/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc4 -o w -- $CMD > $OUT/pmc4.log 2>&1
cd $R
{ echo "# workload: python tools/config_workload.py $CFG   kernel sources: $(python -c 'import bench; print(bench.kernel_source_hash())')   library sources: $(python -c 'from forge3d_amd import _native; print(_native.source_digest()[:16])')"; tail -3 $OUT/trace.log; python tools/rocpd_summary.py $OUT; } > $OUT/summary.txt 2>&1
python tools/trace_gaps.py $OUT/trace >> $OUT/summary.txt 2>&1
rm -rf $OUT/pmc*/*.db   # (keep the trace db for the timeline, drop the counter dbs: the summary has their averages)
grep -A14 "==== trace" $OUT/summary.txt | head -24
