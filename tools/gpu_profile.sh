#!/bin/bash
# rocprofv3 passes for the frame kernel (run on the GPU box via gpurun).  $1 = variant, $2 = tag
V=${1:-0}; TAG=${2:-r1}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --variant $V"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o bench -- $BENCH > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o bench -- $BENCH > $OUT/pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc3 -o bench -- $BENCH > $OUT/pmc3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc4 -o bench -- $BENCH > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -40
python - <<PY
import csv, glob, collections
for d in ("trace","pmc1","pmc2","pmc3","pmc4"):
    for f in glob.glob("$OUT/%s/**/*kernel_stats.csv" % d, recursive=True):
        print("==", f)
        print(open(f).read()[:1500])
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "k_frame" in row.get("Kernel_Name", ""):
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        print("==", d, {k: (v[0] / max(v[1], 1), v[1]) for k, v in acc.items()})
PY
