#!/bin/bash
# rocprofv3 passes for the frame kernel (run on the GPU box via gpurun).  $1 = variant, $2 = tag
# One --kernel-trace --stats pass + four separate --pmc passes (never combined with other trace domains);
# writes text summaries + pmc_traffic.json (with the kernel-source hash bench.py checks) under gpurun_out/prof_$TAG.
V=${1:-0}; TAG=${2:-r02}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --extra-windows 0 --no-terrain-filling --no-configs --variant $V"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o bench -- $BENCH > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o bench -- $BENCH > $OUT/pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc3 -o bench -- $BENCH > $OUT/pmc3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc4 -o bench -- $BENCH > $OUT/pmc4.log 2>&1
cd $R
python tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
python - <<PY
import glob, json, sqlite3, sys
sys.path.insert(0, "$R")
import bench
def avg(db, counter, like):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    row = cur.execute(f"select avg(value), count(*) from counters_collection where counter_name=? and {name_col} like ?", (counter, like)).fetchone()
    return row
out = {"kernel_source_hash": bench.kernel_source_hash(), "variant": int("$V"), "workload": "rainier-proxy 2048^2, 1920x1080, 8 spp/frame, 1 GPU",
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum / --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum, separate passes, bench.py --steps 8 --warmup 2"}
try:
    f = avg(glob.glob("$OUT/pmc3/*.db")[0], "FETCH_SIZE", "%k_frame<0, 6, 4u, false>%")
    w = avg(glob.glob("$OUT/pmc4/*.db")[0], "WRITE_SIZE", "%k_frame<0, 6, 4u, false>%")
    h = avg(glob.glob("$OUT/pmc3/*.db")[0], "TCC_HIT_sum", "%k_frame<0, 6, 4u, false>%")
    m = avg(glob.glob("$OUT/pmc4/*.db")[0], "TCC_MISS_sum", "%k_frame<0, 6, 4u, false>%")
    out.update(FETCH_SIZE_KB_per_dispatch=f[0], WRITE_SIZE_KB_per_dispatch=w[0], dispatches=f[1], gfx950_fetch_correction=2.0,
               hbm_bytes_per_launch=int((2.0 * f[0] + w[0]) * 1024), l2_hit_rate=h[0] / (h[0] + m[0]), sample_lanes=4,
               note="MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE count KB; gfx950 FETCH_SIZE under-reports wide reads, doubled (upper bound)")
except Exception as exc:
    out["error"] = str(exc)
try:  # the roof the kernel is actually under: vector-instruction issue (bench.py roofline_issue)
    K = "%k_frame<0, 6, 4u, false>%"
    p1, p2 = glob.glob("$OUT/pmc1/*.db")[0], glob.glob("$OUT/pmc2/*.db")[0]
    c = {name: avg(db, name, K)[0] for db, names in ((p1, ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES")),
                                                      (p2, ("SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"))) for name in names}
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
    # average duration of the same kernel in the kernel-trace pass (the clock the counters' cycles were counted at)
    kms = None
    try:
        kms = sqlite3.connect(glob.glob("$OUT/trace/*.db")[0]).cursor().execute("select avg(end - start) from kernels where name like ?", (K,)).fetchone()[0] / 1e6
    except Exception:
        kms = None
    out["issue"] = {"kernel_ms_profiled": kms, "counters_per_launch": c, "gpu_cycles_per_launch": cycles, "simds": 1024,
                    "valu_wave_instructions_per_launch": c["SQ_INSTS_VALU"],
                    "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
                    "wait_fraction_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                    "salu_per_valu": c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"],
                    "source": "rocprofv3 --pmc passes 1 and 2 of tools/gpu_profile.sh, averages over the launches of k_frame<0, 6, 4u, false>"}
except Exception as exc:
    out["issue_error"] = str(exc)
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
head -30 $OUT/summary.txt
