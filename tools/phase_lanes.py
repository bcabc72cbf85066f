#!/usr/bin/env python
"""Per-phase lane utilisation of k_frame from the PMC passes of tools/gpu_phase_lanes.sh (the shipped library and the
three timing builds).  A phase's counters are differences: IBL = shipped - (IBL rays not traced), sun = shipped - (sun
rays not traced), primary + shading + head/tail = the build that traces neither.  Lane utilisation of a set of
instructions = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) (active lanes per issued vector instruction)."""
import glob
import json
import os
import sqlite3
import sys

COUNTERS = ("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE")


def read(root, name):
    db = glob.glob(os.path.join(root, name, "**", "*.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    for c in COUNTERS:
        row = cur.execute(f"select avg(value), count(*) from counters_collection where counter_name=? and {name_col} like ?",
                          (c, "%k_frame<0, 6, 4u, false>%")).fetchone()
        out[c] = float(row[0] or 0.0)
        out["dispatches"] = int(row[1])
    try:
        row = cur.execute("select avg(end-start), min(end-start), count(*) from kernels where name like ?", ("%k_frame<0, 6, 4u, false>%",)).fetchone()
        out["kernel_ms_avg_under_pmc"] = row[0] / 1e6
    except Exception:  # noqa: BLE001
        pass
    return out


def lanes(c):
    return c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if c["SQ_ACTIVE_INST_VALU"] else None


def main():
    root = sys.argv[1]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    raw = {n: read(root, n) for n in ("tree", "noibl", "nosun", "neither")}

    def diff(a, b):
        return {k: raw[a][k] - raw[b][k] for k in COUNTERS}

    phases = {"whole kernel": raw["tree"], "IBL rays (shipped - not traced)": diff("tree", "noibl"), "sun rays (shipped - not traced)": diff("tree", "nosun"),
              "primary rays + shading + head record + tail (neither traced)": raw["neither"]}
    table = {}
    for k, c in phases.items():
        table[k] = {"valu_wave_instructions": c["SQ_INSTS_VALU"], "salu_wave_instructions": c["SQ_INSTS_SALU"], "lane_utilisation": lanes(c),
                    "wave_cycles": c["SQ_WAVE_CYCLES"], "waiting_fraction": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c["SQ_WAVE_CYCLES"] else None,
                    "share_of_valu_instructions": c["SQ_INSTS_VALU"] / raw["tree"]["SQ_INSTS_VALU"]}
    print(json.dumps({"kernel_source_hash": bench.kernel_source_hash(), "kernel": "k_frame<0, 6, 4u, false>",
                      "workload": "bench.py --steps 8 --warmup 2 (headline: rainier-proxy 2048^2, 1920x1080, 8 spp/frame)",
                      "method": __doc__.strip(), "phases": table, "raw": raw}, indent=1))


if __name__ == "__main__":
    main()
