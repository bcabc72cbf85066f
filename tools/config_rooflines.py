#!/usr/bin/env python
"""Roofline rows of the dominant kernels of the non-headline configurations and of the strip form, from the rocprofv3
summaries under profiles/ (tools/gpu_profile_config.sh):   python tools/config_rooflines.py r04  ->  profiles/r04_config_rooflines.json

Per kernel: average duration (kernel trace), HBM-side traffic per launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, KB ->
bytes; the factor 2 is MI355X_MICROARCH.md's gfx950 correction, an upper bound), the bandwidth that is and its fraction of the
8 TB/s peak, and what actually bounds these kernels -- the share of the VALU issue roof (1 024 SIMDs, ~2.4 cycles per wave
instruction: DESIGN.md 6), lane utilisation and the share of wave cycles spent waiting.  bench.py attaches the rows to its
`configs` entries when the kernel sources are the ones that were profiled."""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
WHAT = {  # config -> (summary file, kernel name prefix in the summary, launches of it that make one "unit", unit)
    "C4_standin": ("C4", "void f3d::k_frame<0, 6, 4u, true>", "frame of 4096 x 4096 x 8 spp"),
    "C3_gi": ("C3_gi", "void (anonymous namespace)::k_wf_paths<true>" if tag < "r05" else "void (anonymous namespace)::k_wf_paths<true, true>", "launch of 1920 x 1080 x 32 paths"),
    "C5_march": ("C5", "(anonymous namespace)::k_smoke(" if tag < "r05" else "(anonymous namespace)::k_smoke_light(", "1080p frame of the smoke marcher" if tag < "r05" else "self-shadow marches of a 1080p smoke frame"),
    "C5_solver_jacobi": ("C5", "void (anonymous namespace)::k_sim<7u>" if tag < "r05" else "void (anonymous namespace)::k_phase<6u>", "Jacobi sweep of the 96 x 64 x 128 domain"),
    "strip_trace": ("strip", "void f3d::k_trace<6, 8u, false>", "batch of <= 16 frames of an eighth of the 1080p frame"),
    "strip_merge": ("strip", "f3d::k_merge(", "strip-frame"),
    "strip_fused": ("strip_fused", "void f3d::k_frame<0, 6, 8u, false>", "strip-frame (fused kernel, 8 lanes)"),
}
if tag >= "r05":  # the marcher is three launches from round 5 on (csrc/f3d_smoke.hip)
    WHAT["C5_march_collect"] = ("C5", "void (anonymous namespace)::k_smoke_rays<1u>", "ray walk that lists the smoke steps of a 1080p frame")
    WHAT["C5_march_shade"] = ("C5", "(anonymous namespace)::k_smoke_shade(", "shading of the listed steps of a 1080p frame")
out = {}
for key, (cfg, kernel, unit) in WHAT.items():
    path = ROOT / "profiles" / f"{tag}_{cfg}_rocprofv3_summary.txt"
    text = path.read_text().splitlines()
    src = re.search(r"kernel sources: (\w+)", text[0])
    lib_src = re.search(r"library sources: (\w+)", text[0])
    counters, avg_ns, calls = {}, None, None
    for line in text:
        if not line.startswith(kernel[:44]) and not line.startswith(kernel):
            continue
        rest = line[len(kernel[:44]):].split() if not line.startswith(kernel + " ") else None
        parts = line.split()
        m = re.match(r"^(.{44}) (\S+)\s+(\d+)\s+([\d.]+)$", line)
        if m and m.group(2).isupper() or (m and "_" in m.group(2)):
            counters[m.group(2)] = float(m.group(4))
            continue
        m = re.match(r"^(.{64})\s+(\d+)\s+(\d+)\s+([\d.]+)\s+(\d+)\s+(\d+)\s+([\d.]+)$", line)
        if m:
            calls, avg_ns = int(m.group(2)), float(m.group(4))
    if avg_ns is None:
        continue
    c = counters
    row = {"kernel": kernel.replace("void ", "").rstrip("("), "per": unit, "kernel_ms": avg_ns / 1e6, "calls_profiled": calls,
           "kernel_source_hash": src.group(1) if src else None, "library_source_digest": lib_src.group(1) if lib_src else None,
           "profile": f"profiles/{path.name}"}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        row.update(hbm_bytes_per_launch=int(traffic), roofline={"bound": "hbm", "achieved": traffic / (avg_ns * 1e-9) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                                  "frac": traffic / (avg_ns * 1e-9) / 1e9 / 8000.0, "traffic": int(traffic),
                                                                  "note": "measured traffic (PMC, 2 x FETCH_SIZE + WRITE_SIZE), not algorithmic bytes"})
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        row["valu_issue_fraction"] = c["SQ_INSTS_VALU"] / 1024.0 * 2.4 / cycles
        row["salu_per_valu"] = c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_VALU"]
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        row["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        row["wait_fraction_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        row["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    out[key] = row
(ROOT / "profiles" / f"{tag}_config_rooflines.json").write_text(json.dumps(out, indent=1))
for k, v in out.items():
    r = v.get("roofline", {})
    print("%-18s %9.3f ms  traffic %7.1f MB  %7.1f GB/s (%.3f of 8 TB/s)  VALU issue %.2f  lanes %.2f  waiting %.2f" % (
        k, v["kernel_ms"], v.get("hbm_bytes_per_launch", 0) / 1e6, r.get("achieved", 0.0), r.get("frac", 0.0), v.get("valu_issue_fraction", 0.0),
        v.get("lane_utilisation", 0.0), v.get("wait_fraction_of_wave_cycles", 0.0)))
