#!/bin/bash
# Which source lines own the scratch stores / loads of k_wf_paths<true, true>?  tools/wf_spill_lines.sh [-- hipcc flags]
# (-gline-tables-only: the assembly carries .loc directives; counts are static instructions, not executions)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; [ "$1" = "--" ] && shift
OUT=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -gline-tables-only "$@" --cuda-device-only -S "$ROOT/forge3d_amd/csrc/f3d_wavefront.hip" -o "$OUT/wf.s" 2>/dev/null
python3 - "$OUT/wf.s" <<'PY'
import re, collections, sys
files = {}
text = open(sys.argv[1]).read().split("\n")
for l in text:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
inside = False; cur = None; st = collections.Counter(); ld = collections.Counter()
for l in text:
    if l.startswith("_ZN12_GLOBAL__N_110k_wf_pathsILb1ELb1EEEvNS_8WfParamsE:"): inside = True
    if inside and l.startswith(".Lfunc_end"): break
    if not inside: continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = "%s:%s" % (files.get(int(m.group(1)), m.group(1)), m.group(2))
    if 'scratch_store' in l: st[cur] += 1
    if 'scratch_load' in l: ld[cur] += 1
print("scratch stores:", sum(st.values()), dict(st.most_common(14)))
print("scratch loads :", sum(ld.values()), dict(ld.most_common(14)))
PY
rm -rf "$OUT"
