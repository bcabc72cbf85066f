#!/bin/bash
# Development loop for the PBR tracer's kernels (cross-compiles f3d_wavefront.hip for gfx950, no GPU needed):
#   tools/wf_resources.sh [-s out.s] [extra hipcc flags]    -> VGPR / SGPR / scratch / occupancy of k_wf_paths<false|true>
# -s keeps the assembly of k_wf_paths<true> (the instantiation with the heightfield primitive: BASELINE configs[2]).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
ASM=""
while getopts "s:" o; do case $o in s) ASM="$OPTARG";; esac; done
shift $((OPTIND - 1))
OUT=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize "$@" --cuda-device-only -S \
    "$ROOT/forge3d_amd/csrc/f3d_wavefront.hip" -o "$OUT/wf.s" -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} /VGPRs:/ && !/AGPRs/ && !/Spill/ {v=$(NF-1)} /TotalSGPRs:/ {sg=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)}
       /LDS Size/ {if (name ~ /k_wf_paths/) printf "%s: vgpr %s sgpr %s scratch %s B/lane occ %s lds %s\n", name, v, sg, sc, oc, $(NF-1)}'
sed -n "/^_ZN12_GLOBAL__N_110k_wf_pathsILb1ELb1EEEvNS_8WfParamsE:/,/^\.Lfunc_end/p" "$OUT/wf.s" > "$OUT/k.s"
echo "k_wf_paths<true> static: $(grep -c scratch_store "$OUT/k.s") scratch stores, $(grep -c scratch_load "$OUT/k.s") scratch loads, $(grep -c 'v_writelane' "$OUT/k.s") writelane, $(grep -c 'v_readlane' "$OUT/k.s") readlane, $(grep -cE '^\s+[vs]_|^\s+(ds|global|scratch|buffer)_' "$OUT/k.s") instructions"
[ -n "$ASM" ] && cp "$OUT/k.s" "$ASM"
rm -rf "$OUT"
