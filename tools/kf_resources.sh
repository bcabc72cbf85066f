#!/bin/bash
# Development loop for ONE frame-kernel instantiation (cross-compiles for gfx950, no GPU needed, ~10 s):
#   tools/kf_resources.sh [-k 'k_frame<0, 6, 4, false>'] [-s out.s] [extra hipcc flags]
# prints VGPR / SGPR / scratch / occupancy and the static count of scratch loads / stores; -s keeps the assembly.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
KERNEL='k_frame<0, 6, 4, false>'
ASM=""
while getopts "k:s:" o; do case $o in k) KERNEL="$OPTARG";; s) ASM="$OPTARG";; esac; done
shift $((OPTIND - 1))
OUT=$(mktemp -d)
cat > "$OUT/one.hip" <<SRC
#include "f3d_frame.h"
namespace f3d { template __global__ void $KERNEL(const FrameParams); }
SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
    -I "$ROOT/forge3d_amd/csrc" -I "$ROOT/include" "$@" --cuda-device-only -S "$OUT/one.hip" -o "$OUT/one.s" -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} /VGPRs:/ && !/AGPRs/ && !/Spill/ {v=$(NF-1)} /TotalSGPRs:/ {sg=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)}
       /LDS Size/ {if (name ~ /k_frame|k_trace|k_wf/) printf "%s: vgpr %s sgpr %s scratch %s B/lane occ %s lds %s\n", name, v, sg, sc, oc, $(NF-1)}'
sed -n "/^_ZN3f3d\(7k_frame\|7k_trace\|12k_wf\)I/,/^\.Lfunc_end/p" "$OUT/one.s" > "$OUT/k.s"
echo "static: $(grep -c scratch_store "$OUT/k.s") scratch stores, $(grep -c scratch_load "$OUT/k.s") scratch loads, $(grep -c 'v_writelane' "$OUT/k.s") writelane, $(grep -c 'v_readlane' "$OUT/k.s") readlane, $(grep -cE '^\s+[vs]_|^\s+(ds|global|scratch|buffer)_' "$OUT/k.s") instructions"
[ -n "$ASM" ] && cp "$OUT/k.s" "$ASM"
rm -rf "$OUT"
