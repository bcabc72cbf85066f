#!/bin/bash
# Cross-compile f3d_kernels.hip for gfx950 (no GPU needed) and print the register / scratch budget of every kernel:
#   tools/kernel_resources.sh [extra hipcc flags, e.g. -DF3D_NO_SHARE]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize "$@" -c "$ROOT/forge3d_amd/csrc/f3d_kernels.hip" \
    -o "$OUT/k.o" -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$NF} /remark:/ && /Name:/ {name=$(NF-1)}
       /VGPRs:/ && !/AGPRs/ && !/Spill/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /TotalSGPRs:/ {sg=$(NF-1)}
       /ScratchSize/ {sc=$(NF-1)} /VGPR Spill/ {sp=$(NF-1)} /SGPR Spill/ {ssp=$(NF-1)} /Occupancy/ {oc=$(NF-1)}
       /LDS Size/ {printf "%-58s vgpr %3s agpr %3s sgpr %3s scratch %4s B/lane vspill %3s sspill %3s occ %s lds %s\n", name, v, a, sg, sc, sp, ssp, oc, $(NF-1)}' |
  sed 's/_ZN3f3d//; s/EvNS_11FrameParamsE//'
rm -rf "$OUT"
