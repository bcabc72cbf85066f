#!/bin/bash
# Cross-compile a variant of libf3dhip.so HERE (no GPU needed) for an A/B on the GPU box without spending box time on
# the compiler: tools/build_variant.sh <name> [extra hipcc flags] -> build_ab/libf3dhip_<name>.so (git-ignored, travels
# with gpurun); run with F3D_HIP_LIBRARY=build_ab/libf3dhip_<name>.so python bench.py ...
# Flags as forge3d_amd/_native.py: f3d_kernels.hip is compiled apart with KERNEL_FLAGS, then linked with the rest.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build_ab
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize"
KERNEL_FLAGS=${KERNEL_FLAGS-}
/opt/rocm/bin/hipcc $COMMON $KERNEL_FLAGS "$@" -c forge3d_amd/csrc/f3d_kernels.hip -o build_ab/f3d_kernels_$NAME.o 2> build_ab/$NAME.err
/opt/rocm/bin/hipcc $COMMON -shared "$@" build_ab/f3d_kernels_$NAME.o \
    forge3d_amd/csrc/f3d_host.hip forge3d_amd/csrc/f3d_denoise.hip forge3d_amd/csrc/f3d_smoke.hip forge3d_amd/csrc/f3d_smoke_sim.hip forge3d_amd/csrc/f3d_composite.hip \
    forge3d_amd/csrc/f3d_lbvh.hip forge3d_amd/csrc/f3d_wavefront.hip forge3d_amd/csrc/f3d_aether_bake.hip forge3d_amd/csrc/f3d_aether_ref.hip -o build_ab/libf3dhip_$NAME.so 2>> build_ab/$NAME.err
rm -f build_ab/f3d_kernels_$NAME.o
ls -la build_ab/libf3dhip_$NAME.so
