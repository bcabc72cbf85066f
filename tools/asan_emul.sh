#!/bin/bash
# The kernels' shared host/device code (csrc/*.h) under a sanitizer: builds the 64-lane host emulator with
# AddressSanitizer (default) or UBSan incl. the float checks (`ubsan`) and runs every CPU test that goes through it.
# An out-of-bounds read in device code is silent on the GPU (it returns whatever the allocator put next door, so results
# depend on the process's history); here it stops the run.  Found the 4-row reach of the spatial pass
# (tests/test_halo_reach.py).  tools/asan_emul.sh [asan|ubsan] [pytest args]
set -e
cd "$(dirname "$0")/.."
MODE=${1:-asan}; [ $# -gt 0 ] && shift
if [ "$MODE" = ubsan ]; then
    export F3D_EMUL_CXXFLAGS="-fsanitize=undefined,float-cast-overflow,float-divide-by-zero -fno-sanitize-recover=all -fno-omit-frame-pointer -g"
    PRE="$(gcc -print-file-name=libubsan.so) $(gcc -print-file-name=libstdc++.so)"
else
    export F3D_EMUL_CXXFLAGS="-fsanitize=address -fno-omit-frame-pointer -g"
    PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)"
fi
python -c "import sys; sys.path.insert(0, 'tests'); from emul import emul; emul.build(force=True)"
rc=0
LD_PRELOAD="$PRE" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_emul_parity.py tests/test_halo_reach.py tests/test_adversarial_march.py tests/test_wavefront.py \
    tests/test_distributed_gloo.py tests/test_primary_start.py -q -m "not gpu" -p no:cacheprovider "$@" || rc=$?
unset F3D_EMUL_CXXFLAGS
python -c "import sys; sys.path.insert(0, 'tests'); from emul import emul; emul.build(force=True)"
exit $rc
