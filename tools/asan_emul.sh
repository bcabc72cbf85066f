#!/bin/bash
# The kernels' shared host/device code (csrc/*.h) under AddressSanitizer: builds the 64-lane host emulator with
# -fsanitize=address and runs every CPU test that goes through it.  An out-of-bounds read in device code is silent
# on the GPU (it returns whatever the allocator put next door, so results depend on the process's history); here it
# stops the run.  Found the 4-row reach of the spatial pass (tests/test_halo_reach.py).
set -e
cd "$(dirname "$0")/.."
export F3D_EMUL_CXXFLAGS="-fsanitize=address -fno-omit-frame-pointer -g"
python -c "import sys; sys.path.insert(0, 'tests'); from emul import emul; emul.build(force=True)"
rc=0
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_emul_parity.py tests/test_halo_reach.py tests/test_adversarial_march.py tests/test_wavefront.py \
    tests/test_distributed_gloo.py -q -m "not gpu" -p no:cacheprovider "$@" || rc=$?
unset F3D_EMUL_CXXFLAGS
python -c "import sys; sys.path.insert(0, 'tests'); from emul import emul; emul.build(force=True)"
exit $rc
