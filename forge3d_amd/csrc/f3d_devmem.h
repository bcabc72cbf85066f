// forge3d_amd/csrc/f3d_devmem.h
// Every device allocation of the library goes through here, so that one switch (f3d_debug_poison, include/
// f3d_terrain_pt.h) can put the library into POISON mode: each buffer then sits between two guard regions and all three
// are filled with a byte pattern.  A result must not depend on the pattern -- if it does, some kernel reads memory
// nobody wrote (uninitialised, or beyond a buffer: silent on the GPU, where such a read returns whatever the allocator
// put next door and the image depends on the process's history; DESIGN.md 8, the 4-row reach of the spatial pass).
// tools/gpu_fuzz*.py render every configuration under different patterns and compare.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace f3d {

constexpr size_t kPoisonGuardBytes = 256u << 10;

int poison_pattern();  // -1: off (f3d_host.hip)
void poison_register(void *user, void *base);
void *poison_take(void *user);  // base pointer of a poisoned allocation (and forget it); nullptr: a plain allocation

inline hipError_t device_alloc(void **out, size_t bytes) {
    if (bytes == 0) bytes = 16;
    const int pattern = poison_pattern();
    if (pattern < 0) return hipMalloc(out, bytes);
    const size_t padded = (bytes + 255u) & ~(size_t)255u;  // keeps the alignment hipMalloc gives
    void *base = nullptr;
    hipError_t e = hipMalloc(&base, padded + 2u * kPoisonGuardBytes);
    if (e != hipSuccess) return e;
    if ((e = hipMemset(base, pattern, padded + 2u * kPoisonGuardBytes)) != hipSuccess) {
        (void)hipFree(base);
        return e;
    }
    *out = (char *)base + kPoisonGuardBytes;
    poison_register(*out, base);
    return hipSuccess;
}

inline hipError_t device_free(void *p) {
    if (!p) return hipSuccess;
    void *base = poison_take(p);
    return hipFree(base ? base : p);
}

}  // namespace f3d
