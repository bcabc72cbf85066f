// forge3d_amd/csrc/f3d_devmem.h
// Every device allocation of the library goes through here, so that one switch (f3d_debug_poison, include/
// f3d_terrain_pt.h) can put the library into POISON mode: each buffer then sits between two guard regions and all three
// are filled with a byte pattern.  A result must not depend on the pattern -- if it does, some kernel reads memory
// nobody wrote (uninitialised, or beyond a buffer: silent on the GPU, where such a read returns whatever the allocator
// put next door and the image depends on the process's history; DESIGN.md 8, the 4-row reach of the spatial pass).
// tools/gpu_fuzz*.py render every configuration under different patterns and compare.
#pragma once
#include <cstdio>

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>

namespace f3d {

constexpr size_t kPoisonGuardBytes = 256u << 10;

int poison_pattern();  // -1: off (f3d_host.hip)
void poison_register(void *user, void *base);
void *poison_take(void *user);  // base pointer of a poisoned allocation (and forget it); nullptr: a plain allocation

// Freed blocks are kept (per device, exact size, newest first; F3D_DEVICE_POOL_MB, default 1 024, 0 = off) and handed out
// again: a camera path or a smoke sequence allocates the same dozen buffers for every frame, and hipFree waits for the
// device.  Callers free only what no stream still uses (sessions synchronise their streams before they go).  Poisoned
// allocations bypass the pool (their guard regions are part of the pattern test).  f3d_device_pool_trim() empties it (the strip driver
// does so before it sizes its torch / RCCL buffers: memory the pool holds is invisible to torch's allocator).
hipError_t pool_take(void **out, size_t bytes);   // hipErrorOutOfMemory: nothing of that size waiting (f3d_host.hip)
bool pool_give(void *p);                           // false: not taken (pool off or full): the caller frees
void pool_note(void *p, size_t bytes);             // a fresh hipMalloc the pool may take back later
void pool_trim();

inline hipError_t device_alloc(void **out, size_t bytes) {
    if (bytes == 0) bytes = 16;
    const int pattern = poison_pattern();
    if (pattern < 0) {
        if (pool_take(out, bytes) == hipSuccess) return hipSuccess;
        hipError_t e = hipMalloc(out, bytes);
        if (e == hipErrorOutOfMemory) {  // what the pool holds may be what is missing
            (void)hipGetLastError();
            pool_trim();
            e = hipMalloc(out, bytes);
        }
        if (e == hipSuccess) pool_note(*out, bytes);
        return e;
    }
    const size_t padded = (bytes + 255u) & ~(size_t)255u;  // keeps the alignment hipMalloc gives
    void *base = nullptr;
    hipError_t e = hipMalloc(&base, padded + 2u * kPoisonGuardBytes);
    if (e != hipSuccess) return e;
    if ((e = hipMemset(base, pattern, padded + 2u * kPoisonGuardBytes)) != hipSuccess) {
        (void)hipFree(base);
        return e;
    }
    // the fill runs on the null stream and need not have finished when hipMemset returns; a session on a NON-BLOCKING stream
    // (torch's: the strip drivers') is not ordered behind it, and its first kernels raced the pattern -- whole strips of
    // garbage in poison mode only (round 5: the 4096^2 peer-halo case of the device suite under F3D_POISON).  Debug mode: wait.
    if ((e = hipDeviceSynchronize()) != hipSuccess) {
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
    if (!base) {
        // (pool_give waits for the block's own device before it keeps a block; a block it does not keep -- pool off, full,
        // or a pointer it never saw -- goes to hipFree, which waits by itself)
        if (pool_give(p)) return hipSuccess;
    }
    return hipFree(base ? base : p);
}

// ---- stream-ordered scratch that outlives a call (round 5) -----------------------------------------------------------------
// The smoke entry points (f3d_smoke_step / _render / _composite) run on the null stream and used to allocate their scratch
// per call and free it at the end -- which forced every call to wait for its own kernels and, through the pool's wait,
// for the device: 0.55 ms of host time around 1.9 ms of kernels per frame of a resident sequence (BASELINE.json configs[4]).
// A workspace buffer is identified by a tag, grows when a call needs more, and stays: the next call's kernels are behind
// this call's in the stream, so reuse needs no wait, and a call whose results stay on the device can return as soon as its
// launches are enqueued.  One caller at a time per device (workspace_lock: held while a call enqueues); buffers are per stream
// (call_stream() below), since only one stream's launches are ordered behind each other.
// f3d_device_pool_trim() frees the workspace too.
struct WorkspaceEntry {
    void *p = nullptr;
    size_t bytes = 0;
};
inline std::mutex &workspace_lock() {
    static std::mutex &m = *new std::mutex();
    return m;
}
inline std::map<std::pair<int, std::string>, WorkspaceEntry> &workspace_map() {
    static auto &m = *new std::map<std::pair<int, std::string>, WorkspaceEntry>();  // (never destroyed: static destructors run after the HIP runtime has gone)
    return m;
}
// What the smoke entry points (solver step, marcher, composite) of ONE call run in: the stream they enqueue on, the event that
// marks the moment the marcher has finished READING the volume's fields (its pack kernel: what a solver step on another stream
// waits for before it overwrites them), and the name space of their scratch.
//   * a sequence handle (f3d_smoke_seq_*, ABI 6) owns a context: explicit state, two sequences never share anything;
//   * the handle-less entry points (f3d_smoke_step / _render / _composite) use the calling thread's default context -- the null
//     stream, synchronous calls as in ABI 4, until f3d_smoke_set_stream (the ABI-5 shim) names a stream for the thread.
// The "current" pointer is set for the duration of a handle call and restored behind it (ScopedSmokeContext): it is how the
// shared implementation finds its context, not state that outlives a call.
struct SmokeContext {
    hipStream_t stream = nullptr;
    hipEvent_t fields_read = nullptr;
    unsigned long long id = 0;  // 0: the thread's default context (scratch named by stream, as in ABI 5); else "#id"
    bool async_ok = false;      // may a call whose results stay on the device return with its launches enqueued?
    // the marcher's deferred self-shadow list of the last render: capacity, and where its fill count lives (read on request)
    uint32_t shadow_capacity = 0;
    const uint32_t *shadow_cursor = nullptr;
};
inline SmokeContext &thread_default_smoke_context() {
    static thread_local SmokeContext context;
    return context;
}
inline SmokeContext *&current_smoke_context() {
    static thread_local SmokeContext *current = nullptr;
    if (!current) current = &thread_default_smoke_context();
    return current;
}
struct ScopedSmokeContext {
    SmokeContext *saved;
    explicit ScopedSmokeContext(SmokeContext *c) : saved(current_smoke_context()) { current_smoke_context() = c; }
    ~ScopedSmokeContext() { current_smoke_context() = saved; }
};
inline hipStream_t &call_stream() { return current_smoke_context()->stream; }
inline hipEvent_t &fields_read_event() { return current_smoke_context()->fields_read; }

// (call with workspace_lock() held) a buffer of at least `bytes` for `tag` on the current device; hipErrorOutOfMemory etc. on failure
inline hipError_t workspace(void **out, const char *tag, size_t bytes) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    // (a buffer belongs to the tag AND to whoever's launches are ordered behind each other on it: a sequence handle, or --
    // handle-less calls -- the stream: two sequences of a process must not march through each other's records)
    char where[40];
    const SmokeContext &ctx = *current_smoke_context();
    if (ctx.id != 0ull) snprintf(where, sizeof(where), "#%llu", ctx.id);
    else snprintf(where, sizeof(where), "@%p", (void *)ctx.stream);
    WorkspaceEntry &w = workspace_map()[{device, std::string(tag) + where}];
    if (w.bytes < bytes || !w.p) {
        if (w.p) {
            (void)hipDeviceSynchronize();  // (rare: a larger domain than before) whatever still reads the old buffer
            (void)device_free(w.p);
            w = WorkspaceEntry{};
        }
        e = device_alloc(&w.p, bytes);
        if (e != hipSuccess) {
            w = WorkspaceEntry{};
            return e;
        }
        w.bytes = bytes;
    }
    *out = w.p;
    return hipSuccess;
}
// Bytes of scratch held for one name space ("#id" / "@stream"), and their release (a sequence handle that goes away).
inline size_t workspace_bytes(const std::string &suffix) {
    std::lock_guard<std::mutex> lock(workspace_lock());
    size_t total = 0;
    for (auto &kv : workspace_map())
        if (kv.first.second.size() >= suffix.size() && kv.first.second.compare(kv.first.second.size() - suffix.size(), suffix.size(), suffix) == 0) total += kv.second.bytes;
    return total;
}
inline void workspace_release(const std::string &suffix) {  // the caller has drained the streams that used it
    std::lock_guard<std::mutex> lock(workspace_lock());
    for (auto it = workspace_map().begin(); it != workspace_map().end();) {
        const std::string &name = it->first.second;
        if (name.size() >= suffix.size() && name.compare(name.size() - suffix.size(), suffix.size(), suffix) == 0) {
            if (it->second.p) (void)device_free(it->second.p);
            it = workspace_map().erase(it);
        } else {
            ++it;
        }
    }
}
inline void workspace_trim() {
    std::lock_guard<std::mutex> lock(workspace_lock());
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (auto &kv : workspace_map()) {
        (void)hipSetDevice(kv.first.first);
        (void)hipDeviceSynchronize();
        if (kv.second.p) (void)device_free(kv.second.p);
    }
    workspace_map().clear();
    if (prev >= 0) (void)hipSetDevice(prev);
}

}  // namespace f3d
