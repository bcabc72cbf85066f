// forge3d_amd/csrc/f3d_host_halo.h -- a FRAGMENT of f3d_host.hip (included there, once, after the session and its frame
// loops): peer halos -- the strips of one node pull their neighbours' edge rows themselves (include/f3d_terrain_pt.h,
// DESIGN.md 7): the counter / pull / probe kernels, the batch enqueue, and the f3d_session_halo_* entry points.
// Split out of f3d_host.hip in round 4; one translation unit as before.
#pragma once

// ---- peer halos --------------------------------------------------------------------------------------------------
namespace {
__global__ void k_halo_flag(uint32_t *flag, uint32_t frames_merged) {
    __hip_atomic_store(flag, frames_merged, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct HaloPullParams {
    const uint32_t *flag[2];  // the neighbours' "frames merged" counters (null: no neighbour on that side)
    const unsigned long long *src[2];
    unsigned long long *dst[2];
    uint32_t words;           // 8-byte words per halo block
    uint32_t want;            // frames the neighbour must have merged (probe: the nonce its word must hold)
    uint32_t exact;           // 0: wait for flag >= want (frame counters only rise); 1: for flag == want (link probe)
    uint32_t *counters;       // this strip's counter block (f3d_session::halo_flags)
    uint32_t *publish;        // word of it to set to `want` first (the kernels of the frame are behind us on the stream); null: nothing
    unsigned long long timeout_ticks;
    uint32_t *checksum;       // link probe: [side] = sum of the 32-bit halves of the block pulled; null in the frame loop
};
// One workgroup per neighbour: wait until it has merged `want` frames, then copy its edge rows into my halo rows.
// Every access to the neighbour's memory is a system-scope load that bypasses this device's caches.
// The wait is timed into the counter block (words 8..13): the first multi-GPU run then says how long a strip stood here.
// A wait that exceeds timeout_ticks gives up and counts in word 1; a strip that has timed out stops pulling for the rest of
// the CALL that enqueued it (its halo rows are stale, the caller must fail the render: f3d_session_halo_status) -- the
// next f3d_session_enqueue_batch_strip starts with the count cleared.
__global__ __launch_bounds__(1024) void k_halo_pull(const HaloPullParams H) {
    const uint32_t side = blockIdx.x;
    if (H.publish && side == 0u && threadIdx.x == 0u) __hip_atomic_store(H.publish, H.want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!H.flag[side]) return;
    __shared__ uint32_t ok;
    __shared__ uint32_t sum;
    if (threadIdx.x == 0u) {
        sum = 0u;
        const unsigned long long t0 = wall_clock64();
        ok = __hip_atomic_load(H.counters + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u ? 1u : 0u;  // a neighbour is gone: do not wait again
        unsigned long long waited = 0ull;
        while (ok != 0u) {
            const uint32_t seen = __hip_atomic_load(H.flag[side], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (H.exact ? seen == H.want : seen >= H.want) break;
            __builtin_amdgcn_s_sleep(16);
            waited = wall_clock64() - t0;
            if (waited > H.timeout_ticks) {  // the neighbour is gone
                ok = 0u;
                atomicAdd(H.counters + 1, 1u);
            }
        }
        atomicAdd(reinterpret_cast<unsigned long long *>(H.counters + 8) + side, waited);
        atomicAdd(H.counters + 12, 1u);
        atomicMax(H.counters + 13, (uint32_t)(waited > 0xFFFFFFFFull ? 0xFFFFFFFFull : waited));
    }
    __syncthreads();
    if (ok == 0u) return;
    uint32_t part = 0u;
    for (uint32_t i = threadIdx.x; i < H.words; i += blockDim.x) {
        const unsigned long long v = __hip_atomic_load(H.src[side] + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        H.dst[side][i] = v;
        part += (uint32_t)v + (uint32_t)(v >> 32);
    }
    if (H.checksum) {
        atomicAdd(&sum, part);
        __syncthreads();
        if (threadIdx.x == 0u) H.checksum[side] = sum;
    }
}

// Link probe with a REAL block (f3d_session_halo_probe modes 2 / 3): a strip fills its two edge blocks of reservoir buffer
// 0 with a pattern of its nonce -- a kernel of many workgroups, so the lines are dirty in the L2 of every XCD, as the
// frame kernels leave them -- and publishes the nonce behind it; its neighbours wait for the nonce and pull the block with
// the frame loop's own kernel, which also sums it.  A stale line, a mapping of the wrong buffer or an edge-row offset that
// is off by a row shows as a checksum that is not the pattern's.
__host__ __device__ inline uint32_t halo_probe_word(uint32_t nonce, uint32_t index) {  // 32-bit half `index` of the buffer
    uint32_t x = nonce ^ (index * 0x9E3779B9u);
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 12;
    return x;
}
__global__ void k_halo_probe_fill(uint32_t *buffer, uint32_t first, uint32_t count, uint32_t nonce) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) buffer[first + i] = halo_probe_word(nonce, first + i);
}

// Link check (f3d_session_halo_probe): word 2 of a strip's counter block is a nonce its owner stores and its
// neighbours read back, with the accesses the frame loop uses.
__global__ void k_halo_probe_read(const uint32_t *above, const uint32_t *below, uint32_t *out) {
    out[0] = above ? __hip_atomic_load(above + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
    out[1] = below ? __hip_atomic_load(below + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
}

void enqueue_halo_sync(f3d_session &s, uint32_t frame) {  // behind the kernels of `frame`
    if (!s.peer[0].connected && !s.peer[1].connected) {  // nobody to pull from: publish only
        hipLaunchKernelGGL(k_halo_flag, dim3(1), dim3(1), 0, s.stream, s.halo_flags, frame + 1u);
        return;
    }
    HaloPullParams H{};  // one launch: publish my counter, then wait for and copy the neighbours' rows
    H.publish = s.halo_flags;
    H.counters = s.halo_flags;
    H.timeout_ticks = s.halo_timeout_ticks;
    const size_t row = (size_t)s.width, block = (size_t)kHaloRows * row;
    const uint32_t which = frame & 1u;
    H.words = (uint32_t)(block * sizeof(PackedReservoir) / 8u);
    H.want = frame + 1u;
    if (s.peer[0].connected) {  // the strip above: its BOTTOM owned rows -> my halo above
        H.flag[0] = s.peer[0].flags;
        H.src[0] = (const unsigned long long *)(s.peer[0].res[which] + (size_t)s.peer[0].rows * row);
        H.dst[0] = (unsigned long long *)(s.res[which]);
    }
    if (s.peer[1].connected) {  // the strip below: its TOP owned rows -> my halo below
        H.flag[1] = s.peer[1].flags;
        H.src[1] = (const unsigned long long *)(s.peer[1].res[which] + block);
        H.dst[1] = (unsigned long long *)(s.res[which] + ((size_t)s.rows + kHaloRows) * row);
    }
    hipLaunchKernelGGL(k_halo_pull, dim3(2), dim3(1024), 0, s.stream, H);
    hip_check(hipGetLastError(), "halo pull kernel");
}

// Frames [first, first + count) of a strip whose neighbours are connected: enqueue_range with the halo step after
// every frame.
void enqueue_batch_strip(f3d_session &s, uint32_t first, uint32_t count, bool collect_last) {
    if (!s.halo_flags) fail(F3D_STATUS_VALUE, "f3d_session_halo_export has not been called for this session");
    // The neighbours wait for `counter >= frame + 1` on a counter that is never cleared: frames of a connected session only
    // go up (a second pass over frames 0.. would find the neighbours' counters high already and pull rows of the wrong frame).
    if (first < s.halo_frames_published)
        fail(F3D_STATUS_VALUE, "peer halos: frame %u was enqueued before (this strip has published %u frames); frame numbers of a connected session only rise",
             first, s.halo_frames_published);
    s.halo_frames_published = first + count;
    hip_check(hipMemsetAsync(s.halo_flags + 1, 0, sizeof(uint32_t), s.stream), "halo time-out count");  // a new call waits again
    if (s.fd_frames) {
        for (uint32_t done = 0; done < count;) {
            const uint32_t n = trace_batch(s, first + done, count - done);
            enqueue_trace(s, first + done, n);
            for (uint32_t i = 0; i < n; i++) {
                enqueue_merge(s, first + done + i, collect_last && done + i + 1 == count);
                enqueue_halo_sync(s, first + done + i);
            }
            done += n;
        }
        return;
    }
    for (uint32_t i = 0; i < count; i++) {
        fork_bands(s, collect_last && i + 1 == count);
        enqueue_frame(s, first + i, collect_last && i + 1 == count, 0u, false);
        join_bands(s);
        enqueue_halo_sync(s, first + i);
    }
}
}  // namespace


extern "C" {

int f3d_session_halo_export(f3d_session *s, f3d_halo_export *out, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        if (!out) fail(F3D_STATUS_VALUE, "null export record");
        if (!s->owns_reservoirs) fail(F3D_STATUS_VALUE, "peer halos need reservoirs owned by the session (no ext_reservoirs)");
        if (!s->halo_flags) {
            hip_check(hipMalloc((void **)&s->halo_flags, 256), "halo counters");
            hip_check(hipMemset(s->halo_flags, 0, 256), "halo counters");
            int khz = 0;  // the clock wall_clock64() counts (100 MHz on gfx950)
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, s->device) == hipSuccess && khz > 0) s->wall_clock_khz = (double)khz;
            (void)hipGetLastError();
            double ms = 20000.0;  // a neighbour that is merely slow (first-launch code load, a profiler, a shared GPU) is not gone
            if (const char *e = getenv("F3D_HALO_TIMEOUT_MS")) ms = atof(e) > 0.0 ? atof(e) : ms;
            s->halo_timeout_ticks = (unsigned long long)(ms * s->wall_clock_khz);
        }
        memset(out, 0, sizeof(*out));
        void *objects[3] = {s->res[0], s->res[1], s->halo_flags};
        for (int i = 0; i < 3; i++) {
            void *base = nullptr;
            size_t size = 0;
            hip_check(hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)objects[i]), "allocation of a halo object");
            hipIpcMemHandle_t h;
            hip_check(hipIpcGetMemHandle(&h, base), "hipIpcGetMemHandle");
            static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
            memcpy(out->handle[i], &h, 64);
            out->offset[i] = (uint64_t)((char *)objects[i] - (char *)base);
            out->address[i] = (uint64_t)(uintptr_t)objects[i];
        }
        out->rows = s->rows;
        out->width = s->width;
        out->device = s->device;
        out->pid = (uint32_t)getpid();
        hip_check(hipStreamSynchronize(s->stream), "export sync");
    });
}

int f3d_session_halo_connect(f3d_session *s, int32_t side, const f3d_halo_export *peer, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        if (side < 0 || side > 1 || !peer) fail(F3D_STATUS_VALUE, "halo connect: side must be 0 (above) or 1 (below)");
        if (peer->width != s->width) fail(F3D_STATUS_VALUE, "halo connect: the neighbour renders another image width");
        if (!s->halo_flags) fail(F3D_STATUS_VALUE, "f3d_session_halo_export has not been called for this session");
        f3d_session::PeerLink &L = s->peer[side];
        if (L.connected) fail(F3D_STATUS_VALUE, "halo connect: side %d is connected already", side);
        void *mapped[3];
        for (int i = 0; i < 3; i++) {
            hipIpcMemHandle_t h;
            memcpy(&h, peer->handle[i], 64);
            void *base = nullptr;
            if (peer->pid == (uint32_t)getpid()) {
                mapped[i] = (void *)(uintptr_t)peer->address[i];  // a process cannot open its own handles: same address space
                if (peer->device != s->device) {  // (one process driving several GPUs: map the neighbour's memory here)
                    const hipError_t e = hipDeviceEnablePeerAccess(peer->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) hip_check(e, "hipDeviceEnablePeerAccess");
                    (void)hipGetLastError();
                }
                continue;
            } else {
                hip_check(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
                L.opened[i] = base;
            }
            mapped[i] = (char *)base + peer->offset[i];
        }
        L.res[0] = (const PackedReservoir *)mapped[0];
        L.res[1] = (const PackedReservoir *)mapped[1];
        L.flags = (const uint32_t *)mapped[2];
        L.rows = peer->rows;
        L.connected = true;
    });
}

int f3d_session_halo_probe(f3d_session *s, int32_t mode, uint32_t nonce, uint32_t *seen, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        if (!s->halo_flags) fail(F3D_STATUS_VALUE, "f3d_session_halo_export has not been called for this session");
        if (mode == 0) {  // publish: the store the frame loop uses for its counter
            hipLaunchKernelGGL(k_halo_flag, dim3(1), dim3(1), 0, s->stream, s->halo_flags + 2, nonce);
            hip_check(hipStreamSynchronize(s->stream), "halo probe store");
        } else if (mode == 1) {  // read the neighbours' words with the loads the pull uses
            if (!seen) fail(F3D_STATUS_VALUE, "null output");
            hipLaunchKernelGGL(k_halo_probe_read, dim3(1), dim3(1), 0, s->stream, s->peer[0].connected ? s->peer[0].flags : nullptr,
                               s->peer[1].connected ? s->peer[1].flags : nullptr, s->halo_flags + 4);
            hip_check(hipStreamSynchronize(s->stream), "halo probe load");
            hip_check(hipMemcpy(seen, s->halo_flags + 4, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost), "halo probe read-back");
        } else if (mode == 2) {  // fill my two edge blocks of buffer 0 with the pattern of `nonce`, then publish it
            const uint32_t row_words = s->width * (uint32_t)(sizeof(PackedReservoir) / 4u), block_words = kHaloRows * row_words;
            uint32_t *buf = reinterpret_cast<uint32_t *>(s->res[0]);
            for (uint32_t first_row : {kHaloRows, s->rows})  // top owned rows, bottom owned rows (they overlap in strips of < 8 rows: one pattern)
                hipLaunchKernelGGL(k_halo_probe_fill, dim3((block_words + 255u) / 256u), dim3(256), 0, s->stream, buf, first_row * row_words, block_words, nonce);
            hipLaunchKernelGGL(k_halo_flag, dim3(1), dim3(1), 0, s->stream, s->halo_flags + 2, nonce);
            hip_check(hipGetLastError(), "halo probe fill");
        } else if (mode == 3) {  // pull the neighbours' blocks (they publish seen[0] above / seen[1] below) and check their sums
            if (!seen) fail(F3D_STATUS_VALUE, "null nonces");
            HaloPullParams H{};
            H.counters = s->halo_flags;
            H.timeout_ticks = s->halo_timeout_ticks;
            H.exact = 1u;
            H.checksum = s->halo_flags + 6;
            const size_t row = (size_t)s->width, block = (size_t)kHaloRows * row;
            const uint32_t row_words = s->width * (uint32_t)(sizeof(PackedReservoir) / 4u), block_words = kHaloRows * row_words;
            H.words = (uint32_t)(block * sizeof(PackedReservoir) / 8u);
            uint32_t expect[2] = {0u, 0u};
            bool ok = true;
            for (int side = 0; side < 2; side++) {  // one launch per side: each waits for its own neighbour's nonce
                if (!s->peer[side].connected) continue;
                HaloPullParams one = H;
                one.want = seen[side];
                const uint32_t first_row = side == 0 ? s->peer[0].rows : kHaloRows;  // the neighbour's bottom / top owned rows
                one.flag[side] = s->peer[side].flags + 2;
                one.src[side] = (const unsigned long long *)(s->peer[side].res[0] + (size_t)first_row * row);
                one.dst[side] = (unsigned long long *)(s->res[0] + (side == 0 ? 0 : ((size_t)s->rows + kHaloRows) * row));
                hipLaunchKernelGGL(k_halo_pull, dim3(2), dim3(1024), 0, s->stream, one);
                for (uint32_t i = 0; i < block_words; i++) expect[side] += halo_probe_word(seen[side], first_row * row_words + i);
            }
            hip_check(hipStreamSynchronize(s->stream), "halo probe pull");
            uint32_t got[8];
            hip_check(hipMemcpy(got, s->halo_flags, sizeof(got), hipMemcpyDeviceToHost), "halo probe read-back");
            for (int side = 0; side < 2; side++) {
                if (!s->peer[side].connected) continue;
                if (got[1] != 0u || got[6 + side] != expect[side]) ok = false;
                seen[side] = got[6 + side] == expect[side] ? 1u : 0u;
            }
            hip_check(hipMemsetAsync(s->halo_flags + 1, 0, sizeof(uint32_t), s->stream), "halo time-out count");
            hip_check(hipStreamSynchronize(s->stream), "halo probe pull");
            if (!ok)
                fail(F3D_STATUS_DEVICE, "peer halos: the block pulled from a neighbouring strip is not the block it wrote (sums above %08x / %08x, below %08x / %08x, %u time-outs)",
                     got[6], expect[0], got[7], expect[1], got[1]);
        } else if (mode == 4) {  // the probes wrote into reservoir buffer 0 (edge rows in mode 2, halo rows in mode 3): as a new session has it
            const size_t row = (size_t)s->width;
            hip_check(hipMemsetAsync(s->res[0], 0, ((size_t)s->rows + 2u * kHaloRows) * row * sizeof(PackedReservoir), s->stream), "reservoir clear");
            hip_check(hipStreamSynchronize(s->stream), "halo probe clear");
        } else {
            fail(F3D_STATUS_VALUE, "halo probe: mode must be 0 (publish), 1 (read), 2 (fill + publish a block), 3 (pull + check the blocks) or 4 (clear)");
        }
    });
}

int f3d_session_halo_stats(f3d_session *s, f3d_halo_stats *out, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        if (!out) fail(F3D_STATUS_VALUE, "null output");
        const bool reset = out->reset != 0u;
        memset(out, 0, sizeof(*out));
        if (!s->halo_flags) return;
        hip_check(hipStreamSynchronize(s->stream), "halo stats");
        uint32_t w[16];
        hip_check(hipMemcpy(w, s->halo_flags, sizeof(w), hipMemcpyDeviceToHost), "halo stats");
        const double ms_per_tick = 1.0 / s->wall_clock_khz;
        out->frames_published = w[0];
        out->timeouts = w[1];
        out->pulls = w[12];
        out->wait_ms[0] = (double)(((unsigned long long)w[9] << 32) | w[8]) * ms_per_tick;
        out->wait_ms[1] = (double)(((unsigned long long)w[11] << 32) | w[10]) * ms_per_tick;
        out->longest_wait_ms = (double)w[13] * ms_per_tick;
        out->timeout_ms = (double)s->halo_timeout_ticks * ms_per_tick;
        if (reset) hip_check(hipMemsetAsync(s->halo_flags + 8, 0, 6 * sizeof(uint32_t), s->stream), "halo stats reset");
    });
}

int f3d_session_halo_status(f3d_session *s, uint32_t *timeouts, char *err, size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        if (!timeouts) fail(F3D_STATUS_VALUE, "null output");
        *timeouts = 0u;
        if (s->halo_flags) {
            hip_check(hipStreamSynchronize(s->stream), "halo status");
            hip_check(hipMemcpy(timeouts, s->halo_flags + 1, sizeof(uint32_t), hipMemcpyDeviceToHost), "halo status");
        }
    });
}

int f3d_session_enqueue_batch_strip(f3d_session *s, uint32_t first_frame, uint32_t count, int32_t collect_stats_on_last, char *err,
                                    size_t errlen) {
    return c_abi(err, errlen, [&] {
        DeviceGuard g(checked(s).device);
        enqueue_batch_strip(*s, first_frame, count, collect_stats_on_last != 0);
    });
}

int f3d_session_halo(f3d_session *s, int32_t which, int32_t side, void **ptr, uint64_t *bytes) {
    if (!s || !ptr || !bytes || which < 0 || which > 1 || side < 0 || side > 3) return F3D_STATUS_VALUE;
    const size_t row = (size_t)s->width;
    size_t first;
    switch (side) {
        case 0: first = kHaloRows; break;                 // top owned rows
        case 1: first = s->rows; break;                   // bottom owned rows
        case 2: first = 0; break;                         // halo above
        default: first = (size_t)s->rows + kHaloRows; break;  // halo below
    }
    *ptr = (void *)(s->res[which] + first * row);
    *bytes = (uint64_t)kHaloRows * row * sizeof(PackedReservoir);
    return F3D_STATUS_OK;
}

}  // extern "C"

