// Do short dependent kernels on one stream make progress while a chip-filling kernel runs on another?
// hipcc --offload-arch=gfx950 -O3 tools/experiments/stream_overlap.hip -o /tmp/stream_overlap && /tmp/stream_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void big(float *p, int iters) {  // ~chip-filling, latency-bound loops like the trace kernel's waves
    float x = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
    p[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void small(float *p, int n) {  // a few waves with a ~0.1 ms dependent chain each (the merge / re-trace kernels' shape)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x = i < n ? p[i] : 0.0f;
    for (int k = 0; k < 40000; k++) x = __builtin_fmaf(x, 0.9999999f, 1e-7f);
    if (i < n) p[i] = x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float *a, *b;
    hipMalloc(&a, 64 << 20); hipMalloc(&b, 64 << 20);
    hipMemset(a, 0, 64 << 20); hipMemset(b, 0, 64 << 20);
    hipStream_t s1, s2;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least, hi = greatest priority (numerically lower)
    const bool prio = getenv("PRIO") != nullptr;
    hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, prio ? lo : 0);
    hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, prio ? hi : 0);
    printf("priority range %d..%d, chain stream %s\n", lo, hi, prio ? "high" : "default");
    const int grid = 65536, iters = 20000, chain = 32;
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipDeviceSynchronize();
            double t0 = now();
            if (mode == 0) {  // big alone
                hipLaunchKernelGGL(big, dim3(grid), dim3(64), 0, s1, a, iters);
            } else if (mode == 1) {  // chain alone
                for (int k = 0; k < chain; k++) hipLaunchKernelGGL(small, dim3(64), dim3(64), 0, s2, b, 4096);
            } else {  // both, different streams
                hipLaunchKernelGGL(big, dim3(grid), dim3(64), 0, s1, a, iters);
                for (int k = 0; k < chain; k++) hipLaunchKernelGGL(small, dim3(64), dim3(64), 0, s2, b, 4096);
            }
            hipDeviceSynchronize();
            printf("mode %d (%s): %.3f ms\n", mode, mode == 0 ? "big kernel alone" : mode == 1 ? "chain of 32 small kernels alone" : "both on two streams", (now() - t0) * 1e3);
        }
    }
    return 0;
}
