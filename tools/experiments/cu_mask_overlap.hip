// Can a chain of short dependent kernels run BESIDE a chip-filling kernel when the two streams own disjoint CUs?
// (tools/experiments/stream_overlap.hip: without masks the chain waits behind the big kernel's waves -- 11.0 ms against
// 9.0 + 2.6 alone.)  hipExtStreamCreateWithCUMask: bit i of the mask = CU i/8 of XCD i%8 on a multi-XCD device.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/cu_mask_overlap.hip -o /tmp/cu_mask_overlap && /tmp/cu_mask_overlap [chain CUs]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void big(float *p, int iters) {
    float x = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
    p[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void small(float *p, int n, int iters) {  // the merge kernel's shape: a few thousand waves of a short dependent chain
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x = i < n ? p[i] : 0.0f;
    for (int k = 0; k < iters; k++) x = __builtin_fmaf(x, 0.9999999f, 1e-7f);
    if (i < n) p[i] = x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int chain_cus = argc > 1 ? atoi(argv[1]) : 32;
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *a, *b;
    (void)hipMalloc(&a, 64 << 20); (void)hipMalloc(&b, 64 << 20);
    (void)hipMemset(a, 0, 64 << 20); (void)hipMemset(b, 0, 64 << 20);
    std::vector<uint32_t> m1((cus + 31) / 32, 0u), m2((cus + 31) / 32, 0u);
    for (int i = 0; i < cus; i++) (i < chain_cus ? m2 : m1)[i / 32] |= 1u << (i % 32);
    hipStream_t s1, s2, p1, p2;
    hipError_t e1 = hipExtStreamCreateWithCUMask(&s1, (uint32_t)m1.size(), m1.data());
    hipError_t e2 = hipExtStreamCreateWithCUMask(&s2, (uint32_t)m2.size(), m2.data());
    (void)hipStreamCreateWithFlags(&p1, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&p2, hipStreamNonBlocking);
    printf("%d CUs, chain stream owns %d; create: %s / %s\n", cus, chain_cus, hipGetErrorString(e1), hipGetErrorString(e2));
    const int grid = 65536, iters = 20000, chain = 32;
    // chain kernel: 4 050 waves (an eighth of a 1080p frame, one lane per pixel) x 2 000 dependent instructions
    for (int masked = 0; masked < 2; masked++) {
        hipStream_t big_s = masked ? s1 : p1, chain_s = masked ? s2 : p2;
        for (int mode = 0; mode < 3; mode++) {
            double best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                (void)hipDeviceSynchronize();
                double t0 = now();
                if (mode != 1) hipLaunchKernelGGL(big, dim3(grid), dim3(64), 0, big_s, a, iters);
                if (mode != 0) for (int k = 0; k < chain; k++) hipLaunchKernelGGL(small, dim3(4050), dim3(64), 0, chain_s, b, 4050 * 64, 2000);
                (void)hipDeviceSynchronize();
                const double ms = (now() - t0) * 1e3;
                best = ms < best ? ms : best;
            }
            printf("%s, %s: %.3f ms\n", masked ? "disjoint CU masks" : "plain streams", mode == 0 ? "big kernel alone" : mode == 1 ? "chain of 32 merge-sized kernels alone" : "both", best);
        }
    }
    return 0;
}
