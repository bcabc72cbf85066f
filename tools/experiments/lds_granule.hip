// tools/experiments/lds_granule.hip -- how many one-wave workgroups with a given LDS size does a CU of gfx950 hold?
// Two answers: what the runtime's occupancy calculator says, and what the hardware does (waves that spin until every
// wave of the launch has started: the launch only finishes if `expected` waves per CU are resident at once).
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/lds_granule.hip -o /tmp/lds_granule && /tmp/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void k_probe(unsigned *count, unsigned *peak, unsigned long long ticks) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    if (threadIdx.x == 0) {
        const unsigned now = atomicAdd(count, 1u) + 1u;
        atomicMax(peak, now);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
        atomicSub(count, 1u);
    }
    __syncthreads();
    if (lds[threadIdx.x] == 0xFFFFFFFFu) peak[1] = 1;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned *d;
    hipMalloc(&d, 16);
    printf("CUs %d\n", cus);
    for (unsigned bytes = 3584; bytes <= 8704; bytes += 256) {
        int per_cu = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_probe, 64, bytes);
        hipMemset(d, 0, 16);
        // far more workgroups than fit: the peak of concurrently running waves / CUs = resident waves per CU
        hipLaunchKernelGGL(k_probe, dim3(cus * 40), dim3(64), bytes, 0, d, d + 1, 20000ull);  // 0.2 ms each
        hipDeviceSynchronize();
        unsigned h[4];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("lds %5u B: runtime says %2d per CU, measured peak %5u = %.2f per CU\n", bytes, per_cu, h[1], (double)h[1] / cus);
    }
    return 0;
}
