// tools/experiments/valu_calib.hip -- what do the SQ VALU counters read at a KNOWN issue rate on gfx950?
//
// VERDICT r2 (weak 8): profiles/README.md called k_frame "VALU pipes saturated" from 4 * SQ_ACTIVE_INST_VALU /
// (SIMDs * cycles) = 1.1-1.17, which cannot be a utilisation.  This microkernel issues nothing but independent
// v_fma_f32 (16 accumulators, so no dependency stall at any occupancy) from W waves per SIMD on every SIMD of the
// chip and reports cycles per wave-instruction per SIMD from s_memtime, then the same launch is counted by
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
// (tools/gpu_valu_calib.sh).  Build: hipcc --offload-arch=gfx950 -O3 tools/experiments/valu_calib.hip -o build_ab/valu_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define FMA(a) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y))

template <int MASKED>
__global__ __launch_bounds__(64) void k_fma(float *out, unsigned long long *cycles, int iters, float x, float y) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, a8 = 8, a9 = 9, a10 = 10, a11 = 11, a12 = 12,
          a13 = 13, a14 = 14, a15 = 15;
    const unsigned long long t0 = __builtin_readcyclecounter();
    // MASKED: only the first MASKED lanes execute (EXEC mask) -- does a half-empty wave issue faster?
    if (MASKED == 0 || (int)threadIdx.x < MASKED) {
        for (int i = 0; i < iters; i++) {
            FMA(a0); FMA(a1); FMA(a2); FMA(a3); FMA(a4); FMA(a5); FMA(a6); FMA(a7);
            FMA(a8); FMA(a9); FMA(a10); FMA(a11); FMA(a12); FMA(a13); FMA(a14); FMA(a15);
            FMA(a0); FMA(a1); FMA(a2); FMA(a3); FMA(a4); FMA(a5); FMA(a6); FMA(a7);
            FMA(a8); FMA(a9); FMA(a10); FMA(a11); FMA(a12); FMA(a13); FMA(a14); FMA(a15);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int only_w = argc > 2 ? atoi(argv[2]) : 0, only_masked = argc > 3 ? atoi(argv[3]) : -1;  // one configuration (PMC runs)
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    printf("device %s, %d CUs, clock %.0f MHz, %d fma per wave\n", prop.gcnArchName, cus, prop.clockRate / 1000.0, iters * 32);
    for (int masked = 0; masked <= 32; masked += 16) {
        if (only_masked >= 0 && masked != only_masked) continue;
        for (int w = 1; w <= 8; w *= 2) {
            if (only_w && w != only_w) continue;
            const int blocks = simds * w;
            float *out;
            unsigned long long *cyc;
            hipMalloc(&out, blocks * 64 * sizeof(float));
            hipMalloc(&cyc, blocks * sizeof(unsigned long long));
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < (only_w ? 1 : 2); rep++) {
                hipEventRecord(e0);
                if (masked == 0) hipLaunchKernelGGL(k_fma<0>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
                else if (masked == 16) hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k_fma<32>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto v : h) mean += (double)v;
            mean /= blocks;
            const double insts = (double)iters * 32.0;
            // s_memtime ticks per wave-instruction, times waves per SIMD = SIMD time per wave-instruction
            printf("lanes %2d  waves/SIMD %d: %.3f ms, memtime ticks/wave %.0f -> %.3f ticks per instr per wave, %.3f per instr per SIMD; "
                   "event-time per instr per SIMD %.3f ns\n",
                   masked ? masked : 64, w, ms, mean, mean / insts, mean / insts / w, ms * 1e6 / (insts * w));
            hipFree(out);
            hipFree(cyc);
        }
    }
    return 0;
}
