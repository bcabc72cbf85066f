// tools/experiments/issue_calib.hip -- how fast does ONE wave issue instructions on gfx950, by type?
//
// k_frame's march step is ~70 VALU + ~74 SALU (exec-mask bookkeeping of its divergent branches) + 1 VMEM + 3 LDS.
// valu_calib.hip showed that a single wave issues one independent v_fma every 5.5 cycles, so with 6 waves per SIMD the
// per-wave in-order latency, not the VALU pipe, can be the bound.  Does a SALU instruction cost the wave the same issue
// slot?  Do VALU and SALU of ONE wave overlap?  Kernels: pure v_fma, pure s_add_u32 (independent registers), strictly
// alternating v_fma / s_add, and v_fma with an s_cbranch-free s_and_b64 exec-style pair per VALU; W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define VF(a) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y))
#define SA(s) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(k) : "scc")
#define SM(s) asm volatile("s_and_b64 %0, %0, %1" : "+s"(s) : "s"(m) : "scc")

template <int MODE>
__global__ __launch_bounds__(64) void k_issue(float *out, unsigned long long *cycles, int iters, float x, float y, unsigned k,
                                              unsigned long long m) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    unsigned s0 = 0, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    unsigned long long m0 = ~0ull, m1 = ~0ull, m2 = ~0ull, m3 = ~0ull;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 16 VALU
            VF(a0); VF(a1); VF(a2); VF(a3); VF(a4); VF(a5); VF(a6); VF(a7); VF(a0); VF(a1); VF(a2); VF(a3); VF(a4); VF(a5); VF(a6); VF(a7);
        } else if (MODE == 1) {  // 16 SALU
            SA(s0); SA(s1); SA(s2); SA(s3); SA(s4); SA(s5); SA(s6); SA(s7); SA(s0); SA(s1); SA(s2); SA(s3); SA(s4); SA(s5); SA(s6); SA(s7);
        } else if (MODE == 2) {  // 8 VALU + 8 SALU, alternating
            VF(a0); SA(s0); VF(a1); SA(s1); VF(a2); SA(s2); VF(a3); SA(s3); VF(a4); SA(s4); VF(a5); SA(s5); VF(a6); SA(s6); VF(a7); SA(s7);
        } else {  // 8 VALU + 8 64-bit mask ops (the exec bookkeeping of a divergent branch)
            VF(a0); SM(m0); VF(a1); SM(m1); VF(a2); SM(m2); VF(a3); SM(m3); VF(a4); SM(m0); VF(a5); SM(m1); VF(a6); SM(m2); VF(a7); SM(m3);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) + (float)(m0 & m1 & m2 & m3);
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int simds, int iters) {
    for (int w = 1; w <= 8; w *= 2) {
        const int blocks = simds * w;
        float *out;
        unsigned long long *cyc;
        hipMalloc(&out, blocks * 64 * sizeof(float));
        hipMalloc(&cyc, blocks * sizeof(unsigned long long));
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_issue<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0001f, 0.5f, 3u, ~0ull);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= blocks;
        printf("%-28s waves/SIMD %d: %.2f cycles per instruction per wave (16 per iteration)\n", name, w, mean / (16.0 * iters));
        hipFree(out);
        hipFree(cyc);
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int simds = prop.multiProcessorCount * 4;
    run<0>("16 v_fma", simds, iters);
    run<1>("16 s_add_u32", simds, iters);
    run<2>("8 v_fma + 8 s_add alternating", simds, iters);
    run<3>("8 v_fma + 8 s_and_b64", simds, iters);
    return 0;
}
