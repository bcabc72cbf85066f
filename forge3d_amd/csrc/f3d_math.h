// forge3d_amd/csrc/f3d_math.h
// Scalar / vec3 arithmetic shared by every kernel of the terrain path tracer.
//
// Numerics contract (DESIGN.md "Numerics"): IEEE f32, RNE, denormals kept, the whole
// library is compiled with -ffp-contract=off, and fused multiply-adds appear ONLY where
// this code spells fmaf().  Division and sqrt are the correctly rounded forms (hipcc's
// default).  sin/cos/atan2/acos are fixed polynomials, not the libm/ocml ones, so every
// device and host evaluates the same bits.  That makes the HIP path comparable
// bit-for-bit with the CPU oracle under tests/.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define F3D_HD __host__ __device__ __forceinline__
#define F3D_LAMBDA __attribute__((always_inline))
// "Forget" where a register value came from: code on a rarely taken branch recomputes what it needs from this
// value instead of keeping the hot path's temporaries alive across a memory wait (f3d_march.h, corner ties).
#if defined(__HIP_DEVICE_COMPILE__)
#define F3D_OPAQUE(x) asm volatile("" : "+v"(x))
// The same, said of a WAVE-UNIFORM value (a kernel parameter): whatever is computed from it stays where it is written.
// Loop-invariant code motion otherwise hoists every expression of render constants out of the sample loop -- also from
// branches that never run -- and, being float arithmetic, each lands in a VECTOR register that then lives in scratch for
// the whole kernel (64 lanes x 4 bytes per constant; a third of the frame kernel's scratch before round 4).  (A vector
// register on purpose: an "s" constraint is refused wherever the compiler has already moved the value to one.)
#define F3D_OPAQUE_UNIFORM(x) asm volatile("" : "+v"(x))
#else
#define F3D_OPAQUE(x) (void)(x)
#define F3D_OPAQUE_UNIFORM(x) (void)(x)
#endif
#else
#include <cmath>
#include <cstring>
#define F3D_HD inline
#define F3D_LAMBDA __attribute__((always_inline))
#define F3D_OPAQUE(x) (void)(x)
#define F3D_OPAQUE_UNIFORM(x) (void)(x)
// Host-only builds (tests/emul) have no HIP vector types.
struct alignas(16) float4 {
    float x, y, z, w;
};
struct alignas(8) uint2 {
    unsigned int x, y;
};
struct alignas(8) float2 {
    float x, y;
};
#endif

namespace f3d {

struct V3 {
    float x, y, z;
};

// Two independent f32 values in one register pair (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  A/B only
// (-DF3D_PACKED_F32): pairing the x- and z-plane arithmetic of the march gives the same results but measured
// 0.92x on MI355X -- the pairs need aligned registers and extra moves at the 80-VGPR budget (profiles/README.md).
// GCC (the host emulator) gets a plain struct.
#if defined(__clang__)
typedef float F2 __attribute__((ext_vector_type(2)));
F3D_HD F2 f2(float a, float b) { return F2{a, b}; }
F3D_HD F2 fma2(F2 a, F2 b, F2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct F2 {
    float x, y;
};
F3D_HD F2 f2(float a, float b) { return F2{a, b}; }
F3D_HD F2 operator-(F2 a, F2 b) { return F2{a.x - b.x, a.y - b.y}; }
F3D_HD F2 operator*(F2 a, F2 b) { return F2{a.x * b.x, a.y * b.y}; }
F3D_HD F2 fma2(F2 a, F2 b, F2 c) { return F2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
#endif

F3D_HD float f_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
F3D_HD float f_min(float a, float b) { return __builtin_fminf(a, b); }
F3D_HD float f_max(float a, float b) { return __builtin_fmaxf(a, b); }
F3D_HD float f_abs(float a) { return __builtin_fabsf(a); }
F3D_HD float f_sqrt(float a) { return __builtin_sqrtf(a); }
F3D_HD float f_rint(float a) { return __builtin_rintf(a); }
F3D_HD float f_floor(float a) { return __builtin_floorf(a); }
F3D_HD float f_clamp(float x, float lo, float hi) { return f_min(f_max(x, lo), hi); }

// a * b for factors below 2^24: v_mul_u32_u24 issues at full rate, v_mul_lo_u32 at a quarter of it
// (the masks tell the compiler what the callers guarantee)
F3D_HD uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
F3D_HD uint32_t f_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
F3D_HD float f_from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }
F3D_HD bool f_finite(float f) { return (f_bits(f) & 0x7F800000u) != 0x7F800000u; }

F3D_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
F3D_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
F3D_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
F3D_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
F3D_HD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
F3D_HD V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
F3D_HD float dot(V3 a, V3 b) { return f_fma(a.z, b.z, f_fma(a.y, b.y, a.x * b.x)); }
F3D_HD float dot2(float ax, float az, float bx, float bz) { return f_fma(az, bz, ax * bx); }
F3D_HD V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
F3D_HD V3 normalize(V3 a) {
    float inv = 1.0f / f_sqrt(dot(a, a));
    return a * inv;
}
// o + t * d
F3D_HD V3 along(V3 o, float t, V3 d) {
    return V3{f_fma(t, d.x, o.x), f_fma(t, d.y, o.y), f_fma(t, d.z, o.z)};
}
// a*x + b*y + c*z for three basis vectors
F3D_HD V3 combine(float a, V3 x, float b, V3 y, float c, V3 z) {
    return V3{f_fma(c, z.x, f_fma(b, y.x, a * x.x)), f_fma(c, z.y, f_fma(b, y.y, a * x.y)),
              f_fma(c, z.z, f_fma(b, y.z, a * x.z))};
}
// WGSL mix(a, b, t) = a * (1 - t) + b * t
F3D_HD float mix(float a, float b, float t) { return f_fma(b, t, a * (1.0f - t)); }
// Rec.709 luminance (reference hybrid_terrain_traversal.wgsl:416-418)
F3D_HD float luminance(V3 c) { return dot(c, V3{0.2126f, 0.7152f, 0.0722f}); }

// WGSL u32(f32): saturating conversion
F3D_HD uint32_t sat_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

// xorshift32 (reference hybrid_kernel.wgsl:78-85); returns the float draw in [0, 1].
F3D_HD float rng_next(uint32_t &s) {
    uint32_t x = s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    s = x;
    return (float)x / 4294967296.0f;
}

// advance the stream by `draws` draws without producing them
F3D_HD void rng_skip(uint32_t &s, uint32_t draws) {
    uint32_t x = s;
    for (uint32_t k = 0u; k < draws; k++) {
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
    }
    s = x;
}

constexpr float kPi = 3.14159265358979323846f;
constexpr float kHalfPi = 1.57079632679489661923f;
constexpr float kQuarterPi = 0.78539816339744830962f;

// ---- fixed-polynomial transcendentals (single-precision cephes coefficients) ----
F3D_HD float sin_quarter(float x) {  // |x| <= pi/4
    float z = x * x;
    float p = f_fma(f_fma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    return f_fma(p * z, x, x);
}
F3D_HD float cos_quarter(float x) {  // |x| <= pi/4
    float z = x * x;
    float p = f_fma(f_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    return f_fma(p, z * z, f_fma(-0.5f, z, 1.0f));
}
// sin, cos of 2*pi*u for u in [0, 1]
F3D_HD void sincos_turn(float u, float &s_out, float &c_out) {
#if defined(F3D_FAST_NUMERICS) && defined(__HIP_DEVICE_COMPILE__)
    // Tolerance tier (opt-in build, never the parity anchor -- DESIGN.md 5): v_sin_f32 / v_cos_f32 take their
    // argument in revolutions; the library is then also compiled with contraction on and 2.5-ulp divide / sqrt.
    s_out = __builtin_amdgcn_sinf(u);
    c_out = __builtin_amdgcn_cosf(u);
    return;
#endif
    float a = 4.0f * u;
    float k = f_rint(a);
    float x = (a - k) * kHalfPi;
    float s = sin_quarter(x), c = cos_quarter(x);
    int q = ((int)k) & 3;
    s_out = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    c_out = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}
F3D_HD float atan_det(float v) {
    float sign = 1.0f, x = v;
    if (v < 0.0f) {
        sign = -1.0f;
        x = -v;
    }
    float y;
    if (x > 2.414213562373095f) {
        y = kHalfPi;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = kQuarterPi;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    float p = f_fma(f_fma(f_fma(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z,
                    -3.33329491539e-1f);
    y = y + f_fma(p * z, x, x);
    return sign * y;
}
F3D_HD float atan2_det(float y, float x) {
    if (x > 0.0f) return atan_det(y / x);
    if (x < 0.0f) {
        float a = atan_det(y / x);
        return (y >= 0.0f) ? a + kPi : a - kPi;
    }
    if (y > 0.0f) return kHalfPi;
    if (y < 0.0f) return -kHalfPi;
    return 0.0f;
}
F3D_HD float asin_half(float x) {  // |x| <= 0.5
    float z = x * x;
    float p = f_fma(
        f_fma(f_fma(f_fma(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f),
        z, 1.6666752422e-1f);
    return f_fma(x * z, p, x);
}
F3D_HD float acos_det(float x) {  // x in [-1, 1]
    if (x < -0.5f) return kPi - 2.0f * asin_half(f_sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin_half(f_sqrt(0.5f * (1.0f - x)));
    return kHalfPi - asin_half(x);
}

// e^x with every operation spelled (cephes expf scheme: n = rint(x log2 e), two-constant reduction, degree-5
// polynomial, scaling through the exponent bits): < 2 ulp, and the same bits on host and device.
F3D_HD float exp_det(float x) {
    if (x > 88.0f) return __builtin_inff();
    if (x < -103.0f) return 0.0f;
    const float n = f_rint(x * 1.44269504088896341f);
    float r = f_fma(n, -0.693359375f, x);
    r = f_fma(n, 2.12194440e-4f, r);
    const float z = r * r;
    float p = f_fma(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = f_fma(p, r, 8.3334519073e-3f);
    p = f_fma(p, r, 4.1665795894e-2f);
    p = f_fma(p, r, 1.6666665459e-1f);
    p = f_fma(p, r, 5.0000001201e-1f);
    const float y = f_fma(p, z, r) + 1.0f;
    const int e = (int)n;
    if (e < -126) return (y * f_from_bits((uint32_t)(e + 64 + 127) << 23)) * 5.42101086242752217e-20f;  // 2^-64
    return y * f_from_bits((uint32_t)(e + 127) << 23);
}

// natural log of a positive normal float (cephes logf scheme), every operation spelled
F3D_HD float log_det(float x) {
    const uint32_t b = f_bits(x);
    int e = (int)(b >> 23) - 126;
    float m = f_from_bits((b & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float p = f_fma(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = f_fma(p, m, 1.1676998740e-1f);
    p = f_fma(p, m, -1.2420140846e-1f);
    p = f_fma(p, m, 1.4249322787e-1f);
    p = f_fma(p, m, -1.6668057665e-1f);
    p = f_fma(p, m, 2.0000714765e-1f);
    p = f_fma(p, m, -2.4999993993e-1f);
    p = f_fma(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    const float fe = (float)e;
    y = f_fma(-2.12194440e-4f, fe, y);
    y = f_fma(-0.5f, z, y);
    const float r = m + y;
    return f_fma(0.693359375f, fe, r);
}
F3D_HD float pow_det(float x, float y) {  // x >= 0, y > 0
    if (!(x > 0.0f)) return 0.0f;
    if (x < 1.17549435e-38f) return 0.0f;
    return exp_det(y * log_det(x));
}

// ---- IEEE binary16 storage rounding (RGBA16F targets of the reference) ----
F3D_HD uint16_t half_bits(float f) {
    uint32_t x = f_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mag = x & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (mag > 0x7F800000u ? 0x0200u : 0u));
    if (mag >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  // >= 65520 rounds to inf
    if (mag <= 0x33000000u) return (uint16_t)sign;              // <= 2^-25 rounds to zero
    int32_t e = (int32_t)(mag >> 23) - 127;
    uint32_t m = (mag & 0x007FFFFFu) | 0x00800000u;
    uint32_t drop = (e < -14) ? (uint32_t)(13 + (-14 - e)) : 13u;
    uint32_t q = m >> drop;
    uint32_t rem = m & ((1u << drop) - 1u);
    uint32_t halfway = 1u << (drop - 1u);
    if (e >= -14) q = ((uint32_t)(e + 15) << 10) | (q & 0x3FFu);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    return (uint16_t)(sign | q);
}
F3D_HD float half_value(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu;
    uint32_t m = h & 0x3FFu;
    if (e == 0u) {
        // subnormal or zero: value = m * 2^-24 (exact in f32)
        float v = (float)m * 5.9604644775390625e-8f;
        return f_from_bits(f_bits(v) | sign);
    }
    if (e == 31u) return f_from_bits(sign | 0x7F800000u | (m << 13));
    return f_from_bits(sign | ((e + 112u) << 23) | (m << 13));
}
F3D_HD float round_to_half(float v) { return half_value(half_bits(v)); }

}  // namespace f3d
