// forge3d_amd/csrc/f3d_div_known.h -- the IEEE quotient a / d for a divisor known before the launch, in three instructions.
//
// With r = RN(1 / d): q0 = RN(a r) is within an ulp of the quotient, the remainder a - q0 d is exact in one fma, and
// RN(q0 + rem r) is the correctly rounded a / d -- what the reference's `/` and the oracle's produce (Markstein 1990; Muller
// et al., Handbook of Floating-Point Arithmetic, section 4.7: true when r is the correctly rounded reciprocal and d's significand
// is not all ones, as long as nothing over- or underflows on the way).  The division's own expansion on gfx950 is eleven
// instructions (two v_div_scale, v_rcp, five fma, v_div_fmas, v_div_fixup); the smoke marcher's loops divide five times a step
// and their waves are bound by how fast they issue vector instructions (csrc/f3d_smoke.hip).
//
// Conditions, kept by the callers: div_known_divisor(d) on the host (or in a static_assert); |a| < kDivKnownMax, anything
// larger takes the plain division.  For |a| < 2^-60 the result may be an ulp off, which the two uses cannot show: a tap
// coordinate is the quotient minus 0.5, and a smoothstep of an argument below 2^-14 moves nothing it is multiplied into or
// added to.  tests/test_smoke.py checks the identity against the division for every significand (tests/emul).
#pragma once
#include <cstdint>
#include <cstring>

#include "f3d_math.h"

namespace f3d {

constexpr float kDivKnownMax = 0x1p60f;

F3D_HD float div_known(float a, float d, float r) {
    const float q0 = a * r;
    return __builtin_fmaf(__builtin_fmaf(-q0, d, a), r, q0);
}

inline bool div_known_divisor(float d) {  // host side: may div_known() stand for `/ d`?
    uint32_t bits;
    memcpy(&bits, &d, sizeof(bits));
    return d > 0x1p-40f && d < 0x1p40f && (bits & 0x7FFFFFu) != 0x7FFFFFu;
}

}  // namespace f3d
