// forge3d_amd/csrc/f3d_composite.h -- the smoke-over-terrain composites of BASELINE.json configs[4], a pixel at a time.
// The reference does these on the host with numpy and Pillow (examples/california_cigar_smoke_demo.py):
//   atmospheric   composite_atmospheric_smoke(base, smoke_layer)          :8527-8544
//   smoke maps    composite_main_smoke_maps(atmospheric, physical)        :3367-3380  (_scale_rgba_alpha :8729-8732,
//                 _premultiplied_over :3352-3364)
//   over          Image.alpha_composite (Pillow 12.2.0, src/libImaging/AlphaComposite.c: 7 bits of coefficient
//                 precision, the >>8 +self >>8 division by 255) as composite_volume_detail :8721-8725 and
//                 _shift_rgba :3335-3349 use it
// Arithmetic is float32 in numpy's operation order with every operation spelled (no contraction); x^0.9 and e^x are
// the fixed polynomials of f3d_math.h, so host, emulator and device give the same bytes.  Against numpy itself the
// atmospheric mode can differ by one code value where its libm rounds a power the other way (the test bounds how often).
#pragma once

#include "f3d_math.h"

namespace f3d {
namespace composite {

struct Px {
    uint32_t r, g, b, a;
};
F3D_HD Px unpack(uint32_t v) { return Px{v & 255u, (v >> 8) & 255u, (v >> 16) & 255u, v >> 24}; }
F3D_HD uint32_t pack(Px p) { return p.r | (p.g << 8) | (p.b << 16) | (p.a << 24); }

// float -> u8 the way ndarray.astype(np.uint8) does for a value already clipped to [0, 255]: truncation
F3D_HD uint32_t trunc_u8(float v) { return (uint32_t)(int)f_clamp(v, 0.0f, 255.0f); }
// np.clip(np.round(v), 0, 255).astype(np.uint8): round half to even
F3D_HD uint32_t round_u8(float v) { return (uint32_t)(int)f_clamp(f_rint(v), 0.0f, 255.0f); }

// composite_atmospheric_smoke, :8527-8544: the smoke layer as an optical veil over the terrain frame
F3D_HD Px atmospheric(Px base, Px smoke) {
    const float alpha = (float)smoke.a / 255.0f;
    const float optical = pow_det(f_clamp(alpha * 0.98f, 0.0f, 1.0f), 0.90f);
    const float veil[3] = {(float)smoke.r / 255.0f, (float)smoke.g / 255.0f, (float)smoke.b / 255.0f};
    const float terrain[3] = {(float)base.r / 255.0f, (float)base.g / 255.0f, (float)base.b / 255.0f};
    const float warm = f_clamp((terrain[0] - terrain[2]) * 1.55f + (terrain[1] - terrain[2]) * 0.38f, 0.0f, 1.0f);
    // _smoothstep(0.10, 0.48, warm), :1602-1605: the denominator max(0.48 - 0.10, 1e-6) is evaluated in double (0.38)
    // and enters the float32 arithmetic as one constant
    const float tw = f_clamp((warm - 0.10f) / 0.38f, 0.0f, 1.0f);
    const float source_transmission = 1.0f - 0.34f * ((tw * tw) * (3.0f - 2.0f * tw));
    const float transmittance = exp_det((-0.72f * optical) * source_transmission);
    const float back[3] = {0.65f, 0.67f, 0.66f};
    const float back_w = 0.17f * optical;
    uint32_t out[3];
    for (int c = 0; c < 3; c++) {
        const float premul = veil[c] * optical;
        const float backscatter = back[c] * back_w;
        const float glow = ((terrain[c] * warm) * optical) * 0.18f;
        const float lifted = ((terrain[c] * transmittance + premul * 0.92f) + backscatter) + glow;
        out[c] = trunc_u8(lifted * 255.0f);
    }
    return Px{out[0], out[1], out[2], 255u};
}

F3D_HD Px scale_alpha(Px p, float scale) {  // _scale_rgba_alpha, :8729-8732
    p.a = trunc_u8((float)p.a * scale);
    return p;
}
// _premultiplied_over, :3352-3364 (max_alpha = HYBRID_SMOKE_MAX_ALPHA as a fraction, rounded to float32 from double)
F3D_HD Px premultiplied_over(Px bottom, Px top, float max_alpha_fraction) {
    const float ba = (float)bottom.a / 255.0f, ta = (float)top.a / 255.0f;
    const float one_minus = 1.0f - ta;
    float out_a = ta + ba * one_minus;
    const float b[3] = {(float)bottom.r / 255.0f, (float)bottom.g / 255.0f, (float)bottom.b / 255.0f};
    const float t[3] = {(float)top.r / 255.0f, (float)top.g / 255.0f, (float)top.b / 255.0f};
    uint32_t out[3];
    for (int c = 0; c < 3; c++) {
        const float premul = t[c] * ta + (b[c] * ba) * one_minus;
        const float rgb = out_a > 1.0e-6f ? premul / out_a : 0.0f;
        out[c] = round_u8(rgb * 255.0f);
    }
    if (max_alpha_fraction < out_a) out_a = max_alpha_fraction;
    return Px{out[0], out[1], out[2], round_u8(out_a * 255.0f)};
}
// composite_main_smoke_maps, :3367-3380; has_physical = 0 is the `physical_rgba is None` branch
F3D_HD Px smoke_maps(Px atmospheric_px, Px physical_px, bool has_physical, float atmospheric_alpha, float physical_alpha, uint32_t max_alpha,
                     float max_alpha_fraction) {
    const Px blanket = scale_alpha(atmospheric_px, atmospheric_alpha);
    if (!has_physical) return blanket;
    Px c = premultiplied_over(blanket, scale_alpha(physical_px, physical_alpha), max_alpha_fraction);
    if (c.a > max_alpha) c.a = max_alpha;
    return c;
}

// Image.alpha_composite for one pixel (Pillow AlphaComposite.c): integer coefficients with 7 extra bits
F3D_HD uint32_t div255_shift(uint32_t a) { return ((a >> 8) + a) >> 8; }
F3D_HD Px over(Px dst, Px src) {
    if (src.a == 0u) return dst;
    const uint32_t blend = dst.a * (255u - src.a);
    const uint32_t outa255 = src.a * 255u + blend;
    const uint32_t coef1 = (src.a * 255u * 255u * 128u) / outa255;
    const uint32_t coef2 = 255u * 128u - coef1;
    Px o;
    o.r = div255_shift(src.r * coef1 + dst.r * coef2 + (0x80u << 7)) >> 7;
    o.g = div255_shift(src.g * coef1 + dst.g * coef2 + (0x80u << 7)) >> 7;
    o.b = div255_shift(src.b * coef1 + dst.b * coef2 + (0x80u << 7)) >> 7;
    o.a = div255_shift(outa255 + 0x80u);
    return o;
}

struct Params {
    uint32_t mode, width, height;
    uint32_t layer_width, layer_height;
    int32_t offset_x, offset_y;
    uint32_t has_layer;
    float base_alpha, layer_alpha, max_alpha_fraction;
    uint32_t max_alpha;
};
enum Mode : uint32_t { kAtmospheric = 0u, kSmokeMaps = 1u, kOver = 2u };

// One output pixel.  The layer is addressed through the offset in every mode (0, 0 and equal sizes except for `over`);
// outside the layer the base passes through, which is what Pillow's crop-then-composite callers get.
F3D_HD uint32_t pixel(const Params &P, const uint32_t *base, const uint32_t *layer, uint32_t x, uint32_t y) {
    const Px b = unpack(base[(size_t)y * P.width + x]);
    const int64_t lx = (int64_t)x - P.offset_x, ly = (int64_t)y - P.offset_y;
    const bool inside = P.has_layer && lx >= 0 && ly >= 0 && lx < (int64_t)P.layer_width && ly < (int64_t)P.layer_height;
    const Px l = inside ? unpack(layer[(size_t)ly * P.layer_width + (size_t)lx]) : Px{0u, 0u, 0u, 0u};
    if (P.mode == kAtmospheric) return pack(atmospheric(b, l));
    if (P.mode == kSmokeMaps) return pack(smoke_maps(b, l, P.has_layer != 0u, P.base_alpha, P.layer_alpha, P.max_alpha, P.max_alpha_fraction));
    return pack(inside ? over(b, l) : b);
}

}  // namespace composite
}  // namespace f3d
