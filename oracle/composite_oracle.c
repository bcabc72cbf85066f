/* oracle/composite_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked or imported by the product).
 *
 * Restates the smoke-over-terrain composites of the reference's smoke sequence example, whole images at a time:
 *   composite_atmospheric_smoke   examples/california_cigar_smoke_demo.py:8527-8544  (_smoothstep :1602-1605)
 *   composite_main_smoke_maps     examples/california_cigar_smoke_demo.py:3367-3380  (_scale_rgba_alpha :8729-8732,
 *                                 _premultiplied_over :3352-3364, HYBRID_SMOKE_MAX_ALPHA :58)
 *   PIL.Image.alpha_composite     Pillow 12.2.0 (the reference example's dependency; not vendored in /root/reference):
 *                                 src/libImaging/AlphaComposite.c -- per pixel, with PRECISION_BITS = 7:
 *                                 outa255 = sa*255 + da*(255-sa); coef1 = sa*255*255*128 / outa255; coef2 = 255*128 - coef1;
 *                                 c = div255(sc*coef1 + dc*coef2 + (0x80 << 7)) >> 7; a = div255(outa255 + 0x80),
 *                                 div255(v) = ((v >> 8) + v) >> 8; a source pixel with alpha 0 leaves the destination.
 * PARITY PIN: tests/golden/smoke/composite_vectors.npz holds inputs and the outputs of the reference's own functions
 * (imported from /root/reference in the build container by tests/golden/make_composite_vectors.py, with numpy 2.2
 * and Pillow 12.2.0).  Smoke maps and alpha_composite are pinned bit for bit; the atmospheric composite within one
 * code value, because numpy takes x^0.9 and e^x from its libm/SIMD loops while this file (and the HIP kernel) use
 * the fixed polynomials below -- tests/test_composite.py states the bound and the measured mismatch rate.
 * Arithmetic: IEEE float32 in numpy's operation order, no contraction (-ffp-contract=off). */
#include <math.h>
#include <stdint.h>
#include <string.h>

static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
static float bits_f(uint32_t b) {
    float f;
    memcpy(&f, &b, 4);
    return f;
}
static uint32_t f_bits(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return b;
}
/* cephes expf / logf schemes with every operation spelled; the same constants as f3d_math.h (restated, not included) */
static float exp_fixed(float x) {
    if (x > 88.0f) return INFINITY;
    if (x < -103.0f) return 0.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    const float z = r * r;
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float y = fmaf(p, z, r) + 1.0f;
    const int e = (int)n;
    if (e < -126) return (y * bits_f((uint32_t)(e + 64 + 127) << 23)) * 5.42101086242752217e-20f;
    return y * bits_f((uint32_t)(e + 127) << 23);
}
static float log_fixed(float x) {
    const uint32_t b = f_bits(x);
    int e = (int)(b >> 23) - 126;
    float m = bits_f((b & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float p = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    const float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(0.693359375f, fe, m + y);
}
static float pow_fixed(float x, float y) {
    if (!(x > 0.0f) || x < 1.17549435e-38f) return 0.0f;
    return exp_fixed(y * log_fixed(x));
}
static uint8_t truncate_u8(float v) { return (uint8_t)(int)clampf(v, 0.0f, 255.0f); }         /* .astype(np.uint8) */
static uint8_t round_u8(float v) { return (uint8_t)(int)clampf(rintf(v), 0.0f, 255.0f); }      /* np.round: half to even */

/* composite_atmospheric_smoke, :8527-8544 */
void composite_oracle_atmospheric(const uint8_t *base, const uint8_t *smoke, uint32_t width, uint32_t height, uint8_t *out) {
    static const float backscatter_rgb[3] = {0.65f, 0.67f, 0.66f};
    for (size_t i = 0; i < (size_t)width * height; i++) {
        const uint8_t *b = base + 4 * i, *s = smoke + 4 * i;
        float alpha = (float)s[3] / 255.0f;
        float optical = pow_fixed(clampf(alpha * 0.98f, 0.0f, 1.0f), 0.90f);
        float terrain[3], veil[3];
        for (int c = 0; c < 3; c++) {
            veil[c] = (float)s[c] / 255.0f;
            terrain[c] = (float)b[c] / 255.0f;
        }
        float warm_signal = clampf((terrain[0] - terrain[2]) * 1.55f + (terrain[1] - terrain[2]) * 0.38f, 0.0f, 1.0f);
        float t = clampf((warm_signal - 0.10f) / (float)(0.48 - 0.10), 0.0f, 1.0f); /* _smoothstep(0.10, 0.48, .) */
        float smooth = t * t * (3.0f - 2.0f * t);
        float source_transmission = 1.0f - 0.34f * smooth;
        float transmittance = exp_fixed(-0.72f * optical * source_transmission);
        for (int c = 0; c < 3; c++) {
            float premul_smoke = veil[c] * optical;
            float backscatter = backscatter_rgb[c] * (0.17f * optical);
            float glow_through = terrain[c] * warm_signal * optical * 0.18f;
            float lifted = terrain[c] * transmittance + premul_smoke * 0.92f + backscatter + glow_through;
            out[4 * i + c] = truncate_u8(lifted * 255.0f);
        }
        out[4 * i + 3] = 255;
    }
}

/* composite_main_smoke_maps, :3367-3380; physical may be NULL */
void composite_oracle_smoke_maps(const uint8_t *atmospheric, const uint8_t *physical, uint32_t width, uint32_t height, float atmospheric_alpha,
                                 float physical_alpha, uint32_t max_alpha, uint8_t *out) {
    const float cap = (float)((double)max_alpha / 255.0);
    for (size_t i = 0; i < (size_t)width * height; i++) {
        uint8_t blanket[4], detail[4];
        memcpy(blanket, atmospheric + 4 * i, 4);
        blanket[3] = truncate_u8((float)blanket[3] * atmospheric_alpha);
        if (!physical) {
            memcpy(out + 4 * i, blanket, 4);
            continue;
        }
        memcpy(detail, physical + 4 * i, 4);
        detail[3] = truncate_u8((float)detail[3] * physical_alpha);
        /* _premultiplied_over(bottom = blanket, top = detail) */
        float bottom_a = (float)blanket[3] / 255.0f, top_a = (float)detail[3] / 255.0f;
        float out_a = top_a + bottom_a * (1.0f - top_a);
        for (int c = 0; c < 3; c++) {
            float top = (float)detail[c] / 255.0f, bottom = (float)blanket[c] / 255.0f;
            float premul = top * top_a + bottom * bottom_a * (1.0f - top_a);
            float rgb = out_a > 1.0e-6f ? premul / out_a : 0.0f;
            out[4 * i + c] = round_u8(rgb * 255.0f);
        }
        float capped = out_a < cap ? out_a : cap;
        uint8_t a8 = round_u8(capped * 255.0f);
        out[4 * i + 3] = a8 < max_alpha ? a8 : (uint8_t)max_alpha;
    }
}

/* PIL.Image.alpha_composite(base, layer pasted at (offset_x, offset_y)); pixels the layer does not cover keep the base */
void composite_oracle_over(const uint8_t *base, uint32_t width, uint32_t height, const uint8_t *layer, uint32_t layer_width, uint32_t layer_height,
                           int32_t offset_x, int32_t offset_y, uint8_t *out) {
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            const uint8_t *d = base + 4 * ((size_t)y * width + x);
            uint8_t *o = out + 4 * ((size_t)y * width + x);
            const int64_t lx = (int64_t)x - offset_x, ly = (int64_t)y - offset_y;
            if (lx < 0 || ly < 0 || lx >= (int64_t)layer_width || ly >= (int64_t)layer_height) {
                memcpy(o, d, 4);
                continue;
            }
            const uint8_t *s = layer + 4 * ((size_t)ly * layer_width + (size_t)lx);
            if (s[3] == 0) {
                memcpy(o, d, 4);
                continue;
            }
            const uint32_t blend = (uint32_t)d[3] * (255u - s[3]);
            const uint32_t outa255 = (uint32_t)s[3] * 255u + blend;
            const uint32_t coef1 = (uint32_t)s[3] * 255u * 255u * (1u << 7) / outa255;
            const uint32_t coef2 = 255u * (1u << 7) - coef1;
            for (int c = 0; c < 3; c++) {
                uint32_t v = s[c] * coef1 + d[c] * coef2 + (0x80u << 7);
                o[c] = (uint8_t)((((v >> 8) + v) >> 8) >> 7);
            }
            uint32_t a = outa255 + 0x80u;
            o[3] = (uint8_t)(((a >> 8) + a) >> 8);
        }
}
