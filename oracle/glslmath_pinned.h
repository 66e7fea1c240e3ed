// glslmath_pinned.h - the checker's own statement of the GLSL built-in functions as Mesa 23.2 llvmpipe evaluates them.
// It is deliberately NOT the product's gpuraytracer_amd/csrc/rt_glslmath.h: an edit to that header must not move the
// checker along with the device.  Both are held to the same llvmpipe vectors independently: this file by
// tests/test_reference_glsl.py (ref_builtins.npz, on the CPU), the product's by the GPU parity tests against this checker.
//
// The GLSL specification leaves the precision of sin, cos, tan, pow, asin, acos and atan to the GL
// implementation.  The one implementation under which the reference's unmodified shader
// (res/shader/raytracer.glsl) can be executed without a GPU vendor driver is Mesa's llvmpipe, so its
// arithmetic is taken as the definition: every function below is the same fixed sequence of IEEE-754
// binary32 operations that Mesa emits, and returns the same bits (checked against llvmpipe on
// hundreds of thousands of arguments by tests/golden/make_glsl_reference_vectors.py).  Compiled with
// floating-point contraction disabled, so that only the fused multiply-adds written out here are
// fused, the sequence gives identical results on an x86 host and on a gfx950 device.
//
//   sin, cos          gallivm: |x|*4/pi -> octant j, three-step Cody-Waite reduction with FMA,
//                     degree-2 polynomials in z = r*r (raytracer.glsl:108-112, :535)
//   tan               NIR: sin(x) / cos(x)                                     (raytracer.glsl:535)
//   pow               gallivm: exp2(log2(x) * y), log2 by a degree-4 polynomial in ((m-1)/(m+1))^2,
//                     exp2 by a degree-5 polynomial of the fraction            (raytracer.glsl:497)
//   asin, acos        NIR: pi/2 - sqrt(1-|x|) * cubic(|x|)                     (raytracer.glsl:274, :527-528)
//   atan(y, x)        NIR: range reduction to [0,1], odd degree-11 polynomial  (raytracer.glsl:274)
#pragma once

#include <stdint.h>

#define GM_HD inline

namespace gm {

GM_HD float from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }
GM_HD uint32_t to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
GM_HD float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// IEEE minNum / maxNum: a NaN operand yields the other operand
GM_HD float fmin(float a, float b) { return __builtin_fminf(a, b); }
GM_HD float fmax(float a, float b) { return __builtin_fmaxf(a, b); }
GM_HD float fsign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// sin(x) and cos(x) share the reduction.
GM_HD void sincos(float a, float &s_out, float &c_out) {
    const float x_abs = __builtin_fabsf(a);
    const float scaled = x_abs * 1.27323954473516f;                 // 4/pi
    int32_t j = (int32_t)scaled;
    j = (j + 1) & ~1;
    const float y = (float)j;
    float x = fma(y, -0.78515625f, x_abs);
    x = fma(y, -2.4187564849853515625e-4f, x);
    x = fma(y, -3.77489497744594108e-8f, x);
    const float z = x * x;
    float pc = fma(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fma(pc, z, 4.166664568298827e-2f);
    pc = pc * z;
    pc = pc * z;
    pc = pc - z * 0.5f;
    pc = pc + 1.0f;
    float ps = fma(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fma(ps, z, -1.6666654611e-1f);
    ps = ps * z;
    ps = fma(ps, x, x);
    // sine: polynomial chosen by bit 1 of j, sign = sign(a) ^ bit 2 of j
    const uint32_t s_sign = (to_bits(a) & 0x80000000u) ^ (((uint32_t)j & 4u) << 29);
    const float s_mag = ((j & 2) == 0) ? ps : pc;
    s_out = from_bits(to_bits(s_mag) ^ s_sign);
    // cosine: the same with j - 2
    const int32_t jc = j - 2;
    const uint32_t c_sign = ((~(uint32_t)jc) & 4u) << 29;
    const float c_mag = ((jc & 2) == 0) ? ps : pc;
    c_out = from_bits(to_bits(c_mag) ^ c_sign);
}
GM_HD float sin(float x) { float s, c; sincos(x, s, c); return s; }
GM_HD float cos(float x) { float s, c; sincos(x, s, c); return c; }
GM_HD float tan(float x) { float s, c; sincos(x, s, c); return s / c; }

// log2(x) for finite x > 0
GM_HD float log2_pos(float x) {
    const uint32_t i = to_bits(x);
    const float logexp = (float)((int32_t)((i & 0x7f800000u) >> 23) - 127);
    const float mant = from_bits((i & 0x007fffffu) | 0x3f800000u);
    const float y = (mant - 1.0f) / (mant + 1.0f);
    const float z = y * y;
    const float z2 = z * z;
    // even and odd coefficients summed separately, as gallivm's polynomial helper does
    float even = fma(z2, 0.406718052498846252698f, 0.577440339438736392009f);
    even = fma(z2, even, 2.88539009343309178325f);
    const float odd = fma(z2, 0.403343858251329912514f, 0.961791550404184197881f);
    const float p = fma(odd, z, even);
    return fma(y, p, logexp);
}
GM_HD float exp2(float x) {
    x = (128.0f < x) ? 128.0f : x;                                    // NaN stays NaN
    x = (x < -126.99999f) ? -126.99999f : x;
    const float ipart = __builtin_floorf(x);
    const float fpart = x - ipart;
    const float scale = from_bits((uint32_t)(((int32_t)ipart + 127) << 23));
    const float f2 = fpart * fpart;
    float even = fma(f2, 0.00898934009049466391101f, 0.240153617044375388211f);
    even = fma(f2, even, 1.0f);
    float odd = fma(f2, 0.00187757667519147912699f, 0.0558263180532956664775f);
    odd = fma(f2, odd, 0.693153073200168932794f);
    return scale * fma(odd, fpart, even);
}
// pow(x, y): 0 for x == 0, NaN for x < 0
GM_HD float pow(float x, float y) {
    if (x == 0.0f) return 0.0f;
    if (!(x > 0.0f)) return from_bits(0x7fc00000u);
    if (x == __builtin_inff()) return exp2(x * y);
    return exp2(log2_pos(x) * y);
}

GM_HD float asin_magnitude(float ax, float p0, float p1) {
    const float PI_2 = from_bits(0x3fc90fdbu), PI_4_M1 = from_bits(0xbe5bc094u);     // pi/2, pi/4 - 1
    const float root = __builtin_sqrtf(1.0f - ax);
    const float poly = PI_2 + ax * (PI_4_M1 + ax * (p0 + ax * p1));
    return PI_2 - root * poly;
}
// arguments outside [-1, 1] give NaN
GM_HD float asin(float x) {
    return fsign(x) * asin_magnitude(__builtin_fabsf(x), from_bits(0x3db149e5u), from_bits(0xbcfe31afu));    // 0.086566724, -0.03102955
}
GM_HD float acos(float x) {
    const float PI_2 = from_bits(0x3fc90fdbu);
    return PI_2 - fsign(x) * asin_magnitude(__builtin_fabsf(x), from_bits(0x3da68d87u), from_bits(0xbcc19a5fu));   // 0.08132463, -0.02363318
}

GM_HD float atan2(float y, float x) {
    const float PI_2 = from_bits(0x3fc90fdbu);
    const bool flip = 0.0f >= x;
    const float ax = __builtin_fabsf(x);
    const float s = flip ? ax : y;
    const float c = flip ? y : ax;
    const float scale = (__builtin_fabsf(c) >= from_bits(0x5d5e0b6bu)) ? 0.25f : 1.0f;
    const float rcp_c = 1.0f / (c * scale);
    float t = __builtin_fabsf((s * scale) * rcp_c);
    t = (ax == __builtin_fabsf(y)) ? 1.0f : t;
    const float q = fmin(t, 1.0f) / fmax(t, 1.0f);
    const float q2 = q * q;
    const float q3 = q2 * q;
    const float q5 = q3 * q2;
    const float q7 = q5 * q2;
    const float q9 = q7 * q2;
    float sum = q * from_bits(0x3f7ffea5u) + q3 * from_bits(0xbeaa5476u);   //  0.99997932, -0.33267564
    sum = sum + q5 * from_bits(0x3e468bc1u);                           //  0.19389249
    sum = sum + q7 * from_bits(0xbdf0555du);                           // -0.11735032
    sum = sum + q9 * from_bits(0x3d5be101u);                           //  0.05368138
    sum = sum + (q9 * from_bits(0xbc46c6a5u)) * q2;                    // -0.01213232
    const float big = (1.0f < t) ? 1.0f : 0.0f;
    float r = big * (sum * -2.0f + PI_2) + sum;
    r = r * fsign(t);
    const float arc = (flip ? 1.0f : 0.0f) * PI_2 + r;
    return (fmin(y, rcp_c) < 0.0f) ? -arc : arc;
}

}  // namespace gm
