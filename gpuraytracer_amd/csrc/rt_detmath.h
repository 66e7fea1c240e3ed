// Deterministic elementary functions for the path tracer.
//
// Every function here is a fixed sequence of IEEE-754 binary32 additions, subtractions,
// multiplications, divisions, correctly rounded square roots, floor, comparisons and
// integer conversions.  Compiled with floating-point contraction disabled
// (-ffp-contract=off) the sequence gives the same bits on an x86 host and on a gfx950
// device, which is what lets a CPU checker compare rendered images with the device bit
// for bit.  Results are within about 2 float ulps of the exact value on the ranges the
// shader uses (tan: 4 ulps): polynomial kernels with the classic single-precision
// coefficients, Cody-Waite argument reduction.
//
// The GLSL built-ins these stand in for (sin, cos, tan, asin, acos, atan) are used by
// the reference shader at res/shader/raytracer.glsl:108-112 (randomSphere),
// :274 (skyColor) and :527-535 (Oren-Nayar); GLSL leaves their precision to the driver.
#pragma once

#if defined(__HIPCC__)
#define DM_HD __host__ __device__ inline
#else
#define DM_HD inline
#endif

namespace dm {

constexpr float TWO_OVER_PI_F = 0.63661977236758134308f;
// pi/2 split in three parts whose products with small integers are exact (Cody-Waite)
constexpr float PIO2_1 = 1.5703125f;
constexpr float PIO2_2 = 4.837512969970703125e-4f;
constexpr float PIO2_3 = 7.54978995489188216e-8f;
constexpr float PIO2_HI = 1.5707962513e+00f;
constexpr float PIO2_LO = 7.5497894159e-08f;
constexpr float PI_F = 3.14159274101257324f;
constexpr float PIO4_F = 0.785398185253143310f;

// sin(r), cos(r) for |r| <= pi/4
DM_HD float sin_kernel(float r) {
    const float z = r * r;
    const float p = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z;
    return r + r * p;
}
DM_HD float cos_kernel(float r) {
    const float z = r * r;
    const float p = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * (z * z);
    return (1.0f - 0.5f * z) + p;
}

// sin(x) and cos(x); accurate for |x| up to a few hundred.
DM_HD void sincos(float x, float &s, float &c) {
    const float kf = __builtin_floorf(x * TWO_OVER_PI_F + 0.5f);
    const float r = ((x - kf * PIO2_1) - kf * PIO2_2) - kf * PIO2_3;
    const float sr = sin_kernel(r);
    const float cr = cos_kernel(r);
    const int q = (int)kf & 3;
    const float s0 = (q & 1) ? cr : sr;
    const float c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

DM_HD float sin(float x) {
    float s, c;
    sincos(x, s, c);
    return s;
}
DM_HD float cos(float x) {
    float s, c;
    sincos(x, s, c);
    return c;
}
DM_HD float tan(float x) {
    float s, c;
    sincos(x, s, c);
    return s / c;
}

// Rational term of asin/acos: asin(x) = x + x*R(x^2) on |x| <= 0.5
DM_HD float asin_r(float z) {
    const float p = z * (1.6666586697e-01f + z * (-4.2743422091e-02f + z * -8.6563630030e-03f));
    const float q = 1.0f + z * -7.0662963390e-01f;
    return p / q;
}

// acos for x in [-1, 1]; arguments outside are clamped to the interval.
DM_HD float acos(float x) {
    if (x > 1.0f) x = 1.0f;
    if (x < -1.0f) x = -1.0f;
    if (x > -0.5f && x < 0.5f) return PIO2_HI - (x - (PIO2_LO - x * asin_r(x * x)));
    if (x <= -0.5f) {
        const float z = (1.0f + x) * 0.5f;
        const float s = __builtin_sqrtf(z);
        const float w = asin_r(z) * s - PIO2_LO;
        return 2.0f * (PIO2_HI - (s + w));
    }
    const float z = (1.0f - x) * 0.5f;
    const float s = __builtin_sqrtf(z);
    if (s == 0.0f) return 0.0f;
    // df = s with the low 12 mantissa bits cleared, so that df*df is exact
    unsigned int bits = __builtin_bit_cast(unsigned int, s) & 0xFFFFF000u;
    const float df = __builtin_bit_cast(float, bits);
    const float cc = (z - df * df) / (s + df);
    const float w = asin_r(z) * s + cc;
    return 2.0f * (df + w);
}

// asin for x in [-1, 1]; arguments outside are clamped to the interval.
DM_HD float asin(float x) {
    if (x > 1.0f) x = 1.0f;
    if (x < -1.0f) x = -1.0f;
    const float ax = x < 0.0f ? -x : x;
    if (ax < 0.5f) return x + x * asin_r(x * x);
    const float z = (1.0f - ax) * 0.5f;
    const float s = __builtin_sqrtf(z);
    float r;
    if (s == 0.0f) {
        r = PIO2_HI + PIO2_LO;
    } else {
        unsigned int bits = __builtin_bit_cast(unsigned int, s) & 0xFFFFF000u;
        const float df = __builtin_bit_cast(float, bits);
        const float cc = (z - df * df) / (s + df);
        const float w = asin_r(z) * s + cc;
        r = PIO2_HI - (2.0f * (df + w) - PIO2_LO);
    }
    return x < 0.0f ? -r : r;
}

// atan(a) for a in [0, 1]
DM_HD float atan01(float a) {
    float base = 0.0f, t = a;
    if (a > 0.4142135623730950f) {
        t = (a - 1.0f) / (a + 1.0f);
        base = PIO4_F;
    }
    const float z = t * t;
    const float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z;
    return base + (t + t * p);
}

// atan2(y, x) with the usual quadrant rules; atan2(0, 0) = 0.
DM_HD float atan2(float y, float x) {
    const float ax = x < 0.0f ? -x : x;
    const float ay = y < 0.0f ? -y : y;
    const float hi = ax > ay ? ax : ay;
    const float lo = ax > ay ? ay : ax;
    if (hi == 0.0f) return 0.0f;
    float r = atan01(lo / hi);
    if (ay > ax) r = PIO2_HI - (r - PIO2_LO);
    if (x < 0.0f) r = PI_F - r;
    if (y < 0.0f) r = -r;
    return r;
}

}  // namespace dm
