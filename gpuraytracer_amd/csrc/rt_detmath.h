// Deterministic elementary functions for the path tracer.
//
// Every function here is a fixed sequence of IEEE-754 binary32/binary64 additions,
// subtractions, multiplications, divisions, correctly rounded float square roots,
// comparisons and integer conversions.  Compiled with floating-point contraction
// disabled (-ffp-contract=off) the sequence gives the same bits on an x86 host and
// on a gfx950 device, which is what lets a CPU checker compare rendered images with
// the device bit for bit.  Results are within one float ulp of the exact value.
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

constexpr double PIO2_HI = 1.57079632679489655800e+00;
constexpr double PIO2_LO = 6.12323399573676603587e-17;
constexpr double PI_D = 3.14159265358979311600e+00;
constexpr double PIO4_D = 7.85398163397448278999e-01;
constexpr double TWO_OVER_PI = 6.36619772367581382433e-01;

// floor() for |x| < 2^62 without calling a library routine.
DM_HD long long floor_ll(double x) {
    long long k = (long long)x;            // truncates toward zero
    if ((double)k > x) k -= 1;
    return k;
}

// sin and cos of r, |r| <= pi/4, Taylor polynomials evaluated by Horner's rule in double.
DM_HD double sin_kernel(double r) {
    const double z = r * r;
    double p = 1.0 / 6227020800.0;
    p = p * z + (-1.0 / 39916800.0);
    p = p * z + (1.0 / 362880.0);
    p = p * z + (-1.0 / 5040.0);
    p = p * z + (1.0 / 120.0);
    p = p * z + (-1.0 / 6.0);
    return r + (r * z) * p;
}

DM_HD double cos_kernel(double r) {
    const double z = r * r;
    double p = -1.0 / 87178291200.0;
    p = p * z + (1.0 / 479001600.0);
    p = p * z + (-1.0 / 3628800.0);
    p = p * z + (1.0 / 40320.0);
    p = p * z + (-1.0 / 720.0);
    p = p * z + (1.0 / 24.0);
    return (1.0 - 0.5 * z) + (z * z) * p;
}

// sin(x) and cos(x) in double for a float argument of moderate size (|x| < 1e6).
DM_HD void sincos_d(float xf, double &s, double &c) {
    const double x = (double)xf;
    const long long k = floor_ll(x * TWO_OVER_PI + 0.5);
    const double kd = (double)k;
    const double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    const double sr = sin_kernel(r);
    const double cr = cos_kernel(r);
    switch ((int)(k & 3)) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
}

DM_HD void sincos(float x, float &s, float &c) {
    double sd, cd;
    sincos_d(x, sd, cd);
    s = (float)sd;
    c = (float)cd;
}

DM_HD float sin(float x) {
    double sd, cd;
    sincos_d(x, sd, cd);
    return (float)sd;
}

DM_HD float cos(float x) {
    double sd, cd;
    sincos_d(x, sd, cd);
    return (float)cd;
}

DM_HD float tan(float x) {
    double sd, cd;
    sincos_d(x, sd, cd);
    return (float)(sd / cd);
}

// Square root of a double in [0, 1]: float sqrt (correctly rounded) plus one Newton step.
DM_HD double sqrt01_d(double z) {
    const float s0f = __builtin_sqrtf((float)z);
    if (s0f == 0.0f) return 0.0;
    const double s0 = (double)s0f;
    return s0 + (z - s0 * s0) / (2.0 * s0);
}

// x * R(x^2) term of asin on |x| <= 0.5 (rational approximation, fdlibm coefficients).
DM_HD double asin_r(double z) {
    const double p = z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 +
                     z * (2.01212532134862925881e-01 + z * (-4.00555345006794114027e-02 +
                     z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
    const double q = 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 +
                     z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
    return p / q;
}

// asin for x in [-1, 1]; arguments outside are clamped to the interval.
DM_HD double asin_d(float xf) {
    double x = (double)xf;
    if (x > 1.0) x = 1.0;
    if (x < -1.0) x = -1.0;
    const double ax = x < 0.0 ? -x : x;
    if (ax < 0.5) return x + x * asin_r(x * x);
    const double z = (1.0 - ax) * 0.5;
    const double s = sqrt01_d(z);
    const double r = (PIO2_HI - 2.0 * (s + s * asin_r(z))) + PIO2_LO;
    return x < 0.0 ? -r : r;
}

DM_HD float asin(float x) { return (float)asin_d(x); }

// acos for x in [-1, 1]; arguments outside are clamped to the interval.
DM_HD float acos(float xf) {
    double x = (double)xf;
    if (x > 1.0) x = 1.0;
    if (x < -1.0) x = -1.0;
    const double ax = x < 0.0 ? -x : x;
    if (ax < 0.5) return (float)((PIO2_HI - (x + x * asin_r(x * x))) + PIO2_LO);
    const double z = (1.0 - ax) * 0.5;
    const double s = sqrt01_d(z);
    const double w = 2.0 * (s + s * asin_r(z));
    return (float)(x < 0.0 ? (PI_D - w) : w);
}

// atan(a) for a in [0, 1].
DM_HD double atan01_d(double a) {
    double base = 0.0;
    double t = a;
    if (a > 0.41421356237309503) {
        t = (a - 1.0) / (a + 1.0);
        base = PIO4_D;
    }
    const double z = t * t;
    // alternating series atan(t) = t + t*z*(-1/3 + z*(1/5 + z*(-1/7 + ...))), 20 terms
    double p = 0.0;
    for (int n = 20; n >= 1; --n) {
        const double coef = 1.0 / (double)(2 * n + 1);
        p = ((n & 1) ? -coef : coef) + z * p;
    }
    return base + (t + (t * z) * p);
}

// atan2(y, x) with the usual quadrant rules; atan2(0, 0) = 0.
DM_HD float atan2(float yf, float xf) {
    const double x = (double)xf, y = (double)yf;
    const double ax = x < 0.0 ? -x : x;
    const double ay = y < 0.0 ? -y : y;
    const double hi = ax > ay ? ax : ay;
    const double lo = ax > ay ? ay : ax;
    if (hi == 0.0) return 0.0f;
    double r = atan01_d(lo / hi);
    if (ay > ax) r = (PIO2_HI - r) + PIO2_LO;
    if (x < 0.0) r = PI_D - r;
    if (y < 0.0) r = -r;
    return (float)r;
}

}  // namespace dm
