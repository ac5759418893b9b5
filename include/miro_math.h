/*
 * miro_math.h -- the transcendental functions of the reference's path-tracing ray generators (Ray::random, and the
 * PATH_TRACING branches of Ray::reflect / Ray::refract, Ray.h:124-158,235-239; alignHemisphereToVector, Utility.h:34-50)
 * as ONE deterministic implementation for every side that has to agree bit for bit.
 *
 * The reference calls libm's float functions (asinf, acosf, sinf, cosf, powf through <cmath>); their last bit is not
 * specified, glibc and the ROCm device library round some arguments differently, and a ray set that is to be compared
 * bit for bit between the device and its CPU checker cannot depend on either.  Here every function is evaluated in
 * IEEE double arithmetic with fixed-length series (add, multiply, divide and square root only -- each correctly
 * rounded on the host and on gfx950, compiled with -ffp-contract=off on both) and rounded to float once: the float
 * result is the correctly rounded one except when the exact value lies within ~1e-15 relative of a rounding boundary,
 * i.e. it equals a correctly rounded libm float function on all but isolated arguments (tests/test_path_rays.py::
 * test_miro_math_matches_correctly_rounded_double compares 400 k arguments per function with the rounded double result).
 *
 * Plain C99 / C++ / HIP: `static inline`, no libm call, no state.
 */
#ifndef MIRO_MATH_H
#define MIRO_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MM_FN __host__ __device__ static inline
#else
#define MM_FN static inline
#endif

/* sqrt is the one primitive taken from the platform: correctly rounded by IEEE 754 on both sides */
#if defined(__HIPCC__)
#define MM_SQRT(x) __builtin_sqrt(x)
#else
#define MM_SQRT(x) __builtin_sqrt(x)
#endif

#define MM_PI_2_HI 1.57079632679489655800e+00 /* pi/2 rounded to double */
#define MM_PI_2_LO 6.12323399573676603587e-17 /* pi/2 - MM_PI_2_HI */
#define MM_LN2_HI 6.93147180369123816490e-01  /* ln 2 with the low 21 bits cleared */
#define MM_LN2_LO 1.90821492927058770002e-10  /* ln 2 - MM_LN2_HI */

/* the Newton loops stay loops: unrolled inside the fused level kernel they cost it ~60 more spilled registers */
#if defined(__clang__)
#define MM_NOUNROLL _Pragma("clang loop unroll(disable)")
#else
#define MM_NOUNROLL
#endif

/* sin and cos of r, |r| <= pi/4 + a little: Taylor series to r^19 / r^18 (truncation < 1e-19) */
MM_FN double mm_sin_poly(double r) {
    const double z = r * r;
    double s = 1.0 / 121645100408832000.0;                 /* 1/19! */
    s = s * z - 1.0 / 355687428096000.0;                   /* 1/17! */
    s = s * z + 1.0 / 1307674368000.0;                     /* 1/15! */
    s = s * z - 1.0 / 6227020800.0;                        /* 1/13! */
    s = s * z + 1.0 / 39916800.0;                          /* 1/11! */
    s = s * z - 1.0 / 362880.0;                            /* 1/9! */
    s = s * z + 1.0 / 5040.0;                              /* 1/7! */
    s = s * z - 1.0 / 120.0;                               /* 1/5! */
    s = s * z + 1.0 / 6.0;                                 /* 1/3! */
    return r - (r * z) * s;
}
MM_FN double mm_cos_poly(double r) {
    const double z = r * r;
    double c = 1.0 / 6402373705728000.0;                   /* 1/18! */
    c = c * z - 1.0 / 20922789888000.0;                    /* 1/16! */
    c = c * z + 1.0 / 87178291200.0;                       /* 1/14! */
    c = c * z - 1.0 / 479001600.0;                         /* 1/12! */
    c = c * z + 1.0 / 3628800.0;                           /* 1/10! */
    c = c * z - 1.0 / 40320.0;                             /* 1/8! */
    c = c * z + 1.0 / 720.0;                               /* 1/6! */
    c = c * z - 1.0 / 24.0;                                /* 1/4! */
    c = c * z + 0.5;                                       /* 1/2! */
    return 1.0 - z * c;
}

/* sin(x), cos(x) for |x| <= 16 (the callers pass [0, 2 pi] and [0, pi/2]) */
MM_FN void mm_sincos(double x, double *s, double *c) {
    const double kf = x * (1.0 / MM_PI_2_HI);
    const int k = (int)(kf < 0 ? kf - 0.5 : kf + 0.5);      /* nearest quadrant */
    const double r = (x - k * MM_PI_2_HI) - k * MM_PI_2_LO;
    const double sr = mm_sin_poly(r), cr = mm_cos_poly(r);
    switch (k & 3) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}

/* asin(t), 0 <= t <= 1: Newton's iteration on sin from a cubic guess for t <= 0.75, the half-angle identity above.
 * The guess is off by < 0.029 (t <= 0.75) / < 4e-4 (h <= 0.354) and every step squares the error (times tan p / 2 < 0.6):
 * four / three steps reach the last bit of a double (measured against libm on 4 M arguments: 3.3e-16; one step fewer:
 * 5.6e-15 / 7.8e-16), and the float results equal those of six / five steps on every one of them. */
MM_FN double mm_asin01(double t) {
    if (t > 0.75) {
        const double h = MM_SQRT((1.0 - t) * 0.5);          /* asin t = pi/2 - 2 asin sqrt((1-t)/2), argument <= 0.354 */
        double p = h + (h * h * h) * (1.0 / 6.0);
        MM_NOUNROLL
        for (int i = 0; i < 3; i++) {
            double s, c;
            mm_sincos(p, &s, &c);
            p = p - (s - h) / c;
        }
        return (MM_PI_2_HI - 2.0 * p) + MM_PI_2_LO;
    }
    double p = t + (t * t * t) * (1.0 / 6.0);
    MM_NOUNROLL
    for (int i = 0; i < 4; i++) {
        double s, c;
        mm_sincos(p, &s, &c);
        p = p - (s - t) / c;
    }
    return p;
}
MM_FN double mm_acos01(double t) {                          /* acos(t), 0 <= t <= 1 */
    if (t > 0.5) {
        const double h = MM_SQRT((1.0 - t) * 0.5);          /* acos t = 2 asin sqrt((1-t)/2) */
        return 2.0 * mm_asin01(h);
    }
    return (MM_PI_2_HI - mm_asin01(t)) + MM_PI_2_LO;
}

/* ln(x), x > 0 finite and normal: x = m 2^e with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m-1)/(m+1)) */
MM_FN double mm_log(double x) {
    uint64_t b;
    memcpy(&b, &x, 8);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    memcpy(&m, &b, 8);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;     /* |s| <= 0.1716 */
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    const double lm = 2.0 * s + 2.0 * (s * z) * p;
    return (e * MM_LN2_HI + lm) + e * MM_LN2_LO;
}

/* exp(z), |z| < 700: z = k ln2 + r, |r| <= 0.35, Taylor to r^18, scaled by 2^k through the exponent field */
MM_FN double mm_exp(double z) {
    const double kf = z * (1.0 / 0.6931471805599453);
    const int k = (int)(kf < 0 ? kf - 0.5 : kf + 0.5);
    const double r = (z - k * MM_LN2_HI) - k * MM_LN2_LO;
    double p = 1.0 / 6402373705728000.0;                   /* 1/18! */
    p = p * r + 1.0 / 355687428096000.0;
    p = p * r + 1.0 / 20922789888000.0;
    p = p * r + 1.0 / 1307674368000.0;
    p = p * r + 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const uint64_t sb = (uint64_t)(k + 1023) << 52;         /* 2^k, k in the normal range for the callers' arguments */
    double scale;
    memcpy(&scale, &sb, 8);
    return p * scale;
}

/* ---- the float functions the ray generators use (float in, float out, rounded once) ---------------------------- */
MM_FN float mm_sinf(float x) { double s, c; mm_sincos((double)x, &s, &c); return (float)s; }
MM_FN float mm_cosf(float x) { double s, c; mm_sincos((double)x, &s, &c); return (float)c; }
MM_FN float mm_asinf01(float t) { return (float)mm_asin01((double)t); }        /* 0 <= t <= 1 */
MM_FN float mm_acosf01(float t) { return (float)mm_acos01((double)t); }        /* 0 <= t <= 1 */
/* acosf on [-1, 1] (NaN outside, as libm): acos(-t) = pi - acos(t) */
MM_FN float mm_acosf(float t) {
    if (!(t >= -1.0f && t <= 1.0f)) { uint32_t q = 0x7fc00000u; float f; memcpy(&f, &q, 4); return f; }
    if (t >= 0.0f) return (float)mm_acos01((double)t);
    return (float)((2.0 * MM_PI_2_HI - mm_acos01(-(double)t)) + 2.0 * MM_PI_2_LO);
}
/* powf(x, y) for 0 <= x <= 1, 0 <= y <= 1 (x = frand(), y = 1 / (1 + shininess)); powf(x, 0) = 1, powf(0, y > 0) = 0 */
MM_FN float mm_powf01(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return 0.0f;
    return (float)mm_exp((double)y * mm_log((double)x));
}

#endif /* MIRO_MATH_H */
