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

/* the series loops stay loops: unrolled inside the fused level kernel they cost it dozens of spilled registers */
#if defined(__clang__)
#define MM_NOUNROLL _Pragma("clang loop unroll(disable)")
#else
#define MM_NOUNROLL
#endif

/* sin and cos of r, |r| <= pi/4 + a little: Taylor series to r^19 / r^18 (truncation < 1e-19), Horner in r^2 from the
 * highest term down.  (These two are short and called four times per child: their loops may unroll; the longer series of
 * asin, log and exp stay loops over their tables, see MM_NOUNROLL.) */
MM_FN double mm_sin_poly(double r) {
    static const double c[9] = {
        1.0 / 6.0,                      /* 1/3! */
        -1.0 / 120.0,                   /* 1/5! */
        1.0 / 5040.0,                   /* 1/7! */
        -1.0 / 362880.0,                /* 1/9! */
        1.0 / 39916800.0,               /* 1/11! */
        -1.0 / 6227020800.0,            /* 1/13! */
        1.0 / 1307674368000.0,          /* 1/15! */
        -1.0 / 355687428096000.0,       /* 1/17! */
        1.0 / 121645100408832000.0};    /* 1/19! */
    const double z = r * r;
    double s = c[8];
    for (int i = 7; i >= 0; i--) s = s * z + c[i];          /* nine terms: unrolled by the compiler, constants in registers */
    return r - (r * z) * s;
}
MM_FN double mm_cos_poly(double r) {
    static const double c[9] = {
        0.5,                            /* 1/2! */
        -1.0 / 24.0,                    /* 1/4! */
        1.0 / 720.0,                    /* 1/6! */
        -1.0 / 40320.0,                 /* 1/8! */
        1.0 / 3628800.0,                /* 1/10! */
        -1.0 / 479001600.0,             /* 1/12! */
        1.0 / 87178291200.0,            /* 1/14! */
        -1.0 / 20922789888000.0,        /* 1/16! */
        1.0 / 6402373705728000.0};      /* 1/18! */
    const double z = r * r;
    double p = c[8];
    for (int i = 7; i >= 0; i--) p = p * z + c[i];
    return 1.0 - z * p;
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

/* asin(t), 0 <= t <= 1: the Taylor series asin x = x * sum c_k x^2k, c_k = (2k)! / (4^k k!^2 (2k+1)), to k = 24 for
 * x <= 0.5 (next term < 5e-18), and asin t = pi/2 - 2 asin sqrt((1-t)/2) above, whose argument is <= 0.5 again.  25 multiply-
 * adds instead of the four Newton steps on sin (each a sine, a cosine and a division) this function used to take; against
 * libm on 8 M arguments the result is within 4.4e-16, and its float rounding equals the Newton version's on every one. */
MM_FN double mm_asin_series(double x) {               /* 0 <= x <= 0.5 */
    static const double c[25] = {
        1.0, 0.16666666666666666, 0.074999999999999997, 0.044642857142857144, 0.030381944444444444,
        0.022372159090909092, 0.017352764423076924, 0.013964843750000001, 0.011551800896139705, 0.0097616095291940784,
        0.0083903358096168151, 0.0073125258735988454, 0.0064472103118896487, 0.0057400376708419236, 0.0051533096823199046,
        0.0046601434869150962, 0.0042409070936793632, 0.0038809645588376691, 0.0035692053938259347, 0.0032970595034734849,
        0.0030578216492580306, 0.0028461784011089421, 0.0026578706382072901, 0.0024894486782468836, 0.002338091892111975};
    const double z = x * x;
    double p = c[24];
    MM_NOUNROLL
    for (int i = 23; i >= 0; i--) p = p * z + c[i];
    return x * p;
}
MM_FN double mm_asin01(double t) {
    if (t > 0.5) {
        const double h = MM_SQRT((1.0 - t) * 0.5);          /* asin t = pi/2 - 2 asin sqrt((1-t)/2), argument <= 0.5 */
        return (MM_PI_2_HI - 2.0 * mm_asin_series(h)) + MM_PI_2_LO;
    }
    return mm_asin_series(t);
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
    static const double c[11] = {1.0 / 3.0, 1.0 / 5.0, 1.0 / 7.0, 1.0 / 9.0, 1.0 / 11.0, 1.0 / 13.0, 1.0 / 15.0, 1.0 / 17.0,
                                 1.0 / 19.0, 1.0 / 21.0, 1.0 / 23.0};
    double p = c[10];
    MM_NOUNROLL
    for (int i = 9; i >= 0; i--) p = p * z + c[i];
    const double lm = 2.0 * s + 2.0 * (s * z) * p;
    return (e * MM_LN2_HI + lm) + e * MM_LN2_LO;
}

/* exp(z), |z| < 700: z = k ln2 + r, |r| <= 0.35, Taylor to r^18, scaled by 2^k through the exponent field */
MM_FN double mm_exp(double z) {
    const double kf = z * (1.0 / 0.6931471805599453);
    const int k = (int)(kf < 0 ? kf - 0.5 : kf + 0.5);
    const double r = (z - k * MM_LN2_HI) - k * MM_LN2_LO;
    static const double c[19] = {
        1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0, 1.0 / 362880.0,
        1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0, 1.0 / 87178291200.0,
        1.0 / 1307674368000.0, 1.0 / 20922789888000.0, 1.0 / 355687428096000.0, 1.0 / 6402373705728000.0};   /* 1/k! */
    double p = c[18];
    MM_NOUNROLL
    for (int i = 17; i >= 0; i--) p = p * r + c[i];
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
