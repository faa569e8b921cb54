/*
 * crt_libm.h -- cosf / sinf with the values glibc's libm returns, for arguments in [0, 120).
 *
 * The reference calls libm's cosf / sinf in mapSampleToDisk (sampling.cpp:606-612) and links the
 * platform's glibc.  glibc (>= 2.28) evaluates both in double precision -- quadrant reduction with a
 * pre-scaled 2/pi, a degree-8 / degree-7 polynomial pair -- and rounds once; the result is within
 * 0.56 ulp but not always the correctly rounded one (about 1 argument in 1000 differs), and the
 * device's own cosf / sinf differ far more often.  Because the reference offsets no ray, whether a
 * surface shadows itself hangs on the last bit of a hit point, so the sampled directions have to be
 * glibc's to the bit for images to agree.  This is that published algorithm restated (double
 * arithmetic only, every operation rounded separately; identical on host and device).
 * tests/test_libm.py compares it with the platform libm on every float in [0, 2*pi].
 */
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef CRT_HD
#ifdef __HIPCC__
#define CRT_HD __host__ __device__ __forceinline__
#else
#define CRT_HD inline
#endif
#endif

/* sign of the reduced argument per quadrant, and the polynomial coefficients; the second set (quadrants
 * 2 and 3) has the cosine coefficients negated */
CRT_HD float crt_libm_poly(double x, double x2, bool negate_cos, int n)
{
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double t1 = s2 + x2 * s3;
        const double x7 = x3 * x2;
        const double s = x + x3 * s1;
        return (float)(s + x7 * t1);
    }
    const double sg = negate_cos ? -1.0 : 1.0;
    const double x4 = x2 * x2;
    const double d2 = sg * c3 + x2 * (sg * c4);
    const double d1 = sg * c0 + x2 * (sg * c1);
    const double x6 = x4 * x2;
    const double c = d1 + x4 * (sg * c2);
    return (float)(c + x6 * d2);
}

CRT_HD uint32_t crt_libm_top12(float x)
{
#ifdef __HIP_DEVICE_COMPILE__
    return (__float_as_uint(x) >> 20) & 0x7ffu;
#else
    uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ffu;
#endif
}

/* which: 0 = sinf, 1 = cosf.  0 <= y < 120 (the callers pass 2*pi*u, u in [0, 1)). */
CRT_HD float crt_libm_sincosf(float y, int which)
{
    double x = (double)y;
    if (crt_libm_top12(y) < crt_libm_top12(0x1.921FB6p-1f)) {                 /* |y| < pi/4 */
        if (crt_libm_top12(y) < crt_libm_top12(0x1p-12f)) return which ? 1.0f : y;
        return crt_libm_poly(x, x * x, false, which);
    }
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;   /* 2/pi * 2^24, pi/2 */
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * hpi;
    const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return crt_libm_poly(x * sign, x * x, (n & 2) != 0, n ^ which);
}
CRT_HD float crt_libm_sinf(float y) { return crt_libm_sincosf(y, 0); }
CRT_HD float crt_libm_cosf(float y) { return crt_libm_sincosf(y, 1); }
