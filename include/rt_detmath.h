/* rt_detmath.h — the numeric convention for HLSL intrinsics that have no IEEE-exact definition.
 *
 * The reference's kernels call sin, cos, log, pow, acos, asin and atan2
 * (BVH_ReSTIR_DL/Assets/Scripts/ComputeShaders/RendererResources.hlsl:93-123, BRDF.hlsl:3,
 * ReSTIR_BVH_RayTracer.compute:107-108).  A GPU driver implements those with vendor-specific
 * approximations, so no bit pattern of theirs can be reproduced.  This header fixes ONE definition for each:
 * single-precision polynomial evaluations built only from IEEE-754 add, subtract, multiply, divide and
 * square root (no fused multiply-add, no libm), after the classic Cephes single-precision routines.  Compiled
 * with -ffp-contract=off they return identical bits under g++ on x86-64 and under hipcc on gfx950, which is
 * what lets CPU and GPU results be compared bit for bit.  Each is within a few ulp of the correctly rounded
 * value (checked against libm in tests/test_detmath.py).
 *
 * It also fixes the float<->integer conversions, which HLSL defines (NaN -> 0, saturating) and C++ does not.
 */
#ifndef RT_DETMATH_H
#define RT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD static inline
#endif

namespace rtdm {

RT_HD uint32_t f2bits(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
RT_HD float bits2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
RT_HD float inf() { return bits2f(0x7f800000u); }
RT_HD float qnan() { return bits2f(0x7fc00000u); }
RT_HD bool isnan_(float x) { return x != x; }

/* HLSL (int)f: truncate toward zero, NaN -> 0, saturate at the int range. */
RT_HD int32_t f2i(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int32_t)0x80000000u;
    return (int32_t)f;
}
/* HLSL (uint)f: truncate, NaN and negatives -> 0, saturate at 2^32-1. */
RT_HD uint32_t f2u(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

RT_HD float fabs_(float x) { return bits2f(f2bits(x) & 0x7fffffffu); }

/* min/max as D3D defines them: if one operand is NaN the other is returned. */
RT_HD float fmin_(float a, float b) { if (a != a) return b; if (b != b) return a; return b < a ? b : a; }
RT_HD float fmax_(float a, float b) { if (a != a) return b; if (b != b) return a; return a < b ? b : a; }

#define RT_PI_F 3.1415926f /* Constants.hlsl:3 (the reference's own truncated pi) */

/* ---- sin / cos: Cephes sinf/cosf. Octant reduction with a three-part pi/4, valid for |x| < 8192. ---- */
RT_HD void sincos_reduce(float x, float* r, int* oct) {
    const float FOPI = 1.27323954473516f; /* 4/pi */
    const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    int j = (int)(FOPI * x); /* x >= 0 here */
    if (j & 1) j += 1;
    float y = (float)j;
    *oct = j & 7;
    *r = ((x - y * DP1) - y * DP2) - y * DP3;
}
RT_HD float sin_poly(float r) {
    float z = r * r;
    return ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * r + r;
}
RT_HD float cos_poly(float r) {
    float z = r * r;
    float y = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}
RT_HD float sin_(float x) {
    if (x != x || fabs_(x) >= 8192.0f) return (x != x || fabs_(x) == inf()) ? qnan() : 0.0f;
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    float r; int j;
    sincos_reduce(x, &r, &j);
    if (j > 3) { sign = -sign; j -= 4; }
    float y = (j == 1 || j == 2) ? cos_poly(r) : sin_poly(r);
    return sign < 0.0f ? -y : y;
}
RT_HD float cos_(float x) {
    if (x != x || fabs_(x) >= 8192.0f) return (x != x || fabs_(x) == inf()) ? qnan() : 1.0f;
    if (x < 0.0f) x = -x;
    float sign = 1.0f;
    float r; int j;
    sincos_reduce(x, &r, &j);
    if (j > 3) { j -= 4; sign = -sign; }
    if (j > 1) sign = -sign;
    float y = (j == 1 || j == 2) ? sin_poly(r) : cos_poly(r);
    return sign < 0.0f ? -y : y;
}

/* ---- natural log: Cephes logf. ---- */
RT_HD float log_(float x) {
    if (x != x) return x;
    if (x < 0.0f) return qnan();
    if (x == 0.0f) return -inf();
    if (x == inf()) return x;
    uint32_t u = f2bits(x);
    int e = 0;
    if ((u & 0x7f800000u) == 0) { /* subnormal: scale by 2^23 exactly */
        x = x * 8388608.0f;
        u = f2bits(x);
        e = -23;
    }
    /* frexp: x = m * 2^e, m in [0.5, 1) */
    e += (int)((u >> 23) & 0xff) - 126;
    float m = bits2f((u & 0x007fffffu) | 0x3f000000u);
    const float SQRTHF = 0.707106781186547524f;
    if (m < SQRTHF) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = ((((((((7.0376836292E-2f * m - 1.1514610310E-1f) * m + 1.1676998740E-1f) * m - 1.2420140846E-1f) * m
                   + 1.4249322787E-1f) * m - 1.6668057665E-1f) * m + 2.0000714765E-1f) * m - 2.4999993993E-1f) * m
               + 3.3333331174E-1f) * m * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = m + y;
    z = z + 0.693359375f * fe;
    return z;
}

/* ---- exp: Cephes expf. ---- */
RT_HD float exp_(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return inf();
    if (x < -103.278929903431851103f) return 0.0f;
    const float LOG2EF = 1.44269504088896341f;
    const float C1 = 0.693359375f, C2 = -2.12194440e-4f;
    float t = LOG2EF * x + 0.5f;
    /* floor */
    int n = (int)t;
    if ((float)n > t) n -= 1;
    float fn = (float)n;
    x = x - fn * C1;
    x = x - fn * C2;
    float z = x * x;
    z = (((((1.9875691500E-4f * x + 1.3981999507E-3f) * x + 8.3334519073E-3f) * x + 4.1665795894E-2f) * x
          + 1.6666665459E-1f) * x + 5.0000001201E-1f) * z + x + 1.0f;
    /* ldexp(z, n) in two exact steps so that results in the subnormal range round once */
    int n1 = n / 2, n2 = n - n1;
    z = z * bits2f((uint32_t)(n1 + 127) << 23);
    z = z * bits2f((uint32_t)(n2 + 127) << 23);
    return z;
}

/* ---- pow(x, y) for the shader's use (x >= 0): exp(y * log(x)), with the HLSL corner cases. ---- */
RT_HD float pow_(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return qnan();
    if (x < 0.0f) return qnan();
    if (x == 0.0f) return y > 0.0f ? 0.0f : inf();
    if (x == 1.0f) return 1.0f;
    return exp_(y * log_(x));
}

/* ---- asin / acos: Cephes asinf/acosf. ---- */
RT_HD float sqrt_(float x);
RT_HD float asin_(float x) {
    float a = fabs_(x);
    if (x != x || a > 1.0f) return qnan();
    bool neg = x < 0.0f;
    float z, s;
    bool flag = false;
    if (a < 1.0e-4f) return x;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); s = sqrt_(z); flag = true; }
    else { s = a; z = s * s; }
    z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z
         + 1.6666752422E-1f) * z * s + s;
    if (flag) { z = z + z; z = 1.5707963267948966192f - z; }
    return neg ? -z : z;
}
RT_HD float acos_(float x) {
    if (x != x || x > 1.0f || x < -1.0f) return qnan();
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asin_(sqrt_(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin_(sqrt_(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - asin_(x);
}

/* ---- atan / atan2: Cephes atanf. ---- */
RT_HD float atan_(float x) {
    if (x != x) return x;
    bool neg = x < 0.0f;
    if (neg) x = -x;
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x);
    return neg ? -y : y;
}
RT_HD float atan2_(float y, float x) {
    const float PIF = 3.14159265358979323846f, PIO2F = 1.5707963267948966192f;
    if (x != x || y != y) return qnan();
    if (x == 0.0f) {
        if (y > 0.0f) return PIO2F;
        if (y < 0.0f) return -PIO2F;
        return 0.0f;
    }
    if (y == 0.0f) return x < 0.0f ? PIF : 0.0f;
    float w = 0.0f;
    if (x < 0.0f) w = y < 0.0f ? -PIF : PIF;
    return w + atan_(y / x);
}

/* IEEE correctly rounded square root on both compilers (hipcc keeps fp32 sqrt and divide correctly
 * rounded unless -fno-hip-fp32-correctly-rounded-divide-sqrt is given; the build never gives it). */
RT_HD float sqrt_(float x) { return __builtin_sqrtf(x); }

} /* namespace rtdm */

#endif /* RT_DETMATH_H */
