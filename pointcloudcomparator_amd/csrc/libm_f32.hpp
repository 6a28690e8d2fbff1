// libm_f32.hpp -- glibc 2.35's sinf / cosf / atan2f, restated so that the device evaluates what the host's libm does.
//
// pcl::eigen33 (NormalEstimation, the RANSAC refit; reference src/segmentation.cpp:232-241, 79-117) takes the roots of the
// characteristic cubic in closed form: theta = atan2f(sqrtf(-q), half_b) / 3, then cosf(theta), sinf(theta).  The smallest
// root is a difference of O(1) terms, so an ulp in those three calls is an ulp-sized slice of the curvature and the normal:
// with the device's own (correctly rounded) versions 2.3 % of all points differed from the host's in their last bits.  A libm
// is not specified to the bit, but an IMPLEMENTATION is: these are the algorithms of the glibc the oracle runs against
// (2.35, x86-64), operation for operation, compiled without contraction on both sides:
//   sinf / cosf : sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h (Szabolcs Nagy / Wilco Dijkstra's double-precision
//                 polynomials: reduction by pi/2 through a scaled integer conversion, degree-7 / degree-8 minimax forms), in
//                 the form glibc runs on CPUs with FMA (__sinf_fma / __cosf_fma: every multiply-add contracted -- explicit
//                 fma() here, v_fma_f64 on the device).  On [0, 1.2] -- theta never leaves [0, pi/3] -- the contracted and the
//                 baseline build return the same float for every one of the 1 067 030 939 arguments; over [0, 120) they
//                 differ for ~1 argument in two million (cancellation next to the zeros).
//   atan2f      : sysdeps/ieee754/flt-32/e_atan2f.c + s_atanf.c (fdlibm: float arithmetic, four reduction intervals, an
//                 11-term odd / even polynomial; one build for all CPUs).  The constants are the ones the binary holds
//                 (aT[0] is 0x3eaaaaab; the source's comment says ...aa).
// tests/test_libm_cpu.py (tests/cpp/test_libm.cpp) compares them with the host's libm: sinf / cosf on every float of
// [2^-13, 1.2] and 80M arguments over (-120, 120), atanf on every 64th float of the whole line plus three whole binades,
// atan2f on 40M pairs and the special cases -- all bits equal (the full sweeps, 2^32 arguments of atanf and 4e8 pairs of
// atan2f, were run once when this file was written: no mismatch).  Beyond 120 in magnitude sinf / cosf fall back to the
// double-precision functions (never reached by the callers).
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#endif
#include <cmath>
#include <cstdint>
#include <cstring>

namespace pcc {

__host__ __device__ inline uint32_t lm_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
#endif
}
__host__ __device__ inline float lm_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

// ---- sinf / cosf ------------------------------------------------------------------------------------------------------
// a * b + c as glibc's build for FMA-capable CPUs evaluates it (__sinf_fma / __cosf_fma: every multiply-add of the
// polynomials and of the reduction contracted); LM_NO_FMA gives the baseline build's form (tests)
#ifdef LM_NO_FMA
#define LM_MADD(a, b, c) ((a) * (b) + (c))
#else
#define LM_MADD(a, b, c) fma((a), (b), (c))
#endif
struct LmSincos {
    double sign[4];
    double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3;
};
__host__ __device__ inline LmSincos lm_sincos_table(int k) {
    // __sincosf_table[0] and [1] (the second negates the cosine polynomial)
    const double sg = k ? -1.0 : 1.0;
    return LmSincos{{1.0, -1.0, -1.0, 1.0},
                    0x1.45F306DC9C883p+23,  // 2 / pi * 2^24
                    0x1.921FB54442D18p0,    // pi / 2
                    sg * 0x1p0, sg * -0x1.ffffffd0c621cp-2, sg * 0x1.55553e1068f19p-5, sg * -0x1.6c087e89a359dp-10, sg * 0x1.99343027bf8c3p-16,
                    -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13};
}
__host__ __device__ inline uint32_t lm_abstop12(float x) { return (lm_bits(x) >> 20) & 0x7ffu; }
// sin (n even) or cos (n odd) polynomial of the reduced argument, evaluated in double and rounded once
__host__ __device__ inline float lm_sinf_poly(double x, double x2, const LmSincos& p, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = LM_MADD(x2, p.s3, p.s2);
        const double x7 = x3 * x2;
        const double s = LM_MADD(x3, p.s1, x);
        return (float)LM_MADD(x7, s1, s);
    }
    const double x4 = x2 * x2;
    const double c2 = LM_MADD(x2, p.c4, p.c3);
    const double c1 = LM_MADD(x2, p.c1, p.c0);
    const double x6 = x4 * x2;
    const double c = LM_MADD(x4, p.c2, c1);
    return (float)LM_MADD(x6, c2, c);
}
__host__ __device__ inline double lm_reduce_fast(double x, const LmSincos& p, int* np) {
    const double r = x * p.hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return LM_MADD(-(double)n, p.hpi, x);
}
__host__ __device__ inline float lm_sinf(float y) {
    double x = (double)y;
    if (lm_abstop12(y) < lm_abstop12(0x1.921FB6p-1f)) {  // |y| < pi / 4
        if (lm_abstop12(y) < lm_abstop12(0x1p-12f)) return y;
        return lm_sinf_poly(x, x * x, lm_sincos_table(0), 0);
    }
    if (lm_abstop12(y) < lm_abstop12(120.0f)) {
        int n;
        const LmSincos p0 = lm_sincos_table(0);
        x = lm_reduce_fast(x, p0, &n);
        const double s = p0.sign[n & 3];
        return lm_sinf_poly(x * s, x * x, lm_sincos_table((n & 2) ? 1 : 0), n);
    }
    return (float)sin((double)y);  // (large arguments: not glibc's reduction; no caller gets here)
}
__host__ __device__ inline float lm_cosf(float y) {
    double x = (double)y;
    if (lm_abstop12(y) < lm_abstop12(0x1.921FB6p-1f)) {
        if (lm_abstop12(y) < lm_abstop12(0x1p-12f)) return 1.0f;
        return lm_sinf_poly(x, x * x, lm_sincos_table(0), 1);
    }
    if (lm_abstop12(y) < lm_abstop12(120.0f)) {
        int n;
        const LmSincos p0 = lm_sincos_table(0);
        x = lm_reduce_fast(x, p0, &n);
        const double s = p0.sign[n & 3];
        return lm_sinf_poly(x * s, x * x, lm_sincos_table((n & 2) ? 1 : 0), n ^ 1);
    }
    return (float)cos((double)y);
}

// ---- atanf / atan2f (fdlibm, float arithmetic) ---------------------------------------------------------------------------
__host__ __device__ inline float lm_atanf(float x) {
    const float atanhi[4] = {lm_float(0x3eed6338u), lm_float(0x3f490fdau), lm_float(0x3f7b985eu), lm_float(0x3fc90fdau)};
    const float atanlo[4] = {lm_float(0x31ac3769u), lm_float(0x33222168u), lm_float(0x33140fb4u), lm_float(0x33a22168u)};
    const float aT0 = lm_float(0x3eaaaaabu), aT1 = lm_float(0xbe4ccccdu), aT2 = lm_float(0x3e124925u), aT3 = lm_float(0xbde38e38u),
                aT4 = lm_float(0x3dba2e6eu), aT5 = lm_float(0xbd9d8795u), aT6 = lm_float(0x3d886b35u), aT7 = lm_float(0xbd6ef16bu),
                aT8 = lm_float(0x3d4bda59u), aT9 = lm_float(0xbd15a221u), aT10 = lm_float(0x3c8569d7u);
    const int32_t hx = (int32_t)lm_bits(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {  // |x| < 0.4375
        if (ix < 0x31000000) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {      // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}
__host__ __device__ inline float lm_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = lm_float(0x3f490fdbu), pi_o_2 = lm_float(0x3fc90fdbu), pi = lm_float(0x40490fdbu),
                pi_lo = lm_float(0xb3bbbd2eu);
    const int32_t hx = (int32_t)lm_bits(x), ix = hx & 0x7fffffff, hy = (int32_t)lm_bits(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  // NaN
    if (hx == 0x3f800000) return lm_atanf(y);              // x = 1
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);     // 2 * sign(x) + sign(y)
    if (iy == 0) {
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            if (m == 0) return pi_o_4 + tiny;
            if (m == 1) return -pi_o_4 - tiny;
            return m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        }
        if (m == 0) return 0.0f;
        if (m == 1) return -0.0f;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = lm_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return lm_float(lm_bits(z) ^ 0x80000000u);
    return m == 2 ? pi - (z - pi_lo) : (z - pi_lo) - pi;
}

}  // namespace pcc
