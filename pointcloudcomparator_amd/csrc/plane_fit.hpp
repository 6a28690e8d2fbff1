// plane_fit.hpp -- PCL's plane fit of a 3x3 covariance, shared by the device (k_normals) and the host
// (optimizeModelCoefficients of the RANSAC refit): pcl::eigen33 = scale by the largest entry, closed-form
// roots (pcl::computeRoots / computeRoots2), eigenvector of the smallest root from the largest cross product
// of two rows of A - lambda I; curvature = |lambda_0 / trace|.  Unfused fp32 throughout (-ffp-contract=off).
// The three transcendental calls (atan2f, cosf, sinf) are glibc's algorithms restated in libm_f32.hpp: the same bits on the
// device as the host libm PCL itself calls (round 5; until then the device rounded double results once and 2.3 % of all
// points differed from the host in their last bits), see DESIGN.md 4.6.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else  // plain C++ (the sanitizer build of the host-side code, `make asan`)
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#endif
#include <cfloat>
#include <cmath>
#include "libm_f32.hpp"

namespace pcc {

__host__ __device__ inline void pf_roots2(float b, float c, float r[3]) {
    r[0] = 0.f;
    float d = b * b - 4.f * c;
    if (d < 0.f) d = 0.f;
    const float sd = sqrtf(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}

// eigenvalues of the (scaled) symmetric matrix m (row-major), increasing
__host__ __device__ inline void pf_roots3(const float m[9], float r[3]) {
    const float c0 = m[0] * m[4] * m[8] + 2.f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const float c2 = m[0] + m[4] + m[8];
    if (fabsf(c0) < FLT_EPSILON) { pf_roots2(c2, c1, r); return; }
    const float s_inv3 = (float)(1.0 / 3.0);
    const float s_sqrt3 = 1.7320508f;  // sqrtf(3.0f)
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.f) q = 0.f;
    const float rho = sqrtf(-a_over_3);
    // (PCL calls std::atan2 / std::cos / std::sin on floats: the host libm's atan2f, cosf, sinf -- restated bit for bit in
    // libm_f32.hpp, so that device and host take the same roots)
    const float theta = lm_atan2f(sqrtf(-q), half_b) * s_inv3;
    const float ct = lm_cosf(theta), st = lm_sinf(theta);
    r[0] = c2_over_3 + 2.f * rho * ct;
    r[1] = c2_over_3 - rho * (ct + s_sqrt3 * st);
    r[2] = c2_over_3 - rho * (ct - s_sqrt3 * st);
    float t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    if (r[1] >= r[2]) {
        t = r[1]; r[1] = r[2]; r[2] = t;
        if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    }
    if (r[0] <= 0.f) pf_roots2(c2, c1, r);
}

__host__ __device__ inline void pf_cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// cov: symmetric 3x3, row-major.  n = unit eigenvector of the smallest eigenvalue, *curvature = |lambda_0 / trace|
__host__ __device__ inline void plane_from_covariance(const float cov[9], float n[3], float* curvature) {
    float scale = 0.f;
    for (int j = 0; j < 9; ++j) scale = fmaxf(scale, fabsf(cov[j]));
    if (scale <= FLT_MIN) scale = 1.f;
    float sm[9], ev[3];
    for (int j = 0; j < 9; ++j) sm[j] = cov[j] / scale;
    pf_roots3(sm, ev);
    const float eigenvalue = ev[0] * scale;
    sm[0] -= ev[0]; sm[4] -= ev[0]; sm[8] -= ev[0];
    float v1[3], v2[3], v3[3];
    pf_cross3(sm + 0, sm + 3, v1);
    pf_cross3(sm + 0, sm + 6, v2);
    pf_cross3(sm + 3, sm + 6, v3);
    const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    const float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
    const float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
    const float* v;
    float l;
    if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
    else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
    else { v = v3; l = l3; }
    const float s = sqrtf(l);
    n[0] = v[0] / s; n[1] = v[1] / s; n[2] = v[2] / s;
    const float eig_sum = cov[0] + cov[4] + cov[8];
    *curvature = eig_sum != 0.f ? fabsf(eigenvalue / eig_sum) : 0.f;
}

// PCL's computeMeanAndCovarianceMatrix tail: the nine single-pass sums (xx xy xz yy yz zz x y z) of cnt points
// -> covariance (row-major) and centroid.  accu /= cnt multiplies by the reciprocal (Eigen 3.2 operator/=).
__host__ __device__ inline void covariance_from_sums(float a[9], unsigned int cnt, float cov[9]) {
    const float inv = 1.0f / (float)cnt;
    for (int i = 0; i < 9; ++i) a[i] *= inv;
    cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
    cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
}

}  // namespace pcc
