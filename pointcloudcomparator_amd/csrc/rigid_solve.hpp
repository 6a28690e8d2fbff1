// rigid_solve.hpp -- rigid transform from the 17 ICP sums (Horn's closed form), shared by the host path
// (pcc_rigid_from_sums, the convergence-checked ICP loop) and the device-resident loop (k_icp_solve, icp.hip).
// Plain IEEE double arithmetic and correctly rounded sqrt on both sides, compiled with -ffp-contract=off: the device
// copy returns the host copy's bits.
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
#include <cmath>
#include <cstring>

namespace pcc {

#define PCC_HD __host__ __device__ inline
#if defined(__HIPCC__)
#define PCC_UNROLL _Pragma("unroll")
#else
#define PCC_UNROLL
#endif

// largest eigenvector of a symmetric 4x4 (cyclic Jacobi), for Horn's closed-form absolute orientation
PCC_HD void sym4_max_eigvec(double A[4][4], double v[4]) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    // (every loop but the sweep is unrolled: on the device all indices are then static and both matrices stay in
    // registers -- with dynamic indices they live in scratch memory and one solve took 165 us)
    // Sweeps stop once the off-diagonal mass is at the rounding floor of the matrix (1e-30 of its squared Frobenius
    // norm, which rotations preserve) or has stopped shrinking.  The old test, off < 1e-300, never fired on real sums --
    // a rotation leaves a residue of ~1e-16 |A|, not zero -- so every solve ran all 64 sweeps: ~25 us on the host per
    // ICP pass, 160 us as one lane of k_icp_solve.
    double fro = 0;
PCC_UNROLL
    for (int p = 0; p < 4; ++p)
PCC_UNROLL
        for (int q = 0; q < 4; ++q) fro += A[p][q] * A[p][q];
    double prev_off = 1.79769313486231570e308;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0;
PCC_UNROLL
        for (int p = 0; p < 4; ++p)
PCC_UNROLL
            for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        if (off <= 1e-30 * fro || !(off < prev_off)) break;
        prev_off = off;
PCC_UNROLL
        for (int p = 0; p < 4; ++p)
PCC_UNROLL
            for (int q = p + 1; q < 4; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
PCC_UNROLL
                for (int k = 0; k < 4; ++k) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq;
                    A[k][q] = sn * akp + c * akq;
                }
PCC_UNROLL
                for (int k = 0; k < 4; ++k) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - sn * aqk;
                    A[q][k] = sn * apk + c * aqk;
                }
PCC_UNROLL
                for (int k = 0; k < 4; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    // (selects instead of a dynamic column index, for the same reason)
    double bestd = A[0][0];
PCC_UNROLL
    for (int k = 0; k < 4; ++k) v[k] = V[k][0];
PCC_UNROLL
    for (int i = 1; i < 4; ++i)
        if (A[i][i] > bestd) {
            bestd = A[i][i];
PCC_UNROLL
            for (int k = 0; k < 4; ++k) v[k] = V[k][i];
        }
}

// rigid transform (rotation + translation, no scale) minimising sum |R p + t - q|^2 from the sums:
// the same optimum TransformationEstimationSVD / Eigen::umeyama(src, tgt, false) returns, obtained
// with Horn's unit-quaternion method in double.  Returns 0, or -1 with < 3 correspondences.
// `center` (nullable): the sums were accumulated over p - center and q - center.  S = sum q p^T - n pm qm^T cancels
// catastrophically for a small cloud far from the origin (43 points 8 cm across at (1e3, 1e5, 1e5): sums of 4e11 for
// a covariance of 0.3 -- the rotation came out 3e-5 rad off, 7 mm on that cloud); pcc_icp_align therefore sums about a
// point of the SOURCE cloud (k_icp_center: its first valid point; the centre of the target's bounding box was no better
// once stray points stretched the box).  The translation is formed with the true means.
PCC_HD int rigid_from_sums(const double sums[17], float T[16], const double* center = nullptr) {
    const double n = sums[16];
    if (n < 3) return -1;  // min_number_correspondences_
    double pm[3], qm[3], S[3][3];
    for (int a = 0; a < 3; ++a) { pm[a] = sums[a] / n; qm[a] = sums[3 + a] / n; }
    for (int a = 0; a < 3; ++a)      // S[a][b] = sum (p_a - pm_a)(q_b - qm_b)
        for (int b = 0; b < 3; ++b) S[a][b] = sums[6 + b * 3 + a] - n * pm[a] * qm[b];
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double qv[4];
    sym4_max_eigvec(N, qv);
    double nrm = sqrt(qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2] + qv[3] * qv[3]);
    if (!(nrm > 0)) return -1;
    const double w = qv[0] / nrm, x = qv[1] / nrm, y = qv[2] / nrm, z = qv[3] / nrm;
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                            {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                            {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
    if (center)
        for (int a = 0; a < 3; ++a) { pm[a] += center[a]; qm[a] += center[a]; }
    for (int r = 0; r < 3; ++r) {
        double t = qm[r];
        for (int c = 0; c < 3; ++c) { T[r * 4 + c] = (float)R[r][c]; t -= R[r][c] * pm[c]; }
        T[r * 4 + 3] = (float)t;
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
    return 0;
}

PCC_HD void mat4_mul_f(const float A[16], const float B[16], float C[16]) {
    float R[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float acc = 0;
            for (int k = 0; k < 4; ++k) acc += A[r * 4 + k] * B[k * 4 + c];
            R[r * 4 + c] = acc;
        }
    for (int k = 0; k < 16; ++k) C[k] = R[k];
}


#undef PCC_HD
}  // namespace pcc
