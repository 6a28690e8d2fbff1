// normals.hip -- surface normals + curvature from k-neighbourhoods: the first half of the reference's
// default segmentation path (src/segmentation.cpp:232-241 pcl::NormalEstimation setKSearch(50); the
// region growing that consumes them is region.hip).
//
// The neighbourhoods come from the wave-cooperative self k-NN (knn.hip).  k_normals is one lane
// per point: it walks its row of keys in ascending (d2, index) order -- the order PCL's
// nearestKSearch hands to computeMeanAndCovarianceMatrix -- so the float accumulators round
// exactly as PCL's do, then solves the smallest eigenpair in closed form the way pcl::eigen33
// does.  The three transcendental calls are evaluated in double and rounded once, which matches
// a correctly rounded libm float result.
#include <algorithm>
#include <cmath>
#include <vector>

#include "pcc_internal.hpp"
#include "grid_device.hpp"

namespace pcc {

namespace {

__device__ __forceinline__ void roots2(float b, float c, float r[3]) {
    r[0] = 0.f;
    float d = b * b - 4.f * c;
    if (d < 0.f) d = 0.f;
    const float sd = sqrtf(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}

// eigenvalues of the (scaled) symmetric matrix, increasing; pcl::computeRoots
__device__ __forceinline__ void roots3(const float m[9], float r[3]) {
    const float c0 = m[0] * m[4] * m[8] + 2.f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const float c2 = m[0] + m[4] + m[8];
    if (fabsf(c0) < 1.1920929e-07f) { roots2(c2, c1, r); return; }
    const float s_inv3 = (float)(1.0 / 3.0);
    const float s_sqrt3 = 1.7320508f;  // sqrtf(3.0f)
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.f) q = 0.f;
    const float rho = sqrtf(-a_over_3);
    const float theta = (float)atan2((double)sqrtf(-q), (double)half_b) * s_inv3;
    const float ct = (float)cos((double)theta), st = (float)sin((double)theta);
    r[0] = c2_over_3 + 2.f * rho * ct;
    r[1] = c2_over_3 - rho * (ct + s_sqrt3 * st);
    r[2] = c2_over_3 - rho * (ct - s_sqrt3 * st);
    float t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    if (r[1] >= r[2]) {
        t = r[1]; r[1] = r[2]; r[2] = t;
        if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    }
    if (r[0] <= 0.f) roots2(c2, c1, r);
}

__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// NR_KC neighbour columns at a time: each wave copies the rows of its 64 points into LDS with coalesced
// loads (lanes over columns), then every lane walks its own row from LDS.  Points are taken in CELL order,
// so the neighbour gathers of adjacent lanes overlap in L1 (in index order every gather was a miss and every
// 8-byte row read touched its own line: 1.5 ms at 1M x K = 50 against 0.3 ms of useful traffic).
constexpr int NR_KC = 32;

__global__ void __launch_bounds__(256)
k_normals(const unsigned long long* __restrict__ keys, const float4* __restrict__ refs, const float4* __restrict__ cell_refs,
          const GridDev* __restrict__ gd, int K, float vpx, float vpy, float vpz, float4* __restrict__ out) {
    __shared__ unsigned int tile_all[4][64][NR_KC + 1];
    unsigned int (*tile)[NR_KC + 1] = tile_all[threadIdx.x >> 6];
    const unsigned int lane = threadIdx.x & 63;
    const float qnan = __uint_as_float(0x7fc00000u);
    const unsigned int n_valid = gd->n_valid;
    const unsigned int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned int base = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64; base < n_valid; base += nwaves * 64) {
        const unsigned int t = base + lane;
        const bool have = t < n_valid;
        const size_t i = have ? (size_t)(unsigned int)__float_as_int(cell_refs[t].w) : 0;
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
        int cnt = 0;
        bool open = have;  // row not exhausted yet
        for (int c0 = 0; c0 < K; c0 += NR_KC) {
            const int kc = min(NR_KC, K - c0);
            __builtin_amdgcn_wave_barrier();
            for (int r = 0; r < 64; ++r) {
                const unsigned int rr = base + r;
                if (rr >= n_valid) break;  // wave-uniform
                const size_t row = (size_t)(unsigned int)__float_as_int(cell_refs[rr].w);
                if ((int)lane < kc) {
                    const unsigned long long key = keys[row * (size_t)K + c0 + lane];
                    tile[r][lane] = key == ~0ull ? 0xffffffffu : (unsigned int)key;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int j = 0; j < kc; ++j) {
                const unsigned int idx = open ? tile[lane][j] : 0xffffffffu;
                if (idx == 0xffffffffu) { open = false; continue; }
                const float4 p = refs[idx];
                ++cnt;
                a0 += p.x * p.x; a1 += p.x * p.y; a2 += p.x * p.z;
                a3 += p.y * p.y; a4 += p.y * p.z; a5 += p.z * p.z;
                a6 += p.x; a7 += p.y; a8 += p.z;
            }
        }
        if (!have) continue;
        if (cnt < 3) { out[i] = make_float4(qnan, qnan, qnan, qnan); continue; }
        // accu /= point_count: Eigen 3.2's operator/=(scalar) multiplies by Scalar(1)/other
        const float fc = 1.0f / (float)cnt;
        a0 *= fc; a1 *= fc; a2 *= fc; a3 *= fc; a4 *= fc; a5 *= fc; a6 *= fc; a7 *= fc; a8 *= fc;
        float cov[9];
        cov[0] = a0 - a6 * a6; cov[1] = a1 - a6 * a7; cov[2] = a2 - a6 * a8;
        cov[4] = a3 - a7 * a7; cov[5] = a4 - a7 * a8; cov[8] = a5 - a8 * a8;
        cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
        // pcl::eigen33: scale, roots, eigenvector of the smallest root from the largest row cross product
        float scale = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) scale = fmaxf(scale, fabsf(cov[j]));
        if (scale <= 1.17549435e-38f) scale = 1.f;
        float sm[9], ev[3];
#pragma unroll
        for (int j = 0; j < 9; ++j) sm[j] = cov[j] / scale;
        roots3(sm, ev);
        const float eigenvalue = ev[0] * scale;
        sm[0] -= ev[0]; sm[4] -= ev[0]; sm[8] -= ev[0];
        float v1[3], v2[3], v3[3];
        cross3(sm + 0, sm + 3, v1);
        cross3(sm + 0, sm + 6, v2);
        cross3(sm + 3, sm + 6, v3);
        const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
        const float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
        const float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
        float vx, vy, vz, l;
        if (l1 >= l2 && l1 >= l3) { vx = v1[0]; vy = v1[1]; vz = v1[2]; l = l1; }
        else if (l2 >= l1 && l2 >= l3) { vx = v2[0]; vy = v2[1]; vz = v2[2]; l = l2; }
        else { vx = v3[0]; vy = v3[1]; vz = v3[2]; l = l3; }
        const float s = sqrtf(l);
        float nx = vx / s, ny = vy / s, nz = vz / s;
        const float eig_sum = cov[0] + cov[4] + cov[8];
        const float curv = eig_sum != 0.f ? fabsf(eigenvalue / eig_sum) : 0.f;
        // flipNormalTowardsViewpoint
        const float4 p = refs[i];
        const float dx = vpx - p.x, dy = vpy - p.y, dz = vpz - p.z;
        const float cos_theta = dx * nx + dy * ny + dz * nz;
        if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
        out[i] = make_float4(nx, ny, nz, curv);
    }
}

}  // namespace

int launch_normals(hipStream_t s, const unsigned long long* keys, const float4* refs, const float4* cell_refs,
                   const GridDev* gd, size_t n, int K, const float vp[3], float4* out) {
    if (n == 0) return PCC_OK;
    // points that are not in the cell order (non-finite) keep the NaN the buffer is filled with
    PCC_HIP(hipMemsetAsync(out, 0xff, n * sizeof(float4), s));
    const size_t blocks = std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_normals, dim3((unsigned)blocks), dim3(256), 0, s, keys, refs, cell_refs, gd, K, vp[0], vp[1], vp[2], out);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

}  // namespace pcc
