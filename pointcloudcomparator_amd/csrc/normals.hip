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
#include "lane_ops.hpp"
#include "grid_device.hpp"
#include "plane_fit.hpp"

namespace pcc {

namespace {

// NR_KC neighbour columns at a time: each wave copies the rows of its 64 points into LDS with coalesced
// loads (lanes over columns), then every lane walks its own row from LDS.  Points are taken in CELL order,
// so the neighbour gathers of adjacent lanes overlap in L1 (in index order every gather was a miss and every
// 8-byte row read touched its own line: 1.5 ms at 1M x K = 50 against 0.3 ms of useful traffic).
constexpr int NR_KC = 32;
constexpr int NR_U = 8;

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
            wave_lds_sync();
            // (the 64 row numbers are the lanes' own point indices: read out with v_readlane, eight rows' loads in flight (sixteen measured the same) --
            // one row at a time, each behind a load of its row number, was 128 dependent round trips per tile)
            for (int r0 = 0; r0 < 64; r0 += 8) {
                if (base + (unsigned int)r0 >= n_valid) break;  // wave-uniform
                unsigned long long kk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const size_t row = (size_t)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)i, r0 + u);
                    kk[u] = ~0ull;
                    if (base + (unsigned int)(r0 + u) < n_valid && (int)lane < kc) kk[u] = keys[row * (size_t)K + c0 + lane];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (base + (unsigned int)(r0 + u) < n_valid && (int)lane < kc)
                        tile[r0 + u][lane] = key_none(kk[u]) ? 0xffffffffu : (unsigned int)kk[u];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            wave_lds_sync();
            // NR_U neighbours at a time: their gathers are issued together, the sums then take them in row order (PCL's
            // order of accumulation).  One gather per trip of the loop left the kernel waiting 92 % of its time.
            for (int j0 = 0; j0 < kc; j0 += NR_U) {
                unsigned int id[NR_U];
                float4 p[NR_U];
#pragma unroll
                for (int u = 0; u < NR_U; ++u) id[u] = (open && j0 + u < kc) ? tile[lane][j0 + u] : 0xffffffffu;
#pragma unroll
                for (int u = 0; u < NR_U; ++u)
                    if (id[u] != 0xffffffffu) p[u] = refs[id[u]];
#pragma unroll
                for (int u = 0; u < NR_U; ++u) {
                    if (id[u] == 0xffffffffu) { open = false; continue; }  // (rows are ascending: nothing follows a missing entry)
                    ++cnt;
                    a0 += p[u].x * p[u].x; a1 += p[u].x * p[u].y; a2 += p[u].x * p[u].z;
                    a3 += p[u].y * p[u].y; a4 += p[u].y * p[u].z; a5 += p[u].z * p[u].z;
                    a6 += p[u].x; a7 += p[u].y; a8 += p[u].z;
                }
            }
        }
        if (!have) continue;
        if (cnt < 3) { out[i] = make_float4(qnan, qnan, qnan, qnan); continue; }
        float acc[9] = {a0, a1, a2, a3, a4, a5, a6, a7, a8}, cov[9], nrm[3], curv;
        covariance_from_sums(acc, (unsigned int)cnt, cov);
        plane_from_covariance(cov, nrm, &curv);
        float nx = nrm[0], ny = nrm[1], nz = nrm[2];
        // flipNormalTowardsViewpoint
        const float4 p = refs[i];
        const float dx = vpx - p.x, dy = vpy - p.y, dz = vpz - p.z;
        const float cos_theta = dx * nx + dy * ny + dz * nz;
        if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
        out[i] = make_float4(nx, ny, nz, curv);
    }
}

// the same plane fit over variable-length rows (sorted radius search results, CSR): one lane per point in cell order
__global__ void __launch_bounds__(256)
k_normals_csr(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ offsets,
              const float4* __restrict__ refs, const float4* __restrict__ cell_refs, const GridDev* __restrict__ gd,
              float vpx, float vpy, float vpz, float4* __restrict__ out) {
    const float qnan = __uint_as_float(0x7fc00000u);
    const unsigned int n_valid = gd->n_valid;
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_valid; t += gridDim.x * blockDim.x) {
        const unsigned int i = (unsigned int)__float_as_int(cell_refs[t].w);
        const unsigned int beg = offsets[i], end = offsets[i + 1];
        float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (unsigned int j = beg; j < end; ++j) {
            const float4 p = refs[(unsigned int)keys[j]];
            acc[0] += p.x * p.x; acc[1] += p.x * p.y; acc[2] += p.x * p.z;
            acc[3] += p.y * p.y; acc[4] += p.y * p.z; acc[5] += p.z * p.z;
            acc[6] += p.x; acc[7] += p.y; acc[8] += p.z;
        }
        const unsigned int cnt = end - beg;
        if (cnt < 3) { out[i] = make_float4(qnan, qnan, qnan, qnan); continue; }
        float cov[9], nrm[3], curv;
        covariance_from_sums(acc, cnt, cov);
        plane_from_covariance(cov, nrm, &curv);
        float nx = nrm[0], ny = nrm[1], nz = nrm[2];
        const float4 p = refs[i];
        const float dx = vpx - p.x, dy = vpy - p.y, dz = vpz - p.z;
        const float cos_theta = dx * nx + dy * ny + dz * nz;
        if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
        out[i] = make_float4(nx, ny, nz, curv);
    }
}

// counts (int32 per point) -> 64-bit total, to refuse what a 32-bit CSR cannot hold
__global__ void __launch_bounds__(256)
k_sum_counts(const int32_t* __restrict__ counts, unsigned int n, unsigned long long* __restrict__ total) {
    unsigned long long s = 0;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += (unsigned int)counts[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(total, s);
}
__global__ void __launch_bounds__(256)
k_widen_offsets(const unsigned int* __restrict__ off32, unsigned int n_plus_1, int64_t* __restrict__ off64) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_plus_1; i += gridDim.x * blockDim.x) off64[i] = (int64_t)off32[i];
}

}  // namespace

// NormalEstimation with setRadiusSearch: self radius search (sorted rows, CSR) + the plane fit per row
int normals_radius(pcc_index* ix, double radius, const float vp[3], float4* out) {
    hipStream_t s = ix->stream;
    const size_t n = ix->n_orig;
    const float r2 = (float)(radius * radius);
    const float4* self = ix->refs.as<float4>();
    // counts -> offsets
    PCC_TRY(ix->vox_a.reserve((n + 1) * sizeof(unsigned int) + (n + 1) * sizeof(int64_t) + 64));
    unsigned int* off32 = ix->vox_a.as<unsigned int>();
    int64_t* off64 = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(ix->vox_a.p) + (((n + 1) * sizeof(unsigned int) + 15) & ~(size_t)15));
    PCC_HIP(hipMemsetAsync(off32, 0, (n + 1) * sizeof(unsigned int), s));
    PCC_TRY(grid_radius(ix, self, n, (float)radius, r2, reinterpret_cast<int32_t*>(off32), nullptr, nullptr, 0));
    unsigned long long* d_total = reinterpret_cast<unsigned long long*>(ix->small.as<unsigned int>() + 44);
    PCC_HIP(hipMemsetAsync(d_total, 0, 8, s));
    const unsigned int blocks = (unsigned int)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_sum_counts, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const int32_t*>(off32), (unsigned int)n, d_total);
    unsigned long long* h_total = reinterpret_cast<unsigned long long*>(static_cast<unsigned int*>(ix->pinned) + 44);
    PCC_HIP(hipMemcpyAsync(h_total, d_total, 8, hipMemcpyDeviceToHost, s));
    PCC_HIP(hipStreamSynchronize(s));
    const unsigned long long total = *h_total;
    if (total >= (1ull << 32)) { set_error("radius neighbourhoods hold %llu entries: more than a 32-bit CSR takes", total); return PCC_ERR_OVERFLOW; }
    PCC_TRY(launch_exclusive_scan(ix, s, off32, n + 1, ix->vox_b));
    hipLaunchKernelGGL(k_widen_offsets, dim3(blocks), dim3(256), 0, s, off32, (unsigned int)(n + 1), off64);
    PCC_HIP(hipGetLastError());
    // sorted fill
    PCC_TRY(ix->out_packed.reserve((size_t)(total ? total : 1) * sizeof(unsigned long long)));
    auto* keys = ix->out_packed.as<unsigned long long>();
    if (total) PCC_TRY(grid_radius(ix, self, n, (float)radius, r2, nullptr, off64, keys, 1, (size_t)total));
    PCC_HIP(hipMemsetAsync(out, 0xff, n * sizeof(float4), s));
    hipLaunchKernelGGL(k_normals_csr, dim3(blocks), dim3(256), 0, s, keys, off32, self, ix->cell_refs.as<float4>(),
                       ix->d_grid.as<GridDev>(), vp[0], vp[1], vp[2], out);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

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
