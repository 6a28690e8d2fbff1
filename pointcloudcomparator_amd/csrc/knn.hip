// knn.hip -- k nearest neighbours and radius search on the GRID index (gfx950).
//
// k-NN replaces pcl::KdTreeFLANN::nearestKSearch(pt, k, ...) with k = 51 inside
// StatisticalOutlierRemoval (reference src/comparator.cpp:1523-1541); radius search
// replaces KdTreeFLANN::radiusSearch(pt, r, ...) as used by pcl::extractEuclideanClusters
// (reference src/segmentation.cpp:125-131).  Same unfused fp32 distance as the k=1 kernels;
// results ordered by (d2, original index) exactly like FLANN's sorted result sets, with the
// lowest index first among exact ties.
//
// One lane owns one query (queries are cell-sorted, so a wave walks neighbouring rows).
// k-NN keeps each query's K best (d2 bits << 32 | position) keys as an ascending list in the
// output buffer itself; a candidate is compared against the current K-th key first, so once
// the list is full almost every candidate costs one compare.
#include "pcc_internal.hpp"
#include "grid_device.hpp"
#include <cmath>

namespace pcc {

__device__ __forceinline__ unsigned long long make_key(float d, const float4& r) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)__float_as_int(r.w);
}

// insert key into the ascending list[0..K) (unused slots hold ~0)
__device__ __forceinline__ void knn_insert(unsigned long long* __restrict__ list, int K, unsigned long long key,
                                           unsigned long long& worst) {
    int j = K - 1;
    while (j > 0) {
        unsigned long long prev = list[j - 1];
        if (prev <= key) break;
        list[j] = prev;
        --j;
    }
    list[j] = key;
    worst = list[K - 1];
}

__device__ __forceinline__ void knn_scan_span(const float4* __restrict__ cell_refs, unsigned int s, unsigned int e,
                                              float qx, float qy, float qz, unsigned long long* __restrict__ list,
                                              int K, unsigned long long& worst) {
    for (unsigned int p = s; p < e; p += 2) {
        const float4 r0 = cell_refs[p];
        const bool two = p + 1 < e;
        const float4 r1 = cell_refs[two ? p + 1 : p];
        const unsigned long long k0 = make_key(dist2(qx, qy, qz, r0), r0);
        const unsigned long long k1 = make_key(dist2(qx, qy, qz, r1), r1);
        if (k0 < worst) knn_insert(list, K, k0, worst);
        if (two && k1 < worst) knn_insert(list, K, k1, worst);
    }
}

__global__ void __launch_bounds__(256)
k_grid_knn(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
           const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned int* __restrict__ order,
           const unsigned int* __restrict__ n_sorted_ptr, unsigned int n, int K,
           unsigned long long* __restrict__ keys) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    const unsigned int n_valid = gd->n_valid;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int ns = *n_sorted_ptr;
    if (t >= ns || n_valid == 0) return;
    const unsigned int qi = order[t];
    const float4 qv = q[qi];
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    const int cx = cell_coord(qx, g.org[0], g.inv_h, g.dim[0]);
    const int cy = cell_coord(qy, g.org[1], g.inv_h, g.dim[1]);
    const int cz = cell_coord(qz, g.org[2], g.inv_h, g.dim[2]);
    unsigned long long* list = keys + (size_t)qi * K;
    const int want = (unsigned int)K < n_valid ? K : (int)n_valid;  // k clamped to the valid points
    int k = 1;
    for (;;) {
        const bool whole = k > GRID_KMAX;  // past KMAX: scan the whole grid (exact, slow, rare)
        const int x0 = whole ? 0 : max(cx - k, 0), x1 = whole ? g.dim[0] - 1 : min(cx + k, g.dim[0] - 1);
        const int y0 = whole ? 0 : max(cy - k, 0), y1 = whole ? g.dim[1] - 1 : min(cy + k, g.dim[1] - 1);
        const int z0 = whole ? 0 : max(cz - k, 0), z1 = whole ? g.dim[2] - 1 : min(cz + k, g.dim[2] - 1);
        unsigned long long worst = ~0ull;
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
                knn_scan_span(cell_refs, cell_start[row + x0], cell_start[row + x1 + 1], qx, qy, qz, list, K, worst);
            }
        const float lb2 = outside_bound2(qx, qy, qz, x0, x1, y0, y1, z0, z1, g, slack);
        const unsigned long long kth = list[want - 1];
        if (lb2 == __builtin_inff()) break;  // whole grid scanned
        if (kth != ~0ull && __uint_as_float((unsigned int)(kth >> 32)) < lb2) break;
        // grow and rescan from scratch (a rescan must not insert a point twice)
        int kn = k + 1;
        if (kth != ~0ull) {
            const float need = sqrtf(__uint_as_float((unsigned int)(kth >> 32))) * g.inv_h;
            kn = need < (float)GRID_KMAX ? max((int)need + 1, k + 1) : GRID_KMAX + 1;
        } else if (k >= 2) {
            kn = 2 * k;
        }
        k = kn;
        for (int j = 0; j < K; ++j) list[j] = ~0ull;
    }
}

int grid_knn(pcc_index* ix, const float4* q, size_t nq, int K, unsigned long long* keys) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)nq;
    PCC_HIP(hipMemsetAsync(keys, 0xff, nq * (size_t)K * sizeof(unsigned long long), s));
    unsigned int *order = nullptr, *n_sorted = nullptr;
    PCC_TRY(grid_sort_queries(ix, q, nq, &order, &n_sorted));
    ev_mark(ix, EV_MAIN0);
    hipLaunchKernelGGL(k_grid_knn, dim3((n + 255) / 256), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                       ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, order, n_sorted, n, K, keys);
    PCC_HIP(hipGetLastError());
    ev_mark(ix, EV_MAIN1);
    return PCC_OK;
}

// ---- radius search ---------------------------------------------------------------------------
// FILL = false: counts[i] = #refs with d2 < r2 (strict).
// FILL = true : keys written at offsets[i]; with sorted the row is then ordered by (d2, position)
//               with an in-place insertion sort (rows are short: tens to a few hundred entries).
template <bool FILL>
__global__ void __launch_bounds__(256)
k_grid_radius(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
              const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned int* __restrict__ order,
              const unsigned int* __restrict__ n_sorted_ptr, unsigned int n, float r, float r2,
              int32_t* __restrict__ counts, const int64_t* __restrict__ offsets,
              unsigned long long* __restrict__ keys, int sorted) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *n_sorted_ptr) return;
    const unsigned int qi = order[t];
    const float4 qv = q[qi];
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    int x0, x1, y0, y1, z0, z1;
    const float rr = r + slack;  // cells that can hold a point within r (conservative)
    cell_range(qx, rr, g.org[0], g.inv_h, g.dim[0], x0, x1);
    cell_range(qy, rr, g.org[1], g.inv_h, g.dim[1], y0, y1);
    cell_range(qz, rr, g.org[2], g.inv_h, g.dim[2], z0, z1);
    unsigned int cnt = 0;
    unsigned long long* row_out = FILL ? keys + offsets[qi] : nullptr;
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y) {
            const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
            const unsigned int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
            for (unsigned int p = s; p < e; ++p) {
                const float4 rp = cell_refs[p];
                const float d = dist2(qx, qy, qz, rp);
                if (d < r2) {
                    if (FILL) row_out[cnt] = make_key(d, rp);
                    ++cnt;
                }
            }
        }
    if (!FILL) {
        counts[qi] = (int32_t)cnt;
    } else if (sorted) {
        for (unsigned int i = 1; i < cnt; ++i) {
            unsigned long long key = row_out[i];
            unsigned int j = i;
            while (j > 0 && row_out[j - 1] > key) { row_out[j] = row_out[j - 1]; --j; }
            row_out[j] = key;
        }
    }
}

int grid_radius(pcc_index* ix, const float4* q, size_t nq, float r, float r2, int32_t* counts,
                const int64_t* offsets, unsigned long long* keys, int sorted) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)nq;
    unsigned int *order = nullptr, *n_sorted = nullptr;
    PCC_TRY(grid_sort_queries(ix, q, nq, &order, &n_sorted));
    const GridDev* gd = ix->d_grid.as<GridDev>();
    ev_mark(ix, EV_MAIN0);
    if (keys)
        hipLaunchKernelGGL((k_grid_radius<true>), dim3((n + 255) / 256), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), gd, q, order, n_sorted, n, r, r2, counts,
                           offsets, keys, sorted);
    else
        hipLaunchKernelGGL((k_grid_radius<false>), dim3((n + 255) / 256), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), gd, q, order, n_sorted, n, r, r2, counts,
                           offsets, keys, sorted);
    PCC_HIP(hipGetLastError());
    ev_mark(ix, EV_MAIN1);
    return PCC_OK;
}

}  // namespace pcc
