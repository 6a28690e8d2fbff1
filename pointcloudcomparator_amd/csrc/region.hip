// region.hip -- pcl::RegionGrowing::extract (reference src/segmentation.cpp:259-271) on the GPU.
//
// PCL grows regions one after another: the unlabelled point of lowest curvature seeds a region,
// which then spreads breadth first along the k-neighbour rows -- u claims a still unlabelled
// neighbour v when |n_u . n_v| >= cos(theta), and v keeps spreading when its curvature is not above
// the curvature threshold.  With the reference's threshold (1.0; PCL's curvature never exceeds
// 1/3) EVERY claimed point spreads, and then the outcome has an order-free description:
//
//     label(v) = the lowest-ranked point (curvature, index) that reaches v along valid edges.
//
// (The lowest-ranked ancestor u* of v can not have been claimed by an earlier seed s -- s would reach
// v through u* and rank below u* -- so u* seeds a region; the same argument keeps every point of the
// path u* -> v unclaimed until then, so that region takes v, and no earlier one does.)
//
// That is what the kernels compute, without ever forming PCL's queue:
//   prepare : rank key of every point, the K-th key of every row (u is in row(v) iff its key is
//             <= that, distances being bitwise symmetric), count of points that would NOT spread
//   link    : one wave per point, one lane per neighbour: valid edges present in BOTH rows are
//             merged with the lock-free union-find -- both ends reach each other, so they share
//             their label.  On smooth surfaces this already builds the regions.
//   compmin : label of a component = lowest rank key among its members (64-bit atomicMin)
//   sweep   : one-directional valid edges between different components push the lower label
//             across; repeated until nothing changes (points without such an edge are skipped
//             after the first sweep, so later sweeps are short)
//   count / collect / label : region sizes, size filter, ids in PCL's creation order (= rank of
//             the seed), labels
// When some point would not spread (a curvature threshold below the data's curvatures) the growth
// is order dependent; that case runs PCL's sequential walk on the host over the same GPU rows.
#include <algorithm>
#include <cmath>
#include <vector>

#include "pcc_internal.hpp"
#include "grid_device.hpp"
#include "uf_device.hpp"

namespace pcc {

namespace {

struct SeedRec {
    unsigned long long key;  // rank key of the seed: (ordered curvature bits << 32) | index
    unsigned int size, pad;
};

// (curvature, index) as one unsigned key: ascending curvature, -0 == +0, NaN last, ties by index
__device__ __forceinline__ unsigned long long rank_key(float curv, unsigned int i) {
    unsigned int b = __float_as_uint(curv);
    unsigned int o;
    if (curv != curv) o = 0xffffffffu;
    else if (curv == 0.f) o = 0x80000000u;
    else o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)o << 32) | i;
}

// RegionGrowing::validatePoint, smooth mode: rejected only when |n_u . n_v| < cos(theta)
__device__ __forceinline__ bool smooth_edge(const float4 a, const float4 b, float cos_thr) {
    const float dot = fabsf(b.x * a.x + b.y * a.y + b.z * a.z);
    return !(dot < cos_thr);
}

__global__ void __launch_bounds__(256)
k_rg_prepare(const unsigned long long* __restrict__ keys, const float4* __restrict__ normals, unsigned int n, int K,
             float curv_thr, unsigned int* __restrict__ parent, unsigned long long* __restrict__ comp_label,
             unsigned long long* __restrict__ kth, unsigned int* __restrict__ size, int* __restrict__ id_of_seed,
             unsigned char* __restrict__ has_cross, unsigned int* __restrict__ n_no_spread) {
    for (unsigned int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const unsigned int i = base + threadIdx.x;
        int no_spread = 0;
        if (i < n) {
            parent[i] = i;
            comp_label[i] = ~0ull;
            kth[i] = keys[(size_t)i * K + (K - 1)];
            size[i] = 0;
            id_of_seed[i] = -1;
            has_cross[i] = 0;
            no_spread = normals[i].w > curv_thr;
        }
        const int c = __syncthreads_count(no_spread);
        if (threadIdx.x == 0 && c) atomicAdd(n_no_spread, (unsigned int)c);
    }
}

// wave w owns the w-th valid point in CELL order: its row is read coalesced, its neighbours'
// normals and K-th keys are gathers into a spatially compact set
__global__ void __launch_bounds__(256)
k_rg_link(const unsigned long long* __restrict__ keys, const float4* __restrict__ normals,
          const unsigned long long* __restrict__ kth, const float4* __restrict__ cell_refs,
          const GridDev* __restrict__ gd, int K, float cos_thr, unsigned int* __restrict__ parent) {
    const unsigned int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned int lane = threadIdx.x & 63;
    if (w >= gd->n_valid) return;
    const unsigned int u = (unsigned int)__float_as_int(cell_refs[w].w);
    const float4 nu = normals[u];
    const unsigned long long* row = keys + (size_t)u * K;
    for (int base = 0; base < K; base += 64) {
        const int j = base + (int)lane;
        const unsigned long long key = j < K ? row[j] : ~0ull;
        if (key == ~0ull) continue;
        const unsigned int v = (unsigned int)key;
        if (v >= u) continue;  // every mutual pair is seen from both ends: the higher one links
        if (!smooth_edge(nu, normals[v], cos_thr)) continue;
        const unsigned long long mine = (key & 0xffffffff00000000ull) | u;  // u's key in v's ordering
        if (mine <= kth[v]) uf_union(parent, u, v);
    }
}

__global__ void __launch_bounds__(256)
k_rg_compmin(const float4* __restrict__ normals, unsigned int n, unsigned int* __restrict__ parent,
             unsigned long long* __restrict__ comp_label) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned int r = uf_find(parent, i);
        atomicMin(&parent[i], r);
        atomicMin(&comp_label[r], rank_key(normals[i].w, i));
    }
}

// parent[] is flat here.  FIRST: every point is visited and remembers whether it has an edge into
// another component; later sweeps visit those points only.
template <bool FIRST>
__global__ void __launch_bounds__(256)
k_rg_sweep(const unsigned long long* __restrict__ keys, const float4* __restrict__ normals,
           const float4* __restrict__ cell_refs, const GridDev* __restrict__ gd, int K, float cos_thr,
           const unsigned int* __restrict__ parent, unsigned long long* __restrict__ comp_label,
           unsigned char* __restrict__ has_cross, unsigned int* __restrict__ changed) {
    const unsigned int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned int lane = threadIdx.x & 63;
    if (w >= gd->n_valid) return;
    const unsigned int u = (unsigned int)__float_as_int(cell_refs[w].w);
    if (!FIRST && !has_cross[u]) return;
    const float4 nu = normals[u];
    const unsigned int ru = parent[u];
    const unsigned long long lu = __hip_atomic_load(&comp_label[ru], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long* row = keys + (size_t)u * K;
    bool cross = false, moved = false;
    for (int base = 0; base < K; base += 64) {
        const int j = base + (int)lane;
        const unsigned long long key = j < K ? row[j] : ~0ull;
        if (key == ~0ull) continue;
        const unsigned int v = (unsigned int)key;
        const unsigned int rv = parent[v];
        if (rv == ru) continue;
        if (!smooth_edge(nu, normals[v], cos_thr)) continue;
        cross = true;
        if (atomicMin(&comp_label[rv], lu) > lu) moved = true;
    }
    if (FIRST) {
        const bool any_cross = __ballot(cross) != 0ull;
        if (lane == 0 && any_cross) has_cross[u] = 1;
    }
    if (__ballot(moved) != 0ull && lane == 0) atomicOr(changed, 1u);
}

// parent[i] becomes the seed of i's region; sizes are counted per seed
__global__ void __launch_bounds__(256)
k_rg_count(const unsigned long long* __restrict__ comp_label, unsigned int n, unsigned int* __restrict__ parent,
           unsigned int* __restrict__ size) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned int seed = (unsigned int)comp_label[parent[i]];
        parent[i] = seed;  // only thread i reads parent[i]
        atomicAdd(&size[seed], 1u);
    }
}

__global__ void __launch_bounds__(256)
k_rg_collect(const float4* __restrict__ normals, const unsigned int* __restrict__ size, unsigned int n,
             unsigned int min_size, unsigned int max_size, SeedRec* __restrict__ list, unsigned int* __restrict__ count,
             unsigned int cap) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned int sz = size[i];  // non-zero for seeds only
        if (sz == 0 || sz < min_size || sz > max_size) continue;
        const unsigned int slot = atomicAdd(count, 1u);
        if (slot < cap) list[slot] = SeedRec{rank_key(normals[i].w, i), sz, 0u};
    }
}

__global__ void __launch_bounds__(256)
k_rg_set_ids(const SeedRec* __restrict__ sorted_list, unsigned int ncl, int* __restrict__ id_of_seed) {
    for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < ncl; k += gridDim.x * blockDim.x)
        id_of_seed[(unsigned int)sorted_list[k].key] = (int)k;
}

__global__ void __launch_bounds__(256)
k_rg_label(const unsigned int* __restrict__ seed_of, const int* __restrict__ id_of_seed, unsigned int n,
           int32_t* __restrict__ labels) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        labels[i] = id_of_seed[seed_of[i]];
}

inline int g1(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// pcl::RegionGrowing::extract over precomputed neighbour rows (host side of pcc_region_growing).
// Regions are grown in PCL's order: seeds by ascending curvature (ties: lower index), each region
// a breadth-first walk of the neighbour rows; a neighbour joins when |n_cur . n_nbr| >= cos(theta)
// and continues the walk when its curvature is not above the threshold.
int region_growing_host(size_t n, const float* normals4, const int32_t* nbr, int K, float smoothness,
                        float curvature_threshold, uint32_t min_size, uint32_t max_size, int32_t* labels,
                        int32_t* n_clusters) {
    std::vector<int32_t> seg(n, -1), order(n), queue(n);
    for (size_t i = 0; i < n; ++i) order[i] = (int32_t)i;
    // NaN curvatures (points without a normal) go last; PCL's std::sort leaves them unspecified
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const float ca = normals4[(size_t)a * 4 + 3], cb = normals4[(size_t)b * 4 + 3];
        const bool na = ca != ca, nb = cb != cb;
        if (na != nb) return nb;
        if (!na && ca != cb) return ca < cb;
        return a < b;
    });
    std::vector<uint32_t> seg_size;
    const float cosine_threshold = cosf(smoothness);
    size_t segmented = 0, seed_pos = 0;
    while (segmented < n) {
        while (seg[order[seed_pos]] != -1) ++seed_pos;
        const int32_t id = (int32_t)seg_size.size();
        size_t qh = 0, qt = 0;
        queue[qt++] = order[seed_pos];
        seg[order[seed_pos]] = id;
        uint32_t cnt = 1;
        while (qh < qt) {
            const int32_t cur = queue[qh++];
            const float* nc = normals4 + (size_t)cur * 4;
            const int32_t* row = nbr + (size_t)cur * K;
            for (int j = 0; j < K; ++j) {
                const int32_t t = row[j];
                if (t < 0) break;
                if (seg[t] != -1) continue;
                const float* nn = normals4 + (size_t)t * 4;
                const float dot = fabsf(nn[0] * nc[0] + nn[1] * nc[1] + nn[2] * nc[2]);
                if (dot < cosine_threshold) continue;
                seg[t] = id;
                ++cnt;
                if (!(nn[3] > curvature_threshold)) queue[qt++] = t;
            }
        }
        seg_size.push_back(cnt);
        segmented += cnt;
    }
    std::vector<int32_t> remap(seg_size.size());
    int32_t kept = 0;
    for (size_t s = 0; s < seg_size.size(); ++s) remap[s] = (seg_size[s] >= min_size && seg_size[s] <= max_size) ? kept++ : -1;
    for (size_t i = 0; i < n; ++i) labels[i] = remap[seg[i]];
    *n_clusters = kept;
    return PCC_OK;
}


}  // namespace

// keys: the self k-NN rows of the index (n x K); normals: float4 (nx, ny, nz, curvature) on the device
int grid_region_growing(pcc_index* ix, const unsigned long long* keys, const float4* normals, int K, float smoothness,
                        float curvature_threshold, uint32_t min_size, uint32_t max_size, int32_t* labels_dev,
                        int32_t* n_clusters) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)ix->n_orig;
    const float cos_thr = cosf(smoothness);
    const unsigned int cap = min_size > 0 ? n / min_size + 1 : n;
    PCC_TRY(ix->scratch_a.reserve((size_t)n * 8));  // comp_label
    PCC_TRY(ix->scratch_b.reserve((size_t)cap * sizeof(SeedRec) + 16));
    PCC_TRY(ix->scratch_c.reserve((size_t)n * 4));  // parent -> seed_of
    PCC_TRY(ix->scratch_d.reserve((size_t)n * 4));  // size
    PCC_TRY(ix->scratch_e.reserve((size_t)n * 4));  // id_of_seed
    PCC_TRY(ix->scratch_f.reserve((size_t)n * 8));  // kth
    PCC_TRY(ix->scratch_g.reserve((size_t)n));      // has_cross
    auto* comp_label = ix->scratch_a.as<unsigned long long>();
    auto* list = ix->scratch_b.as<SeedRec>();
    auto* parent = ix->scratch_c.as<unsigned int>();
    auto* size = ix->scratch_d.as<unsigned int>();
    auto* id_of_seed = ix->scratch_e.as<int>();
    auto* kth = ix->scratch_f.as<unsigned long long>();
    auto* has_cross = ix->scratch_g.as<unsigned char>();
    unsigned int* d_words = ix->small.as<unsigned int>() + 40;  // [0] no-spread count, [1] changed, [2] list count
    unsigned int* h = static_cast<unsigned int*>(ix->pinned);
    ev_mark(ix, EV_MAIN0);
    PCC_HIP(hipMemsetAsync(d_words, 0, 12, s));
    hipLaunchKernelGGL(k_rg_prepare, dim3(g1(n)), dim3(256), 0, s, keys, normals, n, K, curvature_threshold, parent,
                       comp_label, kth, size, id_of_seed, has_cross, d_words);
    PCC_HIP(hipGetLastError());
    PCC_HIP(hipMemcpyAsync(h, d_words, 4, hipMemcpyDeviceToHost, s));
    PCC_HIP(hipStreamSynchronize(s));
    if (h[0] != 0) {
        // some point would join a region without spreading: PCL's sequential walk, same rows
        PCC_TRY(ix->out_idx.reserve((size_t)n * K * sizeof(int32_t)));
        PCC_TRY(launch_unpack(s, keys, nullptr, (size_t)n * K, ix->out_idx.as<int32_t>(), nullptr));
        std::vector<int32_t> nbr((size_t)n * K), hl(n);
        std::vector<float> hn((size_t)n * 4);
        PCC_HIP(hipMemcpyAsync(nbr.data(), ix->out_idx.p, nbr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        PCC_HIP(hipMemcpyAsync(hn.data(), normals, hn.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        PCC_HIP(hipStreamSynchronize(s));
        PCC_TRY(region_growing_host(n, hn.data(), nbr.data(), K, smoothness, curvature_threshold, min_size, max_size,
                                    hl.data(), n_clusters));
        PCC_HIP(hipMemcpyAsync(labels_dev, hl.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
        PCC_HIP(hipStreamSynchronize(s));
        ev_mark(ix, EV_MAIN1);
        return PCC_OK;
    }
    const unsigned int wave_blocks = (n + 3) / 4;
    hipLaunchKernelGGL(k_rg_link, dim3(wave_blocks), dim3(256), 0, s, keys, normals, kth, ix->cell_refs.as<float4>(),
                       ix->d_grid.as<GridDev>(), K, cos_thr, parent);
    hipLaunchKernelGGL(k_rg_compmin, dim3(g1(n)), dim3(256), 0, s, normals, n, parent, comp_label);
    PCC_HIP(hipGetLastError());
    for (int sweep = 0;; ++sweep) {
        PCC_HIP(hipMemsetAsync(d_words + 1, 0, 4, s));
        if (sweep == 0)
            hipLaunchKernelGGL(k_rg_sweep<true>, dim3(wave_blocks), dim3(256), 0, s, keys, normals, ix->cell_refs.as<float4>(),
                               ix->d_grid.as<GridDev>(), K, cos_thr, parent, comp_label, has_cross, d_words + 1);
        else
            hipLaunchKernelGGL(k_rg_sweep<false>, dim3(wave_blocks), dim3(256), 0, s, keys, normals, ix->cell_refs.as<float4>(),
                               ix->d_grid.as<GridDev>(), K, cos_thr, parent, comp_label, has_cross, d_words + 1);
        PCC_HIP(hipGetLastError());
        PCC_HIP(hipMemcpyAsync(h, d_words + 1, 4, hipMemcpyDeviceToHost, s));
        PCC_HIP(hipStreamSynchronize(s));
        if (h[0] == 0) break;
    }
    hipLaunchKernelGGL(k_rg_count, dim3(g1(n)), dim3(256), 0, s, comp_label, n, parent, size);
    hipLaunchKernelGGL(k_rg_collect, dim3(g1(n)), dim3(256), 0, s, normals, size, n, min_size, max_size, list, d_words + 2, cap);
    PCC_HIP(hipGetLastError());
    PCC_HIP(hipMemcpyAsync(h, d_words + 2, 4, hipMemcpyDeviceToHost, s));
    PCC_HIP(hipStreamSynchronize(s));
    const unsigned int ncl = h[0];
    if (ncl > cap) { set_error("region list overflow (%u > %u)", ncl, cap); return PCC_ERR_OVERFLOW; }
    std::vector<SeedRec> host_list(ncl);
    if (ncl) {
        PCC_HIP(hipMemcpyAsync(host_list.data(), list, (size_t)ncl * sizeof(SeedRec), hipMemcpyDeviceToHost, s));
        PCC_HIP(hipStreamSynchronize(s));
        // PCL numbers the regions in creation order = ascending rank of their seeds
        std::sort(host_list.begin(), host_list.end(), [](const SeedRec& a, const SeedRec& b) { return a.key < b.key; });
        PCC_HIP(hipMemcpyAsync(list, host_list.data(), (size_t)ncl * sizeof(SeedRec), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_rg_set_ids, dim3(g1(ncl)), dim3(256), 0, s, list, ncl, id_of_seed);
    }
    hipLaunchKernelGGL(k_rg_label, dim3(g1(n)), dim3(256), 0, s, parent, id_of_seed, n, labels_dev);
    PCC_HIP(hipGetLastError());
    ev_mark(ix, EV_MAIN1);
    PCC_HIP(hipStreamSynchronize(s));  // host_list must outlive the H2D copy
    *n_clusters = (int32_t)ncl;
    return PCC_OK;
}

}  // namespace pcc
