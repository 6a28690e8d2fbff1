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
// When some points would NOT spread (curvature above the threshold) the rule gains one condition: such a
// point u passes its label on only if it is a seed itself (label(u) == rank(u); PCL pushes the seed of a
// region on the queue unconditionally).  "Is a seed" depends on lower-ranked points only, so the labels are
// still well defined; they are computed by repeating  label_new = min(own component's lowest rank, labels
// pushed along the edges that are active under label_old)  from scratch until nothing changes (the in-place
// minimum above would keep a label pushed by a point that turns out not to be a seed).
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
          const GridDev* __restrict__ gd, int K, float cos_thr, float curv_thr, unsigned int* __restrict__ parent) {
    const unsigned int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned int lane = threadIdx.x & 63;
    if (w >= gd->n_valid) return;
    const unsigned int u = (unsigned int)__float_as_int(cell_refs[w].w);
    const float4 nu = normals[u];
    if (nu.w > curv_thr) return;  // u does not spread: it shares nobody's label by construction
    const unsigned long long* row = keys + (size_t)u * K;
    for (int base = 0; base < K; base += 64) {
        const int j = base + (int)lane;
        const unsigned long long key = j < K ? row[j] : ~0ull;
        if (key_none(key)) continue;
        const unsigned int v = (unsigned int)key;
        if (v >= u) continue;  // every mutual pair is seen from both ends: the higher one links
        const float4 nv = normals[v];
        if (nv.w > curv_thr || !smooth_edge(nu, nv, cos_thr)) continue;
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
           unsigned char* __restrict__ has_cross, unsigned int* __restrict__ changed, unsigned int* __restrict__ cross_list,
           unsigned int* __restrict__ cross_count, unsigned int list_n) {
    // FIRST: a wave per point; the points with an edge into another component are LISTED, and the later sweeps are
    // launched over that list alone (a million waves that look up has_cross and leave cost 116 us a sweep, nine times)
    unsigned int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned int lane = threadIdx.x & 63;
    if (FIRST) {
        if (w >= gd->n_valid) return;
    } else {
        if (w >= list_n) return;
        w = cross_list[w];
    }
    const unsigned int u = (unsigned int)__float_as_int(cell_refs[w].w);
    const float4 nu = normals[u];
    const unsigned int ru = parent[u];
    const unsigned long long lu = __hip_atomic_load(&comp_label[ru], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long* row = keys + (size_t)u * K;
    bool cross = false, moved = false;
    for (int base = 0; base < K; base += 64) {
        const int j = base + (int)lane;
        const unsigned long long key = j < K ? row[j] : ~0ull;
        if (key_none(key)) continue;
        const unsigned int v = (unsigned int)key;
        const unsigned int rv = parent[v];
        if (rv == ru) continue;
        if (!smooth_edge(nu, normals[v], cos_thr)) continue;
        cross = true;
        if (atomicMin(&comp_label[rv], lu) > lu) moved = true;
    }
    if (FIRST) {
        const bool any_cross = __ballot(cross) != 0ull;
        if (lane == 0 && any_cross) {
            has_cross[u] = 1;
            cross_list[atomicAdd(cross_count, 1u)] = w;
        }
    }
    if (__ballot(moved) != 0ull && lane == 0) atomicOr(changed, 1u);
}

// general case (some points do not spread): one Jacobi step.  label_new was preset to the components' own lowest
// ranks; every point whose edges are active under label_old pushes label_old of its component across its valid
// edges into other components.
template <bool FIRST>
__global__ void __launch_bounds__(256)
k_rg_sweep_general(const unsigned long long* __restrict__ keys, const float4* __restrict__ normals,
                   const float4* __restrict__ cell_refs, const GridDev* __restrict__ gd, int K, float cos_thr, float curv_thr,
                   const unsigned int* __restrict__ parent, const unsigned long long* __restrict__ label_old,
                   unsigned long long* __restrict__ label_new, unsigned char* __restrict__ has_cross) {
    const unsigned int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned int lane = threadIdx.x & 63;
    if (w >= gd->n_valid) return;
    const unsigned int u = (unsigned int)__float_as_int(cell_refs[w].w);
    if (!FIRST && !has_cross[u]) return;
    const float4 nu = normals[u];
    const unsigned int ru = parent[u];
    const unsigned long long lu = label_old[ru];
    // a point that does not spread is a component of its own; it acts only while nobody has claimed it
    const bool active = !(nu.w > curv_thr) || lu == rank_key(nu.w, u);
    if (!FIRST && !active) return;
    const unsigned long long* row = keys + (size_t)u * K;
    bool cross = false;
    for (int base = 0; base < K; base += 64) {
        const int j = base + (int)lane;
        const unsigned long long key = j < K ? row[j] : ~0ull;
        if (key_none(key)) continue;
        const unsigned int v = (unsigned int)key;
        const unsigned int rv = parent[v];
        if (rv == ru) continue;
        if (!smooth_edge(nu, normals[v], cos_thr)) continue;
        cross = true;
        if (active) atomicMin(&label_new[rv], lu);
    }
    if (FIRST) {
        const bool any_cross = __ballot(cross) != 0ull;
        if (lane == 0 && any_cross) has_cross[u] = 1;
    }
}

__global__ void __launch_bounds__(256)
k_rg_diff(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b, unsigned int n,
          unsigned int* __restrict__ changed) {
    bool d = false;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d |= a[i] != b[i];
    if (__syncthreads_or(d) && threadIdx.x == 0) atomicOr(changed, 1u);
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
    unsigned int* d_words = ix->small.as<unsigned int>() + 40;  // [0] no-spread count, [1] changed, [2] list count, [3] points with a cross edge
    unsigned int* h = static_cast<unsigned int*>(ix->pinned);
    ev_mark(ix, EV_MAIN0);
    PCC_HIP(hipMemsetAsync(d_words, 0, 16, s));
    hipLaunchKernelGGL(k_rg_prepare, dim3(g1(n)), dim3(256), 0, s, keys, normals, n, K, curvature_threshold, parent,
                       comp_label, kth, size, id_of_seed, has_cross, d_words);
    PCC_HIP(hipGetLastError());
    PCC_HIP(hipMemcpyAsync(h, d_words, 4, hipMemcpyDeviceToHost, s));
    PCC_HIP(hipStreamSynchronize(s));
    const bool general = h[0] != 0;  // some point would join a region without spreading
    const unsigned int wave_blocks = (n + 3) / 4;
    hipLaunchKernelGGL(k_rg_link, dim3(wave_blocks), dim3(256), 0, s, keys, normals, kth, ix->cell_refs.as<float4>(),
                       ix->d_grid.as<GridDev>(), K, cos_thr, curvature_threshold, parent);
    hipLaunchKernelGGL(k_rg_compmin, dim3(g1(n)), dim3(256), 0, s, normals, n, parent, comp_label);
    PCC_HIP(hipGetLastError());
    if (general) {
        // comp_label holds the components' own lowest ranks; label_old / label_new alternate
        PCC_TRY(ix->vox_a.reserve((size_t)n * 16));
        unsigned long long* lab[2] = {ix->vox_a.as<unsigned long long>(), ix->vox_a.as<unsigned long long>() + n};
        PCC_HIP(hipMemcpyAsync(lab[0], comp_label, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        int cur = 0;
        for (int sweep = 0;; ++sweep) {
            if (sweep > 100000) { set_error("region growing did not settle"); return PCC_ERR_DEVICE; }
            PCC_HIP(hipMemcpyAsync(lab[cur ^ 1], comp_label, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
            PCC_HIP(hipMemsetAsync(d_words + 1, 0, 4, s));
            if (sweep == 0)
                hipLaunchKernelGGL(k_rg_sweep_general<true>, dim3(wave_blocks), dim3(256), 0, s, keys, normals,
                                   ix->cell_refs.as<float4>(), ix->d_grid.as<GridDev>(), K, cos_thr, curvature_threshold, parent,
                                   lab[cur], lab[cur ^ 1], has_cross);
            else
                hipLaunchKernelGGL(k_rg_sweep_general<false>, dim3(wave_blocks), dim3(256), 0, s, keys, normals,
                                   ix->cell_refs.as<float4>(), ix->d_grid.as<GridDev>(), K, cos_thr, curvature_threshold, parent,
                                   lab[cur], lab[cur ^ 1], has_cross);
            hipLaunchKernelGGL(k_rg_diff, dim3(g1(n)), dim3(256), 0, s, lab[cur], lab[cur ^ 1], n, d_words + 1);
            PCC_HIP(hipGetLastError());
            PCC_HIP(hipMemcpyAsync(h, d_words + 1, 4, hipMemcpyDeviceToHost, s));
            PCC_HIP(hipStreamSynchronize(s));
            cur ^= 1;
            if (h[0] == 0) break;
        }
        PCC_HIP(hipMemcpyAsync(comp_label, lab[cur], (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    } else {
        PCC_TRY(ix->knn_fb.reserve(((size_t)n + 1) * sizeof(unsigned int)));  // (free here: the rows were searched before)
        unsigned int* cross_list = ix->knn_fb.as<unsigned int>();
        unsigned int n_cross = 0;
        for (int sweep = 0;; ++sweep) {
            PCC_HIP(hipMemsetAsync(d_words + 1, 0, 4, s));
            if (sweep == 0)
                hipLaunchKernelGGL(k_rg_sweep<true>, dim3(wave_blocks), dim3(256), 0, s, keys, normals, ix->cell_refs.as<float4>(),
                                   ix->d_grid.as<GridDev>(), K, cos_thr, parent, comp_label, has_cross, d_words + 1, cross_list,
                                   d_words + 3, 0u);
            else if (n_cross)
                hipLaunchKernelGGL(k_rg_sweep<false>, dim3((n_cross + 3) / 4), dim3(256), 0, s, keys, normals,
                                   ix->cell_refs.as<float4>(), ix->d_grid.as<GridDev>(), K, cos_thr, parent, comp_label, has_cross,
                                   d_words + 1, cross_list, d_words + 3, n_cross);
            PCC_HIP(hipGetLastError());
            PCC_HIP(hipMemcpyAsync(h, d_words + 1, 12, hipMemcpyDeviceToHost, s));  // changed, (list count of collect), cross count
            PCC_HIP(hipStreamSynchronize(s));
            if (sweep == 0) n_cross = h[2];
            if (h[0] == 0 || n_cross == 0) break;
        }
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
