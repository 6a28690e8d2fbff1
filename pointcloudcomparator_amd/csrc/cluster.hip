// cluster.hip -- Euclidean clustering as connected components of the radius graph (gfx950).
//
// Replaces pcl::EuclideanClusterExtraction::extract (reference src/segmentation.cpp:125-131):
// PCL grows clusters with a sequential BFS over radiusSearch results; the clusters it returns
// are exactly the connected components of the graph {(i, j) : d2(i, j) < r2}, so
// the GPU path builds the components directly with a lock-free union-find over the GRID index
// and never materialises neighbour lists:
//   link    : every point scans the cells its r-ball touches and unions itself with each
//             neighbour of LOWER position (each edge once).  Roots only ever point to smaller
//             positions, so a component's root is its lowest member -- PCL's seed point.
//   flatten : parent[i] = root(i)
//   size    : histogram of roots
//   collect : roots whose size lies in [min, max]; the (few) survivors are ordered on the host
//             by size descending, lowest member first among equal sizes
//   label   : labels[original index] = cluster id or -1
#include "pcc_internal.hpp"
#include "grid_device.hpp"
#include "uf_device.hpp"
#include <algorithm>
#include <vector>

namespace pcc {

__global__ void __launch_bounds__(256)
k_uf_init(unsigned int* __restrict__ parent, unsigned int n) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) parent[i] = i;
}

// thread t owns the t-th point in CELL order (neighbouring lanes touch neighbouring cells)
__global__ void __launch_bounds__(256)
k_uf_link(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
          const GridDev* __restrict__ gd, float r, float r2, unsigned int* __restrict__ parent) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= gd->n_valid) return;  // cell_refs holds the valid points only
    const float4 me = cell_refs[t];
    const unsigned int mypos = (unsigned int)__float_as_int(me.w);
    int x0, x1, y0, y1, z0, z1;
    const float rr = r + slack;
    cell_range(me.x, rr, g.org[0], g.inv_h, g.dim[0], x0, x1);
    cell_range(me.y, rr, g.org[1], g.inv_h, g.dim[1], y0, y1);
    cell_range(me.z, rr, g.org[2], g.inv_h, g.dim[2], z0, z1);
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y) {
            const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
            const unsigned int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
            for (unsigned int p = s; p < e; ++p) {
                const float4 o = cell_refs[p];
                const unsigned int opos = (unsigned int)__float_as_int(o.w);
                if (opos < mypos && dist2(me.x, me.y, me.z, o) < r2) uf_union(parent, mypos, opos);
            }
        }
}

__global__ void __launch_bounds__(256)
k_uf_flatten_count(const float4* __restrict__ refs, unsigned int* __restrict__ parent, unsigned int n,
                   unsigned int* __restrict__ size) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (__float_as_int(refs[i].w) < 0) continue;  // non-finite point: in no cluster
        unsigned int r = uf_find(parent, i);
        atomicMin(&parent[i], r);  // other threads still reach the same root through older values
        atomicAdd(&size[r], 1u);
    }
}

__global__ void __launch_bounds__(256)
k_uf_collect(const unsigned int* __restrict__ parent, const unsigned int* __restrict__ size, unsigned int n,
             unsigned int min_size, unsigned int max_size, uint2* __restrict__ list, unsigned int* __restrict__ count,
             unsigned int cap) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (parent[i] != i) continue;
        unsigned int sz = size[i];  // 0 for a non-finite point (never counted)
        if (sz == 0) continue;
        if (sz < min_size || sz > max_size) continue;
        unsigned int slot = atomicAdd(count, 1u);
        if (slot < cap) list[slot] = make_uint2(i, sz);
    }
}

__global__ void __launch_bounds__(256)
k_uf_set_ids(const uint2* __restrict__ sorted_list, unsigned int ncl, int* __restrict__ id_of_root) {
    for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < ncl; k += gridDim.x * blockDim.x)
        id_of_root[sorted_list[k].x] = (int)k;
}

__global__ void __launch_bounds__(256)
k_uf_label(const float4* __restrict__ refs, const unsigned int* __restrict__ parent, const int* __restrict__ id_of_root,
           unsigned int n, int32_t* __restrict__ labels) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (__float_as_int(refs[i].w) >= 0) labels[i] = id_of_root[parent[i]];  // position == original index
}

static inline int g1(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int grid_clusters(pcc_index* ix, float r, float r2, uint32_t min_size, uint32_t max_size,
                  int32_t* labels_dev, int32_t* n_clusters, int32_t* sizes, int max_sizes) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)ix->n_orig;
    // scratch: parent[n] | size[n] | id_of_root[n] | list[cap]
    const unsigned int cap = min_size > 0 ? n / min_size + 1 : n;
    PCC_TRY(ix->scratch_c.reserve((size_t)n * 4));
    PCC_TRY(ix->scratch_d.reserve((size_t)n * 4));
    PCC_TRY(ix->scratch_e.reserve((size_t)n * 4));
    PCC_TRY(ix->scratch_b.reserve((size_t)cap * sizeof(uint2) + 16));
    unsigned int* parent = ix->scratch_c.as<unsigned int>();
    unsigned int* size = ix->scratch_d.as<unsigned int>();
    int* id_of_root = ix->scratch_e.as<int>();
    uint2* list = ix->scratch_b.as<uint2>();
    unsigned int* d_count = ix->small.as<unsigned int>() + 40;
    ev_mark(ix, EV_MAIN0);
    hipLaunchKernelGGL(k_uf_init, dim3(g1(n)), dim3(256), 0, s, parent, n);
    PCC_HIP(hipMemsetAsync(size, 0, (size_t)n * 4, s));
    PCC_HIP(hipMemsetAsync(id_of_root, 0xff, (size_t)n * 4, s));
    PCC_HIP(hipMemsetAsync(d_count, 0, 4, s));
    PCC_HIP(hipMemsetAsync(labels_dev, 0xff, ix->n_orig * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_uf_link, dim3((n + 255) / 256), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                       ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), r, r2, parent);
    hipLaunchKernelGGL(k_uf_flatten_count, dim3(g1(n)), dim3(256), 0, s, ix->refs.as<float4>(), parent, n, size);
    hipLaunchKernelGGL(k_uf_collect, dim3(g1(n)), dim3(256), 0, s, parent, size, n, min_size, max_size, list, d_count, cap);
    PCC_HIP(hipGetLastError());
    unsigned int* h = static_cast<unsigned int*>(ix->pinned);
    PCC_HIP(hipMemcpyAsync(h, d_count, 4, hipMemcpyDeviceToHost, s));
    PCC_HIP(hipStreamSynchronize(s));
    unsigned int ncl = h[0];
    if (ncl > cap) { set_error("cluster list overflow (%u > %u)", ncl, cap); return PCC_ERR_OVERFLOW; }
    std::vector<uint2> host_list(ncl);
    if (ncl) {
        PCC_HIP(hipMemcpyAsync(host_list.data(), list, (size_t)ncl * sizeof(uint2), hipMemcpyDeviceToHost, s));
        PCC_HIP(hipStreamSynchronize(s));
        // PCL: std::sort(clusters.rbegin(), clusters.rend(), by size) -> largest first; equal sizes are
        // left unspecified there, here: lowest member index first (root == lowest member)
        std::sort(host_list.begin(), host_list.end(), [](const uint2& a, const uint2& b) {
            if (a.y != b.y) return a.y > b.y;
            return a.x < b.x;
        });
        PCC_HIP(hipMemcpyAsync(list, host_list.data(), (size_t)ncl * sizeof(uint2), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_uf_set_ids, dim3(g1(ncl)), dim3(256), 0, s, list, ncl, id_of_root);
    }
    hipLaunchKernelGGL(k_uf_label, dim3(g1(n)), dim3(256), 0, s, ix->refs.as<float4>(), parent, id_of_root, n, labels_dev);
    PCC_HIP(hipGetLastError());
    ev_mark(ix, EV_MAIN1);
    PCC_HIP(hipStreamSynchronize(s));  // host_list must outlive the H2D copy
    if (n_clusters) *n_clusters = (int32_t)ncl;
    if (sizes)
        for (unsigned int k = 0; k < ncl && (int)k < max_sizes; ++k) sizes[k] = (int32_t)host_list[k].y;
    return PCC_OK;
}

}  // namespace pcc
