// voxel.hip -- pcl::VoxelGrid on the GPU (gfx950): the down-sampling both segmentation paths of the
// reference start with (src/segmentation.cpp:69-74 and :224-229, leaf 0.025 m).
//
// PCL (pcl/filters/impl/voxel_grid.hpp; the tests hold a CPU restatement): voxel of a point =
// floor(p * inverse_leaf) - min_b on the world-aligned leaf lattice; points are sorted by voxel index
// and every voxel is replaced by the centroid of its points (x, y, z and, with colour, the mean
// r, g, b truncated to integers); output order = ascending voxel index.
// Here: pack (+bbox) -> cell sort with the voxel id as key (cellsort.hip, no global atomics) ->
// head flags + scan -> one thread per voxel sums its points.  PCL adds the points in the order an
// unstable std::sort left them (float); this kernel adds x, y, z in double in whatever order the
// sort produced and rounds once, so each coordinate is within a float ulp or two of PCL's and
// does not depend on the order; colour sums are integers (exact) divided in float exactly as PCL
// does.  HBM-bound streaming passes apart from the sort's scatter.
#include "pcc_internal.hpp"
#include "grid_device.hpp"
#include <cmath>
#include <cstring>
#include <vector>

namespace pcc {

constexpr unsigned int VOX_MAX_CELLS = 1u << 26;

__global__ void __launch_bounds__(256)
k_vox_heads(const float4* __restrict__ pts, const unsigned int* __restrict__ order, const unsigned int* __restrict__ n_sorted,
            const GridDev* __restrict__ gd, unsigned int* __restrict__ flags, unsigned int n) {
    const GridParams g = gd->g;
    const unsigned int ns = *n_sorted;
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t <= n; t += gridDim.x * blockDim.x) {
        unsigned int f = 0;
        if (t < ns) f = (t == 0 || voxel_id(pts[order[t]], g) != voxel_id(pts[order[t - 1]], g)) ? 1u : 0u;
        flags[t] = f;  // flags[n] = 0: after the exclusive scan it holds the voxel count
    }
}

__global__ void __launch_bounds__(256)
k_vox_centroids(const float4* __restrict__ pts, const char* __restrict__ raw, size_t stride, int has_rgb,
                const unsigned int* __restrict__ order, const unsigned int* __restrict__ n_sorted,
                const GridDev* __restrict__ gd, const unsigned int* __restrict__ pos /* scanned flags */,
                char* __restrict__ out, size_t out_stride, unsigned int /*n*/) {
    const GridParams g = gd->g;
    const unsigned int ns = *n_sorted;
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < ns; t += gridDim.x * blockDim.x) {
        const bool head = pos[t + 1] != pos[t];  // exclusive scan of 0/1 flags: a head bumps the next entry
        if (!head) continue;
        const unsigned int key = voxel_id(pts[order[t]], g);
        double sx = 0, sy = 0, sz = 0;
        unsigned int sr = 0, sg = 0, sb = 0, cnt = 0;
        for (unsigned int u = t; u < ns; ++u) {
            const unsigned int i = order[u];
            const float4 p = pts[i];
            if (u != t && voxel_id(p, g) != key) break;
            sx += p.x; sy += p.y; sz += p.z;
            if (has_rgb) {
                const unsigned int c = *reinterpret_cast<const unsigned int*>(raw + (size_t)i * stride + 16);
                sr += (c >> 16) & 0xffu; sg += (c >> 8) & 0xffu; sb += c & 0xffu;
            }
            ++cnt;
        }
        float* o = reinterpret_cast<float*>(out + (size_t)pos[t] * out_stride);
        o[0] = (float)(sx / cnt); o[1] = (float)(sy / cnt); o[2] = (float)(sz / cnt);
        if (out_stride >= 16) o[3] = 1.0f;
        if (has_rgb) {
            // PCL: centroid /= float(count); rgb = int(r) << 16 | int(g) << 8 | int(b)
            const float fc = (float)cnt;
            const int r = (int)((float)sr / fc), gg = (int)((float)sg / fc), b = (int)((float)sb / fc);
            reinterpret_cast<unsigned int*>(o)[4] = ((unsigned int)r << 16) | ((unsigned int)gg << 8) | (unsigned int)b;
        }
    }
}

int voxel_grid(pcc_index* ix, const void* pts, size_t n, size_t stride, int mem, float leaf, int has_rgb,
               void* out, size_t out_stride, size_t* out_n) {
    hipStream_t s = ix->stream;
    // 1. pack + bbox (host round trip: the leaf lattice is sized on the host like PCL does)
    PCC_TRY(ix->vox_a.reserve(n * sizeof(float4)));
    const void* src = pts;
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(ix->q_raw.reserve(n * stride));
        // a strided host view may end with its last row: copy up to the last byte read (x, y, z [, rgb at 16]), not n rows
        PCC_HIP(hipMemcpyAsync(ix->q_raw.p, pts, (n - 1) * stride + (has_rgb ? 20 : 12), hipMemcpyHostToDevice, s));
        src = ix->q_raw.p;
    }
    int nblk = 0;
    PCC_TRY(launch_pack(s, src, n, stride, ix->vox_a.as<float4>(), ix->blk_stats.as<float>(), &nblk));
    float* h_blk = static_cast<float*>(ix->pinned) + 64;
    PCC_HIP(hipMemcpyAsync(h_blk, ix->blk_stats.p, (size_t)nblk * 8 * sizeof(float), hipMemcpyDeviceToHost, s));
    PCC_HIP(hipStreamSynchronize(s));
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    size_t bad = 0;
    for (int b = 0; b < nblk; ++b) {
        unsigned int u;
        memcpy(&u, &h_blk[b * 8], 4);
        bad += u;
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], h_blk[b * 8 + 1 + a]); hi[a] = std::max(hi[a], h_blk[b * 8 + 4 + a]); }
    }
    if (bad == n) return PCC_OK;  // no finite point: empty output
    // 2. PCL's lattice: min_b = int(floor(min_p * inverse_leaf)), div_b = max_b - min_b + 1
    const float inv = 1.0f / leaf;
    GridDev hd;
    memset(&hd, 0, sizeof(hd));
    double cells = 1;
    for (int a = 0; a < 3; ++a) {
        const int mn = (int)std::floor(lo[a] * inv), mx = (int)std::floor(hi[a] * inv);
        hd.g.org[a] = (float)mn;
        hd.g.dim[a] = mx - mn + 1;
        hd.g.ax[a] = a;  // (PCL's voxel index is x fastest, z slowest: the output order depends on it)
        cells *= (double)hd.g.dim[a];
    }
    if (cells > (double)VOX_MAX_CELLS) {
        // PCL itself refuses when the index would overflow int32 ("Leaf size is too small for the input
        // dataset"); this implementation draws the line at 2^26 voxels
        set_error("leaf size %g too small for this cloud: %.0f voxels (limit %u)", leaf, cells, VOX_MAX_CELLS);
        return PCC_ERR_UNSUPPORTED;
    }
    hd.g.h = leaf;
    hd.g.inv_h = inv;
    hd.g.ncells = hd.g.dim[0] * hd.g.dim[1] * hd.g.dim[2];
    hd.voxel = 1;
    hd.n_valid = (unsigned int)(n - bad);
    PCC_TRY(ix->vox_b.reserve(sizeof(GridDev) + (n + 8) * sizeof(unsigned int) * 2));
    GridDev* d_gd = ix->vox_b.as<GridDev>();
    unsigned int* order = reinterpret_cast<unsigned int*>(d_gd + 1) + 4;
    unsigned int* flags = order + n + 4;
    PCC_HIP(hipMemcpyAsync(d_gd, &hd, sizeof(hd), hipMemcpyHostToDevice, s));
    // 3. sort point indices by voxel
    unsigned int* n_sorted = nullptr;
    PCC_TRY(cell_sort(ix, ix->vox_a.as<float4>(), n, false, nullptr, order, nullptr, &n_sorted, d_gd, (unsigned int)hd.g.ncells));
    // 4. voxel heads -> output slots
    int g1 = (int)std::min<size_t>((n + 256) / 256, 4096);
    hipLaunchKernelGGL(k_vox_heads, dim3(g1), dim3(256), 0, s, ix->vox_a.as<float4>(), order, n_sorted, d_gd, flags, (unsigned int)n);
    PCC_HIP(hipGetLastError());
    PCC_TRY(launch_exclusive_scan(ix, s, flags, n + 1, ix->vox_c));
    unsigned int* h_cnt = static_cast<unsigned int*>(ix->pinned) + 48;
    PCC_HIP(hipMemcpyAsync(h_cnt, flags + n, 4, hipMemcpyDeviceToHost, s));
    // 5. centroids
    char* dout = static_cast<char*>(out);
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(ix->scratch_f.reserve(n * out_stride));
        PCC_HIP(hipMemsetAsync(ix->scratch_f.p, 0, n * out_stride, s));
        dout = static_cast<char*>(ix->scratch_f.p);
    }
    hipLaunchKernelGGL(k_vox_centroids, dim3(g1), dim3(256), 0, s, ix->vox_a.as<float4>(), static_cast<const char*>(src), stride,
                       has_rgb, order, n_sorted, d_gd, flags, dout, out_stride, (unsigned int)n);
    PCC_HIP(hipGetLastError());
    PCC_HIP(hipStreamSynchronize(s));
    const size_t nv = h_cnt[0];
    *out_n = nv;
    if (mem == PCC_MEM_HOST && nv) {
        PCC_HIP(hipMemcpyAsync(out, dout, nv * out_stride, hipMemcpyDeviceToHost, s));
        PCC_HIP(hipStreamSynchronize(s));
    }
    return PCC_OK;
}

}  // namespace pcc
