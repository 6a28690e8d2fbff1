// cellsort.hip -- two-level counting sort of points by grid cell, built on LDS atomics (gfx950).
//
// The index build (the role of KDTreeSingleIndex::buildIndex, reference src/comparator.cpp:565)
// and the per-call query ordering both sort points by linear cell id.  One returning
// device-scope atomic per point is what the first version did, and that runs at the chip's
// memory-side atomic rate: 1M scattered atomics = 47 us (1.36 TB/s of 64-byte atomic requests,
// MI355X_MICROARCH.md "Global float atomics": ~1.3 TB/s chip-wide).  This version never issues a
// global atomic:
//   pass 1  k_cs_hist     each workgroup histograms its slice over B coarse buckets (bucket = F
//                         consecutive cells) in LDS; the LDS atomic's return value is the point's
//                         rank inside (workgroup, bucket).  Histogram rows go out bucket-major.
//   scan                  exclusive scan of the B x G matrix -> base of every (bucket, workgroup)
//   pass 2  k_cs_scatter  point -> base + rank: all points of one coarse bucket become contiguous
//   pass 3  k_cs_fine     one workgroup per coarse bucket: LDS histogram over its F cells, LDS
//                         scan (= cell_start for those cells), LDS cursor scatter to final order
// Order inside a cell is arbitrary (LDS atomics race), which no consumer depends on: ties are
// resolved by the explicit (d2, position) key.
#include "pcc_internal.hpp"
#include "grid_device.hpp"
#include "lane_ops.hpp"

namespace pcc {

constexpr int CS_T = 256;
constexpr unsigned int CS_SLICE = 4096;    // points per workgroup in passes 1 and 2
constexpr unsigned int CS_MAX_G = 512;     // workgroups (slices grow beyond 2M points; measured best at 10M)
constexpr unsigned int CS_BUCKETS = 2048;  // coarse buckets aimed for
constexpr unsigned int CS_MAX_F = 24576;   // cells per bucket: up to 96 KiB of LDS counters (10M+ points)
constexpr unsigned int CS_FINE_STAGE = 1024;  // points a pass-3 bucket places in LDS before writing them out in whole lines

struct CsPlan {
    unsigned int F, B, G, slice;
};
static CsPlan cs_plan(unsigned int ncells, unsigned int n) {
    CsPlan p;
    const unsigned int buckets = CS_BUCKETS, max_f = CS_MAX_F, max_g = CS_MAX_G;
    p.F = (ncells + buckets - 1) / buckets;
    if (p.F < 64) p.F = 64;
    if (p.F > max_f) p.F = max_f;
    p.B = (ncells + p.F - 1) / p.F;
    p.G = (n + CS_SLICE - 1) / CS_SLICE;
    if (p.G > max_g) p.G = max_g;
    if (p.G < 1) p.G = 1;
    p.slice = (n + p.G - 1) / p.G;
    return p;
}

// pass 1 ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(CS_T)
k_cs_hist(const float4* __restrict__ pts, unsigned int n, const GridDev* __restrict__ gd, unsigned int F, unsigned int B,
          unsigned int slice, uint2* __restrict__ key_rank, unsigned int* __restrict__ H) {
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
    const GridParams g = gd->g;
    const bool voxel = gd->voxel != 0;
    for (unsigned int b = threadIdx.x; b < B; b += CS_T) lds[b] = 0;
    __syncthreads();
    const unsigned int beg = blockIdx.x * slice, end = min(n, beg + slice);
    // four independent loads in flight per thread: the loop is latency-bound otherwise (each
    // iteration is load -> LDS atomic -> store)
    for (unsigned int i0 = beg + threadIdx.x; i0 < end; i0 += 4 * CS_T) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * CS_T < end) v[u] = pts[i0 + u * CS_T];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned int i = i0 + u * CS_T;
            if (i >= end) break;
            if (__float_as_int(v[u].w) < 0) { key_rank[i] = make_uint2(0xffffffffu, 0u); continue; }  // non-finite point: not indexed
            const unsigned int c = voxel ? voxel_id(v[u], g) : cell_id(v[u], g);
            key_rank[i] = make_uint2(c, atomicAdd(&lds[c / F], 1u));
        }
    }
    __syncthreads();
    for (unsigned int b = threadIdx.x; b < B; b += CS_T) H[(size_t)b * gridDim.x + blockIdx.x] = lds[b];
    if (blockIdx.x == 0 && threadIdx.x == 0) H[(size_t)B * gridDim.x] = 0;  // slot for the grand total
}

// pass 2 ---------------------------------------------------------------------------------------
// Only (cell, point index) pairs move here: scattering the 16-byte payload too costs a partial
// 64-byte line write per point (31 us per 1M points measured); pass 3 gathers it instead.
__global__ void __launch_bounds__(CS_T)
k_cs_scatter(unsigned int n, unsigned int F, unsigned int slice, const uint2* __restrict__ key_rank,
             const unsigned int* __restrict__ H, uint2* __restrict__ tmp_kv) {
    const unsigned int beg = blockIdx.x * slice, end = min(n, beg + slice);
    for (unsigned int i0 = beg + threadIdx.x; i0 < end; i0 += 4 * CS_T) {
        uint2 kr[4];
        unsigned int base[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kr[u] = (i0 + u * CS_T < end) ? key_rank[i0 + u * CS_T] : make_uint2(0xffffffffu, 0u);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            base[u] = kr[u].x != 0xffffffffu ? H[(size_t)(kr[u].x / F) * gridDim.x + blockIdx.x] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (kr[u].x != 0xffffffffu) tmp_kv[base[u] + kr[u].y] = make_uint2(kr[u].x, i0 + u * CS_T);
    }
}

// pass 3 ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int block_excl_scan_256(unsigned int v, unsigned int* wsum /* 4 LDS words */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int inc = wave_incl_scan_add(v);  // DPP, no LDS crossbar
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    __syncthreads();
    return base + inc - v;
}

template <bool REFS>
__global__ void __launch_bounds__(CS_T)
k_cs_fine(const GridDev* __restrict__ gd, unsigned int F, unsigned int G, const unsigned int* __restrict__ H,
          const float4* __restrict__ pts, const uint2* __restrict__ tmp_kv,
          float4* __restrict__ out_pts, unsigned int* __restrict__ out_order,
          unsigned int* __restrict__ cell_start) {
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];  // F counters + 4 scan words
    unsigned int* cnt = lds;
    unsigned int* wsum = lds + F;
    const unsigned int ncells = (unsigned int)gd->g.ncells;  // actual cell count (<= the host's nc_cap)
    const unsigned int b = blockIdx.x;
    const unsigned int beg = H[(size_t)b * G];
    const unsigned int end = H[(size_t)(b + 1) * G];  // next bucket's first base; the last one reads the grand total
    const unsigned int cell0 = b * F;
    for (unsigned int f = threadIdx.x; f < F; f += CS_T) cnt[f] = 0;
    __syncthreads();
    for (unsigned int j0 = beg + threadIdx.x; j0 < end; j0 += 4 * CS_T) {
        unsigned int c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = (j0 + u * CS_T < end) ? tmp_kv[j0 + u * CS_T].x : 0xffffffffu;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c[u] != 0xffffffffu) atomicAdd(&cnt[c[u] - cell0], 1u);
    }
    __syncthreads();
    // exclusive scan of the F counters: thread t owns a contiguous chunk
    const unsigned int per = (F + CS_T - 1) / CS_T;
    const unsigned int f0 = threadIdx.x * per, f1 = min(F, f0 + per);
    unsigned int s = 0;
    for (unsigned int f = f0; f < f1; ++f) s += cnt[f];
    unsigned int run = beg + block_excl_scan_256(s, wsum);
    for (unsigned int f = f0; f < f1; ++f) {
        const unsigned int c = cnt[f];
        cnt[f] = run;  // becomes the cursor of cell f
        if (REFS && cell0 + f <= ncells) cell_start[cell0 + f] = run;  // cells past ncells are empty: entry
        run += c;                                                      // [ncells] receives the total
    }
    __syncthreads();
    // a bucket of up to CS_FINE_STAGE points (the usual case) is placed in LDS and leaves in whole lines
    const bool staged = out_pts && end - beg <= CS_FINE_STAGE;
    float4* st_pts = reinterpret_cast<float4*>(lds + ((F + 4 + 3) & ~3u));
    for (unsigned int j0 = beg + threadIdx.x; j0 < end; j0 += 4 * CS_T) {
        uint2 kv[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kv[u] = (j0 + u * CS_T < end) ? tmp_kv[j0 + u * CS_T] : make_uint2(0xffffffffu, 0u);
        if (out_pts) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (kv[u].x != 0xffffffffu) v[u] = pts[kv[u].y];  // gather from the original-order array
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (kv[u].x == 0xffffffffu) continue;
            const unsigned int pos = atomicAdd(&cnt[kv[u].x - cell0], 1u);
            if (out_pts) {
                v[u].w = __int_as_float((int)kv[u].y);  // cell-sorted copies carry the packed position
                if (staged) st_pts[pos - beg] = v[u];
                else out_pts[pos] = v[u];
            }
            if (out_order) out_order[pos] = kv[u].y;
        }
    }
    if (staged) {
        __syncthreads();
        for (unsigned int i = threadIdx.x; i < end - beg; i += CS_T) out_pts[beg + i] = st_pts[i];
    }
}

// ---- host side ------------------------------------------------------------------------------------
// Sorts pts[0..n) by cell.  refs: out_pts (float4, w = position) + cell_start[ncells+1].
// queries: out_order[0..n_sorted) lists the valid queries cell by cell; *n_sorted_dev = device address of the count.
int cell_sort(pcc_index* ix, const float4* pts, size_t n_pts, bool refs, float4* out_pts, unsigned int* out_order,
              unsigned int* cell_start, unsigned int** n_sorted_dev, const GridDev* gd_override,
              unsigned int nc_cap_override) {
    // clouds past the L2s: three coalesced levels with the payload carried along (cellsort_mp.hip)
    // measured crossovers (MI355X, corridor scene; us, this file vs cellsort_mp): reference clouds 1M 106 / 138,
    // 2M 176 / 152, 4M 322 / 214, 10M 779 / 495; query clouds (order only) 2M 85 / 103, 4M 136 / 133, 10M 340 / 262
    const size_t mp_min = (size_t)ix->opt.sort_mp_min, mp_min_q = (size_t)ix->opt.sort_mp_min_q;
    if (n_pts >= (refs ? mp_min : mp_min_q)) return cell_sort_mp(ix, pts, n_pts, refs, out_pts, out_order, cell_start, n_sorted_dev, gd_override, nc_cap_override);
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)n_pts;
    // planned from the host-known upper bound of the cell count; buckets past the actual grid stay empty
    const CsPlan p = cs_plan((gd_override ? nc_cap_override : ix->nc_cap) + 1, n);
    const GridDev* gd = gd_override ? gd_override : ix->d_grid.as<GridDev>();
    // scratch: key_rank[n] | tmp_kv[n] | H[B*G+1] + scan scratch
    DevBuf& kr_buf = ix->scratch_c;
    DevBuf& kv_buf = ix->scratch_e;
    DevBuf& h_buf = ix->scratch_b;
    PCC_TRY(kr_buf.reserve((size_t)n * sizeof(uint2) + 256));
    PCC_TRY(kv_buf.reserve((size_t)n * sizeof(uint2) + 256));
    const size_t h_elems = (size_t)p.B * p.G + 1;
    PCC_TRY(h_buf.reserve(((h_elems + 3) & ~(size_t)3) * sizeof(unsigned int)));
    uint2* key_rank = kr_buf.as<uint2>();
    uint2* tmp_kv = kv_buf.as<uint2>();
    unsigned int* H = h_buf.as<unsigned int>();
    hipLaunchKernelGGL(k_cs_hist, dim3(p.G), dim3(CS_T), p.B * sizeof(unsigned int), s, pts, n, gd, p.F, p.B, p.slice,
                       key_rank, H);
    PCC_HIP(hipGetLastError());
    PCC_TRY(launch_exclusive_scan(ix, s, H, h_elems, ix->scratch_a));
    if (n_sorted_dev) *n_sorted_dev = H + h_elems - 1;  // grand total == number of valid points
    const size_t lds3 = (((size_t)p.F + 4 + 3) & ~(size_t)3) * sizeof(unsigned int) + (out_pts ? (size_t)CS_FINE_STAGE * sizeof(float4) : 0);
    hipLaunchKernelGGL(k_cs_scatter, dim3(p.G), dim3(CS_T), 0, s, n, p.F, p.slice, key_rank, H, tmp_kv);
    if (refs)
        hipLaunchKernelGGL((k_cs_fine<true>), dim3(p.B), dim3(CS_T), lds3, s, gd, p.F, p.G, H, pts,
                           tmp_kv, out_pts, out_order, cell_start);
    else
        hipLaunchKernelGGL((k_cs_fine<false>), dim3(p.B), dim3(CS_T), lds3, s, gd, p.F, p.G, H, pts,
                           tmp_kv, out_pts, out_order, cell_start);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

}  // namespace pcc
