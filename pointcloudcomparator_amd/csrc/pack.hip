// pack.hip -- streaming helper kernels of libpcc_nn (gfx950): AoS -> float4 pack,
// order-preserving compaction, exclusive scan, result unpack, rigid transform.
// All of these are HBM-bound single-pass kernels: 16-byte accesses per lane,
// grid capped and grid-strided (cdna_hip_programming.md Guideline 11/13).
#include <algorithm>

#include "pcc_internal.hpp"
#include "lane_ops.hpp"
#include "grid_device.hpp"
#include "grid_params_device.hpp"

namespace pcc {

static inline int grid_for(size_t n, int block, int per_thread = 1) {
    size_t b = (n + (size_t)block * per_thread - 1) / ((size_t)block * per_thread);
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (int)b;
}

__device__ __forceinline__ bool finite3(float x, float y, float z) {
    // all three finite <=> none is NaN/Inf; (v - v) == 0 only for finite v
    return (x - x) == 0.0f && (y - y) == 0.0f && (z - z) == 0.0f;
}

// ---- pack -------------------------------------------------------------------
// replaces pcl::KdTreeFLANN::convertCloudToArray's copy loop: the
// first three floats of every element, invalid points flagged for the compaction.
// Per-workgroup partial results instead of global atomics: blk[b*8 + 0] = invalid points seen
// by workgroup b, blk[b*8 + 1..3] = min xyz, blk[b*8 + 4..6] = max xyz of its valid points
// (floats; +inf/-inf when it saw none).  The host reduces the <= PACK_MAX_BLOCKS rows it
// reads back.  (Atomics on six shared words serialise: 0.56 ms for 1M points, measured.)
// cells (query clouds, when the index holds a grid): the grid cell of every point, ~0 for a non-finite one, written beside the
// packed point -- the first level of the three-level sort then reads 4 bytes per query instead of the 16-byte point twice
// (histogram and scatter: 320 MB of 10M queries' 560)
template <bool VEC16, bool STATS>
__global__ void __launch_bounds__(256)
k_pack(const char* __restrict__ aos, size_t n, size_t stride, float4* __restrict__ out,
       float* __restrict__ blk, unsigned int* __restrict__ zero_word, float4* __restrict__ seeds,
       unsigned long long* __restrict__ invalid_keys, unsigned int* __restrict__ cells, const GridDev* __restrict__ gd,
       PackGrid pg) {
    if (zero_word && blockIdx.x == 0 && threadIdx.x < 64) {  // counters of the search that follows: saves a 5 us memset node
        if (threadIdx.x < 2) zero_word[threadIdx.x] = 0u;                      // fallback list, far list
        zero_word[PCC_OPEN_CTR0 - 32 + threadIdx.x * PCC_OPEN_CTR_STRIDE] = 0u;  // the sharded open-lane counters (grid.hip)
    }
    unsigned int bad = 0;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    GridParams g;
    if (!STATS && cells) g = gd->g;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        float x, y, z;
        if (VEC16) {
            float4 v = *reinterpret_cast<const float4*>(aos + i * stride);
            x = v.x; y = v.y; z = v.z;
        } else {
            const float* p = reinterpret_cast<const float*>(aos + i * stride);
            x = p[0]; y = p[1]; z = p[2];
        }
        float4 o;
        if (finite3(x, y, z)) {
            o = make_float4(x, y, z, __int_as_float((int)i));
            if (STATS) {
                lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
                lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
                lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
            }
        } else {
            o = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            ++bad;
            // query clouds: the result key of a non-finite query is "nothing found" from the start (the searches never
            // touch it, and the unpack kernel then needs no second look at the queries)
            if (invalid_keys) invalid_keys[i] = ~0ull;
        }
        out[i] = o;
        if (!STATS && cells) cells[i] = __float_as_int(o.w) >= 0 ? cell_id(o, g) : 0xffffffffu;
        if (STATS && seeds && (i & (PCC_SEED_STRIDE - 1)) == 0) seeds[i >> PCC_SEED_SHIFT] = o;  // upper bounds for far queries
    }
    if (STATS) {
        __shared__ float red[4][8];
        for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            for (int off = 32; off > 0; off >>= 1) {
                lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
                hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
            }
        }
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            red[wave][0] = __uint_as_float(bad);
            for (int a = 0; a < 3; ++a) { red[wave][1 + a] = lo[a]; red[wave][4 + a] = hi[a]; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int b = 0;
            float l[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
            float h[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            for (int w = 0; w < 4; ++w) {
                b += __float_as_uint(red[w][0]);
                for (int a = 0; a < 3; ++a) { l[a] = fminf(l[a], red[w][1 + a]); h[a] = fmaxf(h[a], red[w][4 + a]); }
            }
            float* o = blk + (size_t)blockIdx.x * 8;
            o[0] = __uint_as_float(b);
            for (int a = 0; a < 3; ++a) { o[1 + a] = l[a]; o[4 + a] = h[a]; }
            o[7] = 0.f;
        }
        // The grid of the index straight from here (PackGrid, set by the index build): every workgroup takes a ticket once its
        // row is written (release: fence, then the agent-scope increment); the LAST one to arrive does what k_grid_params did in
        // a launch of its own -- 10-12 us of every build, a twentieth of a 1M x 1M step -- reading the rows past the caches.
        if (pg.out) {
            __shared__ unsigned int last;
            if (threadIdx.x == 0) {
                __threadfence();
                last = atomicAdd(pg.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
            }
            __syncthreads();
            if (last) {  // (block-uniform)
                __threadfence();
                if (threadIdx.x == 0) *pg.ticket = 0u;  // ready for the next build on this handle (stream order)
                grid_params_block<true>(blk, (int)gridDim.x, (unsigned int)n, pg.ppc, pg.nc_cap, pg.trim_k, pg.axes, pg.out, pg.host_mirror);
            }
        }
    }
}

int launch_pack(hipStream_t s, const void* aos, size_t n, size_t stride, float4* out,
                float* blk_stats, int* n_blocks, unsigned int* zero_word, float4* seeds, unsigned long long* invalid_keys,
                unsigned int* cells, const GridDev* gd, const PackGrid* grid) {
    if (n_blocks) *n_blocks = 0;
    if (n == 0) return PCC_OK;
    bool vec = (stride % 16 == 0) && ((reinterpret_cast<uintptr_t>(aos) & 15) == 0);
    int g = grid_for(n, 256, 2);
    if (g > PACK_MAX_BLOCKS) g = PACK_MAX_BLOCKS;
    if (n_blocks) *n_blocks = g;
    const char* a = (const char*)aos;
    bool st = blk_stats != nullptr;
    PackGrid pg{};
    if (grid && st) {
        pg = *grid;
        if (g < 128) pg.trim_k = 0;  // (trimming needs enough rows to tell an outlier from the scene: 128 pack workgroups = 64k points)
    }
    if (vec && st) hipLaunchKernelGGL((k_pack<true, true>), dim3(g), dim3(256), 0, s, a, n, stride, out, blk_stats, zero_word, seeds, invalid_keys, cells, gd, pg);
    else if (vec) hipLaunchKernelGGL((k_pack<true, false>), dim3(g), dim3(256), 0, s, a, n, stride, out, blk_stats, zero_word, seeds, invalid_keys, cells, gd, pg);
    else if (st) hipLaunchKernelGGL((k_pack<false, true>), dim3(g), dim3(256), 0, s, a, n, stride, out, blk_stats, zero_word, seeds, invalid_keys, cells, gd, pg);
    else hipLaunchKernelGGL((k_pack<false, false>), dim3(g), dim3(256), 0, s, a, n, stride, out, blk_stats, zero_word, seeds, invalid_keys, cells, gd, pg);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

// ---- exclusive scan (uint32, in place) ------------------------------------------
constexpr int SCAN_T = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = SCAN_T * SCAN_ITEMS;  // 2048 elements per workgroup

__device__ __forceinline__ unsigned int block_exclusive_scan(unsigned int v, unsigned int* total) {
    // exclusive scan of one value per thread over a 256-thread workgroup
    __shared__ unsigned int wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int inc = wave_incl_scan_add(v);  // DPP, no LDS crossbar
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    unsigned int r = base + inc - v;
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(SCAN_T)
k_scan_reduce(const unsigned int* __restrict__ data, size_t n, unsigned int* __restrict__ bsum) {
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_ITEMS;
    unsigned int s = 0;
    if (base + SCAN_ITEMS <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(data + base);
        uint4 a = p[0], b = p[1];
        s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    } else {
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (base + k < n) s += data[base + k];
    }
    unsigned int tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_T)
k_scan_apply(unsigned int* __restrict__ data, size_t n, const unsigned int* __restrict__ bprefix) {
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_ITEMS;
    unsigned int v[SCAN_ITEMS];
    bool full = base + SCAN_ITEMS <= n;
    if (full) {
        const uint4* p = reinterpret_cast<const uint4*>(data + base);
        uint4 a = p[0], b = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = (base + k < n) ? data[base + k] : 0u;
    }
    unsigned int s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) s += v[k];
    unsigned int off = block_exclusive_scan(s, nullptr) + (bprefix ? bprefix[blockIdx.x] : 0u);
    unsigned int o[SCAN_ITEMS];
    for (int k = 0; k < SCAN_ITEMS; ++k) { o[k] = off; off += v[k]; }
    if (full) {
        uint4* p = reinterpret_cast<uint4*>(data + base);
        p[0] = make_uint4(o[0], o[1], o[2], o[3]);
        p[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (base + k < n) data[base + k] = o[k];
    }
}

// second level fused into the apply pass: every workgroup sums the block totals in front of it
// itself (nb <= SCAN_FUSED_MAX values from L2) instead of waiting for a separate scan launch --
// one dependent launch (~5 us on this chip) less per scan
constexpr size_t SCAN_FUSED_MAX = 8192;
__global__ void __launch_bounds__(SCAN_T)
k_scan_apply_fused(unsigned int* __restrict__ data, size_t n, const unsigned int* __restrict__ bsum) {
    __shared__ unsigned int part[4];
    unsigned int pre = 0;
    for (unsigned int b = threadIdx.x; b < blockIdx.x; b += SCAN_T) pre += bsum[b];
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_down(pre, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = pre;
    __syncthreads();
    const unsigned int prefix = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_ITEMS;
    unsigned int v[SCAN_ITEMS];
    bool full = base + SCAN_ITEMS <= n;
    if (full) {
        const uint4* p = reinterpret_cast<const uint4*>(data + base);
        uint4 a = p[0], b = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = (base + k < n) ? data[base + k] : 0u;
    }
    unsigned int s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) s += v[k];
    unsigned int off = block_exclusive_scan(s, nullptr) + prefix;
    unsigned int o[SCAN_ITEMS];
    for (int k = 0; k < SCAN_ITEMS; ++k) { o[k] = off; off += v[k]; }
    if (full) {
        uint4* p = reinterpret_cast<uint4*>(data + base);
        p[0] = make_uint4(o[0], o[1], o[2], o[3]);
        p[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (base + k < n) data[base + k] = o[k];
    }
}

// The scan in ONE launch (round 5): every workgroup scans its 2048 words, publishes its total in a word of its own tagged with
// the launch's epoch, and adds up the totals of the workgroups in front of it as they appear -- the two-launch form above
// costs a dependent launch (~5 us) per scan, and a 10M-point step runs four to six scans over 32k-131k counters.  The totals
// travel as single 64-bit agent-scope atomics (epoch << 32 | total): no fence, no flag / value pair to order.  A workgroup
// only ever waits for workgroups with LOWER ids, which the dispatcher starts first and which wait for nothing behind them, so
// the chain cannot deadlock whatever else occupies the chip.  `flags` belongs to the handle and is used for nothing else
// (a stale word never carries a future epoch).
constexpr size_t SCAN_CHAIN_MAX = 512;
__global__ void __launch_bounds__(SCAN_T)
k_scan_chained(unsigned int* __restrict__ data, size_t n, unsigned long long* __restrict__ flags, unsigned int epoch) {
    __shared__ unsigned int part[4];
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_ITEMS;
    unsigned int v[SCAN_ITEMS];
    const bool full = base + SCAN_ITEMS <= n;
    if (full) {
        const uint4* p = reinterpret_cast<const uint4*>(data + base);
        uint4 a = p[0], b = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = (base + k < n) ? data[base + k] : 0u;
    }
    unsigned int sum = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) sum += v[k];
    unsigned int tot;
    unsigned int off = block_exclusive_scan(sum, &tot);
    if (threadIdx.x == 0)
        __hip_atomic_store(&flags[blockIdx.x], ((unsigned long long)epoch << 32) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int pre = 0;
    for (unsigned int b = threadIdx.x; b < blockIdx.x; b += SCAN_T) {
        unsigned long long w;
        do {
            w = __hip_atomic_load(&flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((unsigned int)(w >> 32) != epoch);
        pre += (unsigned int)w;
    }
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_down(pre, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = pre;
    __syncthreads();
    off += part[0] + part[1] + part[2] + part[3];
    unsigned int o8[SCAN_ITEMS];
    for (int k = 0; k < SCAN_ITEMS; ++k) { o8[k] = off; off += v[k]; }
    if (full) {
        uint4* p = reinterpret_cast<uint4*>(data + base);
        p[0] = make_uint4(o8[0], o8[1], o8[2], o8[3]);
        p[1] = make_uint4(o8[4], o8[5], o8[6], o8[7]);
    } else {
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (base + k < n) data[base + k] = o8[k];
    }
}

static int scan_rec(hipStream_t s, unsigned int* data, size_t n, unsigned int* tmp, size_t tmp_elems) {
    size_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (nb > 1 && nb <= SCAN_FUSED_MAX && tmp_elems >= nb) {
        hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_T), 0, s, data, n, tmp);
        hipLaunchKernelGGL(k_scan_apply_fused, dim3((unsigned)nb), dim3(SCAN_T), 0, s, data, n, tmp);
        PCC_HIP(hipGetLastError());
        return PCC_OK;
    }
    if (nb <= 1) {
        hipLaunchKernelGGL(k_scan_apply, dim3(1), dim3(SCAN_T), 0, s, data, n, (const unsigned int*)nullptr);
        PCC_HIP(hipGetLastError());
        return PCC_OK;
    }
    if (tmp_elems < nb) { set_error("scan scratch too small"); return PCC_ERR_INVALID; }
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_T), 0, s, data, n, tmp);
    PCC_HIP(hipGetLastError());
    size_t nb_al = (nb + 3) & ~(size_t)3;  // keep the next level 16-byte aligned
    PCC_TRY(scan_rec(s, tmp, nb, tmp + nb_al, tmp_elems - nb_al));
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_T), 0, s, data, n, tmp);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

int launch_exclusive_scan(pcc_index* ix, hipStream_t s, unsigned int* data, size_t n, DevBuf& tmp) {
    if (n == 0) return PCC_OK;
    const size_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (ix && ix->opt.scan_chained && nb > 1 && nb <= SCAN_CHAIN_MAX) {
        if (!ix->scan_flags.p) {  // (zeroed once: epochs start at 1)
            PCC_TRY(ix->scan_flags.reserve(SCAN_CHAIN_MAX * sizeof(unsigned long long)));
            PCC_HIP(hipMemsetAsync(ix->scan_flags.p, 0, SCAN_CHAIN_MAX * sizeof(unsigned long long), s));
            ix->scan_epoch = 0;
        }
        if (++ix->scan_epoch == 0u) {  // (4G scans later: start over from clean words)
            PCC_HIP(hipMemsetAsync(ix->scan_flags.p, 0, SCAN_CHAIN_MAX * sizeof(unsigned long long), s));
            ix->scan_epoch = 1;
        }
        hipLaunchKernelGGL(k_scan_chained, dim3((unsigned)nb), dim3(SCAN_T), 0, s, data, n, ix->scan_flags.as<unsigned long long>(), ix->scan_epoch);
        PCC_HIP(hipGetLastError());
        return PCC_OK;
    }
    size_t elems = 0;
    for (size_t k = (n + SCAN_BLOCK - 1) / SCAN_BLOCK; k > 1; k = (k + SCAN_BLOCK - 1) / SCAN_BLOCK)
        elems += ((k + 3) & ~(size_t)3);
    elems += 16;
    PCC_TRY(tmp.reserve(elems * sizeof(unsigned int)));
    return scan_rec(s, data, n, tmp.as<unsigned int>(), elems);
}

// ---- unpack -------------------------------------------------------------------------
// search keys hold (d2 bits << 32 | position); non-finite references stay in place (flagged),
// so a position IS the original index (PCL's index_mapping_ is the identity here).
__global__ void __launch_bounds__(256)
k_unpack(const unsigned long long* __restrict__ packed, const float4* __restrict__ q, size_t n,
         int32_t* __restrict__ idx, float* __restrict__ d2, const unsigned int* __restrict__ mirror_dev,
         unsigned int* __restrict__ mirror_host) {
    if (mirror_dev && blockIdx.x == 0 && threadIdx.x == 0) *mirror_host = *mirror_dev;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long p = packed[i];
        bool ok = (!q || __float_as_int(q[i].w) >= 0) && !key_none(p);
        if (idx) idx[i] = ok ? (int32_t)(unsigned int)(p & 0xffffffffull) : -1;
        if (d2) d2[i] = ok ? __uint_as_float((unsigned int)(p >> 32)) : __builtin_inff();
    }
}
int launch_unpack(hipStream_t s, const unsigned long long* packed, const float4* q, size_t n,
                  int32_t* idx, float* d2, const unsigned int* mirror_dev, unsigned int* mirror_host) {
    if (n == 0) return PCC_OK;
    hipLaunchKernelGGL(k_unpack, dim3(grid_for(n, 256)), dim3(256), 0, s, packed, q, n, idx, d2, mirror_dev, mirror_host);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

// ---- rigid transform ------------------------------------------------------------------
// pcl::transformPointCloud arithmetic: ((m0*x + m1*y) + m2*z) + m3, every
// op rounded separately (this file is compiled with -ffp-contract=off).
struct Mat34 { float m[12]; };
__global__ void __launch_bounds__(256)
k_transform(Mat34 T, const float* __restrict__ Tdev, const char* __restrict__ src, size_t n,
            size_t sstride, char* __restrict__ dst, size_t dstride, unsigned int* __restrict__ zero_word) {
    if (zero_word && blockIdx.x == 0 && threadIdx.x < 64) {  // counters of the search that follows (as k_pack does): an ICP pass
        if (threadIdx.x < 2) zero_word[threadIdx.x] = 0u;     // otherwise spends three 4-us memset nodes on them
        zero_word[PCC_OPEN_CTR0 - 32 + threadIdx.x * PCC_OPEN_CTR_STRIDE] = 0u;
    }
    float m[12];
    for (int k = 0; k < 12; ++k) m[k] = Tdev ? Tdev[k] : T.m[k];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const float* p = reinterpret_cast<const float*>(src + i * sstride);
        float x = p[0], y = p[1], z = p[2];
        float ox = ((m[0] * x + m[1] * y) + m[2] * z) + m[3];
        float oy = ((m[4] * x + m[5] * y) + m[6] * z) + m[7];
        float oz = ((m[8] * x + m[9] * y) + m[10] * z) + m[11];
        float* o = reinterpret_cast<float*>(dst + i * dstride);
        o[0] = ox; o[1] = oy; o[2] = oz;
    }
}
int launch_transform(hipStream_t s, const float* Tdev, const float T[16], const void* src, size_t n,
                     size_t sstride, void* dst, size_t dstride, unsigned int* zero_word) {
    if (n == 0) return PCC_OK;
    Mat34 M;
    for (int k = 0; k < 12; ++k) M.m[k] = T ? T[k] : 0.f;
    hipLaunchKernelGGL(k_transform, dim3(grid_for(n, 256)), dim3(256), 0, s, M, Tdev, (const char*)src, n,
                       sstride, (char*)dst, dstride, zero_word);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}


__global__ void __launch_bounds__(256)
k_nanify(float4* __restrict__ pts, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = pts[i];
        if (__float_as_int(v.w) < 0) { v.x = v.y = v.z = __builtin_nanf(""); pts[i] = v; }
    }
}
int launch_nanify(hipStream_t s, float4* pts, size_t n) {
    if (n == 0) return PCC_OK;
    hipLaunchKernelGGL(k_nanify, dim3(grid_for(n, 256)), dim3(256), 0, s, pts, n);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

__global__ void __launch_bounds__(256)
k_copy_w(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i].w = src[i].w;
}
// The ICP loop's working set in CELL order (round 5): dst[t] = q[order[t]] for the n_sorted valid points, the rest
// flagged invalid; *ns_word = n_sorted (a word that outlives the sort's scratch).  One gather, once per pcc_icp_align: every
// pass after it reads its queries front to back and stores its keys where it found them.
__global__ void __launch_bounds__(256)
k_gather_sorted(const float4* __restrict__ q, const unsigned int* __restrict__ order, const unsigned int* __restrict__ n_sorted,
                size_t n, float4* __restrict__ dst, unsigned int* __restrict__ ns_word) {
    const unsigned int ns = *n_sorted;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ns_word = ns;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x)
        dst[t] = t < ns ? q[order[t]] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
}
int launch_gather_sorted(hipStream_t s, const float4* q, const unsigned int* order, const unsigned int* n_sorted, size_t n, float4* dst,
                         unsigned int* ns_word) {
    if (n == 0) return PCC_OK;
    hipLaunchKernelGGL(k_gather_sorted, dim3(grid_for(n, 256)), dim3(256), 0, s, q, order, n_sorted, n, dst, ns_word);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

int launch_copy_w(hipStream_t s, const float4* src, float4* dst, size_t n) {
    if (n == 0) return PCC_OK;
    hipLaunchKernelGGL(k_copy_w, dim3(grid_for(n, 256)), dim3(256), 0, s, src, dst, n);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

// pcl::StatisticalOutlierRemoval::applyFilterIndices inner loop: neighbour 0 is the
// point itself; dist_sum (double) += sqrt(d2_j) for j = 1..mean_k; distances[i] = float(dist_sum / mean_k)
__global__ void __launch_bounds__(256)
k_sor_mean(const unsigned long long* __restrict__ keys, const float4* __restrict__ /*refs*/, size_t n, int K,
           float* __restrict__ mean_dist) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long* row = keys + i * (size_t)K;
        if (key_none(row[K - 1])) continue;  // invalid point or fewer than K neighbours: distance stays 0
        double s = 0.0;
        for (int j = 1; j < K; ++j) s += sqrt((double)__uint_as_float((unsigned int)(row[j] >> 32)));
        mean_dist[i] = (float)(s / (double)(K - 1));
    }
}
// the same through LDS: a wave's 64 rows are contiguous in memory, so it streams them with fully coalesced
// loads (lanes over the flat key index), parks the d2 words in LDS and every lane then sums its own row in
// order.  Lane-per-row reads touched 64 different lines per load: 0.8 ms at 1M x 51 against 0.4 GB of keys.
// (E = unsigned long long: rows of search keys; E = unsigned int: rows of squared distances as the k-NN kernels deliver
// them when nobody needs the indices -- half the bytes written and read)
template <class E>
__global__ void __launch_bounds__(256)
k_sor_mean_staged(const E* __restrict__ keys, size_t n, int K, float* __restrict__ mean_dist) {
    extern __shared__ unsigned int sor_tile[];
    const unsigned int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
    unsigned int* tile = sor_tile + (size_t)wave_in_block * 64 * (K + 1);
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t base = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64; base < n; base += nwaves * 64) {
        const unsigned int rows = (unsigned int)min((size_t)64, n - base);
        const unsigned int total = rows * (unsigned int)K;
        const E* src = keys + base * (size_t)K;
        wave_lds_sync();
        for (unsigned int f = lane; f < total; f += 64) {
            const unsigned int r = f / (unsigned int)K, c = f - r * (unsigned int)K;
            if constexpr (sizeof(E) == 8) tile[r * (K + 1) + c] = (unsigned int)(src[f] >> 32);
            else tile[r * (K + 1) + c] = (unsigned int)src[f];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        wave_lds_sync();
        if (lane < rows) {
            const unsigned int* row = tile + lane * (K + 1);
            if (row[K - 1] < 0x7f7fffffu) {  // else: invalid point or fewer than K neighbours (key_none), distance stays 0
                double s = 0.0;
                for (int j = 1; j < K; ++j) s += sqrt((double)__uint_as_float(row[j]));
                mean_dist[base + lane] = (float)(s / (double)(K - 1));
            }
        }
    }
}

// the first k_dst entries of every row of k_src (k-NN rows are ascending: the prefix of a longer row IS the shorter row)
__global__ void __launch_bounds__(256)
k_copy_row_prefix(const unsigned long long* __restrict__ src, int k_src, unsigned long long* __restrict__ dst, int k_dst, size_t total) {
    for (size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (size_t)gridDim.x * blockDim.x) {
        const size_t r = f / (size_t)k_dst;
        dst[f] = src[r * (size_t)k_src + (f - r * (size_t)k_dst)];
    }
}
int launch_copy_row_prefix(hipStream_t s, const unsigned long long* src, int k_src, unsigned long long* dst, int k_dst, size_t n) {
    const size_t total = n * (size_t)k_dst;
    if (total == 0) return PCC_OK;
    hipLaunchKernelGGL(k_copy_row_prefix, dim3(grid_for(total, 256)), dim3(256), 0, s, src, k_src, dst, k_dst, total);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

// keys, or d2_rows (rows of K squared distances, float bits) when keys == nullptr
int launch_sor_mean(hipStream_t s, const unsigned long long* keys, const float4* refs, size_t n, int K,
                    float* mean_dist, const float* d2_rows) {
    if (n == 0) return PCC_OK;
    // four waves per workgroup while their tiles fit 64 KB of LDS (K <= 62; PCL's default mean_k = 50 gives K = 51:
    // 53 KB), two up to K = 126
    const int waves = (size_t)4 * 64 * (K + 1) * sizeof(unsigned int) <= 64 * 1024 ? 4 : 2;
    const size_t lds = (size_t)waves * 64 * (K + 1) * sizeof(unsigned int);
    if (lds <= 64 * 1024) {
        const int bs = waves * 64;
        const int blocks = (int)std::min<size_t>((n + bs - 1) / bs, 4096);
        if (keys) hipLaunchKernelGGL((k_sor_mean_staged<unsigned long long>), dim3(blocks), dim3(bs), lds, s, keys, n, K, mean_dist);
        else hipLaunchKernelGGL((k_sor_mean_staged<unsigned int>), dim3(blocks), dim3(bs), lds, s,
                                reinterpret_cast<const unsigned int*>(d2_rows), n, K, mean_dist);
    } else {
        if (!keys) { set_error("launch_sor_mean: rows of distances need the staged kernel"); return PCC_ERR_INVALID; }
        hipLaunchKernelGGL(k_sor_mean, dim3(grid_for(n, 256)), dim3(256), 0, s, keys, refs, n, K, mean_dist);
    }
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

// ---- SOR statistics on the device ---------------------------------------------------------------------------------
// PCL adds the mean distances up one after the other in double (sum += d; sq_sum += d * d with the product rounded to
// float: pcl/filters/impl/statistical_outlier_removal.hpp).  A sequential chain cannot be parallelised bit for
// bit in general -- but it can whenever no addition of it ROUNDS: every term is a float (24 bits), every partial sum a
// multiple of the smallest term's last bit, so as long as total / (last bit of the smallest positive term) stays below 2^52
// every partial sum of ANY order is exact and all orders give the same double.  The kernels below add in a tree, track the
// smallest positive term of both sums and say whether that condition held (means of centimetres over a million points:
// it does, by ten bits); when it does not -- mean distances of micrometres next to metres -- pcc_sor falls back to the
// in-order host loop.  Threshold and inlier mask follow on the device: no cloud-sized copy, one 48-byte read-back.
struct SorStats {
    double sum, sq, thr;
    unsigned long long kept;
    unsigned int exact, pad;
};
constexpr int SOR_RED_BLOCKS = 1024;
__device__ __forceinline__ double wave_sum_f64(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v) {
    for (int off = 32; off > 0; off >>= 1) v = min(v, (unsigned int)__shfl_down((int)v, off, 64));
    return v;
}
// partial rows: {sum, sq, bits(min positive f) | bits(min positive float(f * f)) << 32}
__global__ void __launch_bounds__(256)
k_sor_reduce(const float* __restrict__ m, size_t n, double* __restrict__ part) {
    double sum = 0.0, sq = 0.0;
    unsigned int fmin = 0x7f800000u, gmin = 0x7f800000u;  // (positive floats order like their bits)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float f = m[i];
        const float g = f * f;  // PCL: distances[i] * distances[i] in float, then widened
        sum += (double)f;
        sq += (double)g;
        if (f > 0.f) fmin = min(fmin, __float_as_uint(f));
        if (g > 0.f) gmin = min(gmin, __float_as_uint(g));
    }
    __shared__ double rs[4], rq[4];
    __shared__ unsigned int rf[4], rg[4];
    sum = wave_sum_f64(sum);
    sq = wave_sum_f64(sq);
    fmin = wave_min_u32(fmin);
    gmin = wave_min_u32(gmin);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = sum; rq[threadIdx.x >> 6] = sq; rf[threadIdx.x >> 6] = fmin; rg[threadIdx.x >> 6] = gmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = ((rs[0] + rs[1]) + rs[2]) + rs[3];
        part[blockIdx.x * 3 + 1] = ((rq[0] + rq[1]) + rq[2]) + rq[3];
        const unsigned long long mm = (unsigned long long)min(min(rf[0], rf[1]), min(rf[2], rf[3])) |
                                      ((unsigned long long)min(min(rg[0], rg[1]), min(rg[2], rg[3])) << 32);
        part[blockIdx.x * 3 + 2] = __longlong_as_double((long long)mm);
    }
}
// rows -> {sum, sq, bits(min positive term of sum), bits(min positive term of sq)} as four doubles (the bit patterns are
// integers below 2^32: exact as doubles, so a MIN all-reduce over ranks can carry them)
__global__ void __launch_bounds__(1024)
k_sor_finish(const double* __restrict__ part, int nb, double* __restrict__ out4) {
    __shared__ double rs[16], rq[16];
    __shared__ unsigned int rf[16], rg[16];
    double sum = 0.0, sq = 0.0;
    unsigned int fmin = 0x7f800000u, gmin = 0x7f800000u;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        sum += part[b * 3 + 0];
        sq += part[b * 3 + 1];
        const unsigned long long mm = (unsigned long long)__double_as_longlong(part[b * 3 + 2]);
        fmin = min(fmin, (unsigned int)mm);
        gmin = min(gmin, (unsigned int)(mm >> 32));
    }
    sum = wave_sum_f64(sum);
    sq = wave_sum_f64(sq);
    fmin = wave_min_u32(fmin);
    gmin = wave_min_u32(gmin);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = sum; rq[threadIdx.x >> 6] = sq; rf[threadIdx.x >> 6] = fmin; rg[threadIdx.x >> 6] = gmin; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    sum = 0.0; sq = 0.0;
    for (int w = 0; w < 16; ++w) { sum += rs[w]; sq += rq[w]; fmin = min(fmin, rf[w]); gmin = min(gmin, rg[w]); }
    out4[0] = sum;
    out4[1] = sq;
    out4[2] = (double)fmin;
    out4[3] = (double)gmin;
}
// PCL's mean / variance / threshold from the (possibly all-reduced) sums, and whether the sums are PCL's in-order ones
__host__ __device__ inline void sor_threshold(const double in4[4], double n_valid, int K, double stddev_mult, double* thr, bool* exact) {
    const double sum = in4[0], sq = in4[1];
    const unsigned int fmin = (unsigned int)in4[2], gmin = (unsigned int)in4[3];
    auto last_bit = [](unsigned int bits) {  // last bit of a positive float given by its bits (denormals: 2^-149), as a double
        const int e = (int)(bits >> 23);
        return ldexp(1.0, (e > 0 ? e - 127 : -126) - 23);
    };
    // no addition rounded if the total, counted in last bits of the smallest positive term, stays below 2^52
    *exact = (fmin == 0x7f800000u || sum < last_bit(fmin) * 4503599627370496.0) &&
             (gmin == 0x7f800000u || sq < last_bit(gmin) * 4503599627370496.0);
    // PCL: valid = points with a full neighbourhood (all finite points once the cloud holds K of them)
    const double valid = n_valid >= (double)K ? n_valid : 0.0;
    const double mean = sum / valid;
    const double var = (sq - sum * sum / valid) / (valid - 1.0);
    *thr = mean + stddev_mult * sqrt(var);
}
__global__ void k_sor_threshold(const double* __restrict__ in4, const GridDev* __restrict__ gd, int K, double stddev_mult,
                                SorStats* __restrict__ st) {
    double thr;
    bool exact;
    sor_threshold(in4, (double)gd->n_valid, K, stddev_mult, &thr, &exact);
    st->sum = in4[0];
    st->sq = in4[1];
    st->thr = thr;
    st->kept = 0ull;
    st->exact = exact ? 1u : 0u;
}
__global__ void __launch_bounds__(256)
k_sor_mask(const float* __restrict__ m, size_t n, SorStats* __restrict__ st, uint8_t* __restrict__ inlier) {
    const double thr = st->thr;
    unsigned int cnt = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const bool in = !((double)m[i] > thr);  // PCL: outlier iff distances[i] > threshold (float widened to double)
        if (inlier) inlier[i] = in ? 1 : 0;
        cnt += in ? 1u : 0u;
    }
    // one atomic per WORKGROUP (16k waves adding to one word took 53 us of a 1.3 ms SOR call)
    __shared__ unsigned int wsum[4];
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int all = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (all) atomicAdd(&st->kept, (unsigned long long)all);
    }
}
// sums of the mean distances m[n] -> out4 (device, 4 doubles: see k_sor_finish); scratch: >= 3 * SOR_RED_BLOCKS doubles
int launch_sor_partial(hipStream_t s, const float* m, size_t n, double* scratch, double* out4_dev) {
    const int nb = (int)std::min<size_t>((n + 255) / 256, SOR_RED_BLOCKS);
    hipLaunchKernelGGL(k_sor_reduce, dim3(nb < 1 ? 1 : nb), dim3(256), 0, s, m, n, scratch);
    hipLaunchKernelGGL(k_sor_finish, dim3(1), dim3(1024), 0, s, scratch, nb < 1 ? 1 : nb, out4_dev);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}
// threshold from in4 (the sums of the WHOLE cloud) -> stats_dev; mask + kept count of the n means given
int launch_sor_threshold_mask(hipStream_t s, const float* m, size_t n, const GridDev* gd, int K, double stddev_mult,
                              const double* in4_dev, void* stats_dev, uint8_t* inlier_dev) {
    const int nb = (int)std::min<size_t>((n + 255) / 256, SOR_RED_BLOCKS);
    hipLaunchKernelGGL(k_sor_threshold, dim3(1), dim3(1), 0, s, in4_dev, gd, K, stddev_mult, static_cast<SorStats*>(stats_dev));
    hipLaunchKernelGGL(k_sor_mask, dim3(nb < 1 ? 1 : nb), dim3(256), 0, s, m, n, static_cast<SorStats*>(stats_dev), inlier_dev);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}
// statistics + threshold + mask of the mean distances m[n] of a whole cloud; stats (device, 48 bytes) and scratch (>= 3 *
// SOR_RED_BLOCKS + 4 doubles) are the caller's
int launch_sor_stats(hipStream_t s, const float* m, size_t n, const GridDev* gd, int K, double stddev_mult, double* scratch,
                     void* stats_dev, uint8_t* inlier_dev) {
    double* out4 = scratch + 3 * SOR_RED_BLOCKS;
    PCC_TRY(launch_sor_partial(s, m, n, scratch, out4));
    return launch_sor_threshold_mask(s, m, n, gd, K, stddev_mult, out4, stats_dev, inlier_dev);
}
void sor_threshold_host(const double in4[4], double n_valid, int K, double stddev_mult, double* thr, int* exact) {
    bool e = false;
    sor_threshold(in4, n_valid, K, stddev_mult, thr, &e);
    *exact = e ? 1 : 0;
}

// ---- radiusSearch(..., max_nn): FLANN keeps the max_nn NEAREST neighbours within the radius (KNNRadiusResultSet) --------
__global__ void __launch_bounds__(256)
k_clamp_counts(int32_t* __restrict__ counts, size_t n, int32_t cap) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) counts[i] = min(counts[i], cap);
}
int launch_clamp_counts(hipStream_t s, int32_t* counts, size_t n, int32_t cap) {
    if (n == 0) return PCC_OK;
    hipLaunchKernelGGL(k_clamp_counts, dim3(grid_for(n, 256)), dim3(256), 0, s, counts, n, cap);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}
// k-NN rows (K ascending entries per query; keys, or idx / d2 rows) -> CSR rows: entry e of query i is the e-th nearest
// neighbour if it lies within the radius (d2 < r2, strict), else "nothing found"
__global__ void __launch_bounds__(256)
k_knn_rows_to_csr(const unsigned long long* __restrict__ keys, const int32_t* __restrict__ ridx, const float* __restrict__ rd2, int K,
                  float r2, const int64_t* __restrict__ offsets, size_t nq, int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (size_t)gridDim.x * blockDim.x) {
        const int64_t b = offsets[i], len = offsets[i + 1] - b;
        for (int64_t e = 0; e < len; ++e) {
            int32_t id = -1;
            float d = __builtin_inff();
            if (e < K) {
                if (keys) {
                    const unsigned long long key = keys[i * (size_t)K + e];
                    if (!key_none(key)) { id = (int32_t)(unsigned int)key; d = __uint_as_float((unsigned int)(key >> 32)); }
                } else {
                    id = ridx[i * (size_t)K + e];
                    d = rd2[i * (size_t)K + e];
                }
            }
            const bool in = id >= 0 && d < r2;
            if (idx_out) idx_out[b + e] = in ? id : -1;
            if (d2_out) d2_out[b + e] = in ? d : __builtin_inff();
        }
    }
}
int launch_knn_rows_to_csr(hipStream_t s, const unsigned long long* keys, const int32_t* ridx, const float* rd2, int K, float r2,
                           const int64_t* offsets, size_t nq, int32_t* idx_out, float* d2_out) {
    if (nq == 0) return PCC_OK;
    hipLaunchKernelGGL(k_knn_rows_to_csr, dim3(grid_for(nq, 256)), dim3(256), 0, s, keys, ridx, rd2, K, r2, offsets, nq, idx_out, d2_out);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

}  // namespace pcc
