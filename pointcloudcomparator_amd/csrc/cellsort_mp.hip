// cellsort_mp.hip -- cell sort for clouds that do not fit the L2s (gfx950): three coalesced levels, the 16-byte
// payload travels with its key, nothing is gathered.
//
// cellsort.hip sorts (cell, index) pairs in two levels and lets the last pass GATHER the 16-byte points through the
// sort order.  Up to ~2M points everything it touches stays in the 32 MiB of L2 and that is the fastest form.  At
// 10M points the pair scatter writes 2048 streams per workgroup in 76-byte pieces and the gather reads one 128-byte
// line per 16 bytes used out of a 160 MB array: 172 us + 453 us per sort, ~0.9 TB/s (profiles/r01_c3_kernel_stats.csv).
// Here every level moves whole points between buffers, reading its input front to back and writing long runs:
//   level 1   B1 <= 256 buckets of F1 = B2 * F2 cells.  Per-workgroup LDS histogram, scan of the B1 x G matrix,
//             scatter through LDS cursors: a workgroup's slice lands in B1 runs of ~slice / B1 points.
//   level 2   inside every level-1 bucket, B2 = 128 buckets of F2 cells.  The input is already grouped, so a slice
//             touches a short window of (b1, b2) pairs: LDS counts over the window, ONE returning global atomic per
//             non-empty pair reserves the slice's run in that bucket, LDS cursors place the points.
//   level 3   one workgroup per (b1, b2) bucket, as cellsort.hip's last pass: LDS histogram over its F2 cells, scan
//             (= cell_start), cursor scatter -- but the points are read front to back from level 2's output.
// Same results as cell_sort() (order inside a cell is arbitrary in both); plays the role of
// KDTreeSingleIndex::buildIndex (reference src/comparator.cpp:565) and of the per-call query ordering.
#include "pcc_internal.hpp"
#include "grid_device.hpp"
#include "lane_ops.hpp"

namespace pcc {

constexpr int MP_T = 256;
constexpr unsigned int MP_F2 = 2048;      // cells per level-2 bucket (8 KiB of LDS counters in level 3)
constexpr unsigned int MP_B2 = 128;       // level-2 buckets per level-1 bucket
constexpr unsigned int MP_MAX_B1 = 256;
constexpr unsigned int MP_MAX_G = 512;    // level-1 workgroups
constexpr unsigned int MP_SLICE2 = 8192;  // points per level-2 workgroup (counting form)
constexpr unsigned int MP_STAGE_BYTES = 32768;  // LDS the placing form stages its slice in (2048 points or 4096 pairs)
constexpr unsigned int MP_WIN = 8;        // level-1 buckets a level-2 slice keeps LDS counters for
// elements a level-3 bucket places in LDS before writing them out in whole lines: 16-byte points (reference clouds) / 4-byte order
// words (query clouds).  Round 5, same box: points 768 / 1536 / 3072 / 6144 -> build 450 / 418 / 413 / 473 us at 10M; order words
// see DESIGN.md 4.3
#ifndef PCC_MP_STAGE_PTS
#define PCC_MP_STAGE_PTS 3072
#endif
#ifndef PCC_MP_STAGE_ORD
#define PCC_MP_STAGE_ORD 6144
#endif
constexpr unsigned int MP_FINE_STAGE_PTS = PCC_MP_STAGE_PTS, MP_FINE_STAGE_ORD = PCC_MP_STAGE_ORD;

struct MpPlan {
    unsigned int F1, F2, B1, G1, slice1;
};
static MpPlan mp_plan(unsigned int ncells_cap, unsigned int n) {
    MpPlan p;
    p.F2 = MP_F2;
    while ((unsigned long long)p.F2 * MP_B2 * MP_MAX_B1 < (unsigned long long)ncells_cap + 1) p.F2 *= 2;
    p.F1 = p.F2 * MP_B2;
    p.B1 = (ncells_cap + p.F1) / p.F1;  // ceil((ncells_cap + 1) / F1)
    p.G1 = (n + 16383) / 16384;
    if (p.G1 > MP_MAX_G) p.G1 = MP_MAX_G;
    if (p.G1 < 1) p.G1 = 1;
    p.slice1 = (n + p.G1 - 1) / p.G1;
    return p;
}

// What travels through the levels: the packed point itself (reference clouds: the last level writes the cell-sorted
// copies) or, for query clouds that only need their ORDER, the 8-byte pair (cell, position) made once in level 1.
__device__ __forceinline__ unsigned int mp_cell(const float4& v, const GridParams& g, bool voxel) {
    return voxel ? voxel_id(v, g) : cell_id(v, g);
}
__device__ __forceinline__ unsigned int mp_cell(const uint2& v, const GridParams&, bool) { return v.x; }
__device__ __forceinline__ unsigned int mp_position(const float4& v) { return (unsigned int)__float_as_int(v.w); }
__device__ __forceinline__ unsigned int mp_position(const uint2& v) { return v.y; }
template <class E> __device__ __forceinline__ E mp_make(const float4& v, unsigned int cell);
template <> __device__ __forceinline__ float4 mp_make<float4>(const float4& v, unsigned int) { return v; }
template <> __device__ __forceinline__ uint2 mp_make<uint2>(const float4& v, unsigned int cell) {
    return make_uint2(cell, (unsigned int)__float_as_int(v.w));
}

// ---- level 1 ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(MP_T)
k_mp_hist1(const float4* __restrict__ pts, unsigned int n, const GridDev* __restrict__ gd, unsigned int F1, unsigned int B1,
           unsigned int slice, unsigned int* __restrict__ H, unsigned int* __restrict__ zero, unsigned int n_zero) {
    // (the level-2 counters are zeroed here, two passes ahead of their use: a memset node of its own costs 4.7 us)
    for (unsigned int z = blockIdx.x * MP_T + threadIdx.x; z < n_zero; z += gridDim.x * MP_T) zero[z] = 0u;
    __shared__ unsigned int cnt[MP_MAX_B1];
    const GridParams g = gd->g;
    const bool voxel = gd->voxel != 0;
    for (unsigned int b = threadIdx.x; b < B1; b += MP_T) cnt[b] = 0;
    __syncthreads();
    const unsigned int beg = blockIdx.x * slice, end = min(n, beg + slice);
    for (unsigned int i0 = beg + threadIdx.x; i0 < end; i0 += 4 * MP_T) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * MP_T < end) v[u] = pts[i0 + u * MP_T];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * MP_T < end && __float_as_int(v[u].w) >= 0) atomicAdd(&cnt[mp_cell(v[u], g, voxel) / F1], 1u);
    }
    __syncthreads();
    for (unsigned int b = threadIdx.x; b < B1; b += MP_T) H[(size_t)b * gridDim.x + blockIdx.x] = cnt[b];
    if (blockIdx.x == 0 && threadIdx.x == 0) H[(size_t)B1 * gridDim.x] = 0;  // slot of the grand total
}

template <class E>
__global__ void __launch_bounds__(MP_T)
k_mp_scatter1(const float4* __restrict__ pts, unsigned int n, const GridDev* __restrict__ gd, unsigned int F1, unsigned int B1,
              unsigned int slice, const unsigned int* __restrict__ H, E* __restrict__ out) {
    __shared__ unsigned int cur[MP_MAX_B1];
    const GridParams g = gd->g;
    const bool voxel = gd->voxel != 0;
    for (unsigned int b = threadIdx.x; b < B1; b += MP_T) cur[b] = H[(size_t)b * gridDim.x + blockIdx.x];
    __syncthreads();
    const unsigned int beg = blockIdx.x * slice, end = min(n, beg + slice);
    for (unsigned int i0 = beg + threadIdx.x; i0 < end; i0 += 4 * MP_T) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * MP_T < end) v[u] = pts[i0 + u * MP_T];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * MP_T < end && __float_as_int(v[u].w) >= 0) {
                const unsigned int c = mp_cell(v[u], g, voxel);
                out[atomicAdd(&cur[c / F1], 1u)] = mp_make<E>(v[u], c);
            }
    }
}

// Level 1 of a QUERY cloud whose cells the pack kernel has written (pcc_index::q_cells): histogram and scatter read 4 bytes per
// query -- its cell, ~0 for a non-finite one -- instead of the 16-byte point, whose cell they would work out again
__global__ void __launch_bounds__(MP_T)
k_mp_hist1_cells(const unsigned int* __restrict__ cells, unsigned int n, unsigned int F1, unsigned int B1, unsigned int slice,
                 unsigned int* __restrict__ H, unsigned int* __restrict__ zero, unsigned int n_zero) {
    for (unsigned int z = blockIdx.x * MP_T + threadIdx.x; z < n_zero; z += gridDim.x * MP_T) zero[z] = 0u;
    __shared__ unsigned int cnt[MP_MAX_B1];
    for (unsigned int b = threadIdx.x; b < B1; b += MP_T) cnt[b] = 0;
    __syncthreads();
    const unsigned int beg = blockIdx.x * slice, end = min(n, beg + slice);
    for (unsigned int i0 = beg + threadIdx.x; i0 < end; i0 += 8 * MP_T) {
        unsigned int c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = i0 + u * MP_T < end ? cells[i0 + u * MP_T] : 0xffffffffu;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c[u] != 0xffffffffu) atomicAdd(&cnt[c[u] / F1], 1u);
    }
    __syncthreads();
    for (unsigned int b = threadIdx.x; b < B1; b += MP_T) H[(size_t)b * gridDim.x + blockIdx.x] = cnt[b];
    if (blockIdx.x == 0 && threadIdx.x == 0) H[(size_t)B1 * gridDim.x] = 0;  // slot of the grand total
}
__global__ void __launch_bounds__(MP_T)
k_mp_scatter1_cells(const unsigned int* __restrict__ cells, unsigned int n, unsigned int F1, unsigned int B1, unsigned int slice,
                    const unsigned int* __restrict__ H, uint2* __restrict__ out) {
    __shared__ unsigned int cur[MP_MAX_B1];
    for (unsigned int b = threadIdx.x; b < B1; b += MP_T) cur[b] = H[(size_t)b * gridDim.x + blockIdx.x];
    __syncthreads();
    const unsigned int beg = blockIdx.x * slice, end = min(n, beg + slice);
    for (unsigned int i0 = beg + threadIdx.x; i0 < end; i0 += 8 * MP_T) {
        unsigned int c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = i0 + u * MP_T < end ? cells[i0 + u * MP_T] : 0xffffffffu;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c[u] != 0xffffffffu) out[atomicAdd(&cur[c[u] / F1], 1u)] = make_uint2(c[u], i0 + u * MP_T);
    }
}

// The same scatter with the slice taken in TILES that are sorted by bucket in LDS first: a wave's store then covers a few
// whole 128-byte runs instead of 64 pieces of 16 (or 8) bytes going to 64 different buckets -- the pass moved exactly its
// algorithmic bytes already (PMC) but at 49 % of the HBM roof: it is the L2's write transactions that it was short of.
// Measured at 10M points: index build 0.447 -> 0.434 ms with the 16-byte points; used for them only.
constexpr unsigned int MP_TILE1 = 2048;  // points of a tile: 32 KB of LDS for points, 16 KB for (cell, position) pairs
template <class E>
__global__ void __launch_bounds__(MP_T)
k_mp_scatter1_staged(const float4* __restrict__ pts, unsigned int n, const GridDev* __restrict__ gd, unsigned int F1, unsigned int B1,
                     unsigned int slice, const unsigned int* __restrict__ H, E* __restrict__ out) {
    constexpr int PER = MP_TILE1 / MP_T;  // points a thread takes per tile
    __shared__ unsigned int cur[MP_MAX_B1], tcnt[MP_MAX_B1], tstart[MP_MAX_B1];
    __shared__ unsigned short sb[MP_TILE1];
    __shared__ __attribute__((aligned(16))) unsigned char stage_raw[MP_TILE1 * sizeof(E)];
    E* stage = reinterpret_cast<E*>(stage_raw);
    const GridParams g = gd->g;
    const bool voxel = gd->voxel != 0;
    for (unsigned int b = threadIdx.x; b < B1; b += MP_T) cur[b] = H[(size_t)b * gridDim.x + blockIdx.x];
    const unsigned int beg = blockIdx.x * slice, end = min(n, beg + slice);
    for (unsigned int t0 = beg; t0 < end; t0 += MP_TILE1) {  // (workgroup-uniform)
        for (unsigned int b = threadIdx.x; b < MP_MAX_B1; b += MP_T) tcnt[b] = 0;
        __syncthreads();
        float4 v[PER];
        unsigned int bk[PER], rk[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const unsigned int i = t0 + threadIdx.x + u * MP_T;
            v[u] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (i < end) v[u] = pts[i];
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            bk[u] = 0xffffffffu;
            rk[u] = 0;
            if (__float_as_int(v[u].w) >= 0) {
                bk[u] = mp_cell(v[u], g, voxel);
                rk[u] = atomicAdd(&tcnt[bk[u] / F1], 1u);  // rank inside the tile's run for that bucket
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // exclusive scan of the <= 256 tile counts: four per lane of the first wave
            unsigned int c[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { c[j] = tcnt[4 * threadIdx.x + j]; sum += c[j]; }
            unsigned int ex = wave_incl_scan_add(sum) - sum;
#pragma unroll
            for (int j = 0; j < 4; ++j) { tstart[4 * threadIdx.x + j] = ex; ex += c[j]; }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (bk[u] != 0xffffffffu) {
                const unsigned int b = bk[u] / F1, slot = tstart[b] + rk[u];
                stage[slot] = mp_make<E>(v[u], bk[u]);
                sb[slot] = (unsigned short)b;
            }
        __syncthreads();
        const unsigned int tile_n = tstart[MP_MAX_B1 - 1] + tcnt[MP_MAX_B1 - 1];  // valid points of the tile
        for (unsigned int j = threadIdx.x; j < tile_n; j += MP_T) {
            const unsigned int b = sb[j];
            out[cur[b] + (j - tstart[b])] = stage[j];
        }
        __syncthreads();
        for (unsigned int b = threadIdx.x; b < MP_MAX_B1; b += MP_T) cur[b] += tcnt[b];
    }
}

// ---- level 2 ---------------------------------------------------------------------------------------------------
// pair id of a cell: (level-1 bucket) * B2 + (level-2 bucket inside it)
__device__ __forceinline__ unsigned int mp_pair(unsigned int c, unsigned int F1, unsigned int F2) {
    return (c / F1) * MP_B2 + (c % F1) / F2;
}

// PLACE = false: C[pair] += points of the slice in that pair.  PLACE = true: C holds the cursors (scanned counts);
// the slice is read ONCE into LDS, reserves its runs and writes its points from there.
template <bool PLACE, class E>
__global__ void __launch_bounds__(MP_T)
k_mp_level2(const E* __restrict__ in, const unsigned int* __restrict__ n_valid_ptr, const GridDev* __restrict__ gd,
            unsigned int F1, unsigned int F2, unsigned int* __restrict__ C, E* __restrict__ out) {
    constexpr unsigned int SLICE = PLACE ? MP_STAGE_BYTES / sizeof(E) : MP_SLICE2;
    __shared__ unsigned int cnt[MP_WIN * MP_B2];
    __shared__ __attribute__((aligned(16))) unsigned char stage_raw[PLACE ? MP_STAGE_BYTES : 16];
    E* stage = reinterpret_cast<E*>(stage_raw);
    const GridParams g = gd->g;
    const bool voxel = gd->voxel != 0;
    const unsigned int n = *n_valid_ptr;
    const unsigned int beg = blockIdx.x * SLICE, end = min(n, beg + SLICE);
    if (beg >= end) return;
    // the input is grouped by level-1 bucket: the slice's first point names the start of the LDS window
    const unsigned int w0 = (mp_cell(in[beg], g, voxel) / F1) * MP_B2;
    for (unsigned int k = threadIdx.x; k < MP_WIN * MP_B2; k += MP_T) cnt[k] = 0;
    __syncthreads();
    for (unsigned int i0 = beg + threadIdx.x; i0 < end; i0 += 4 * MP_T) {
        E v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * MP_T < end) v[u] = in[i0 + u * MP_T];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * MP_T >= end) break;
            if (PLACE) stage[i0 + u * MP_T - beg] = v[u];
            const unsigned int pr = mp_pair(mp_cell(v[u], g, voxel), F1, F2);
            if (pr - w0 < MP_WIN * MP_B2) atomicAdd(&cnt[pr - w0], 1u);
            else if (!PLACE) atomicAdd(&C[pr], 1u);              // beyond the window (tiny level-1 buckets): straight to memory
            else out[atomicAdd(&C[pr], 1u)] = v[u];
        }
    }
    __syncthreads();
    for (unsigned int k = threadIdx.x; k < MP_WIN * MP_B2; k += MP_T) {
        const unsigned int c = cnt[k];
        if (c) {
            if (!PLACE) atomicAdd(&C[w0 + k], c);
            else cnt[k] = atomicAdd(&C[w0 + k], c);  // the slice's run in that bucket: count -> cursor
        }
    }
    if (!PLACE) return;
    __syncthreads();
    for (unsigned int j = threadIdx.x; j < end - beg; j += MP_T) {
        const E v = stage[j];
        const unsigned int pr = mp_pair(mp_cell(v, g, voxel), F1, F2);
        if (pr - w0 < MP_WIN * MP_B2) out[atomicAdd(&cnt[pr - w0], 1u)] = v;
    }
}

__device__ __forceinline__ void mp_store_point(float4* out, unsigned int pos, const float4& v) { if (out) out[pos] = v; }
__device__ __forceinline__ void mp_store_point(float4*, unsigned int, const uint2&) {}

// ---- level 3 ---------------------------------------------------------------------------------------------------
template <bool REFS, class E>
__global__ void __launch_bounds__(MP_T)
k_mp_fine(const E* __restrict__ in, const unsigned int* __restrict__ ends /* cursors after level 2: bucket ends */, const GridDev* __restrict__ gd,
          unsigned int F2, float4* __restrict__ out_pts, unsigned int* __restrict__ out_order,
          unsigned int* __restrict__ cell_start) {
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];  // F2 counters + 4 scan words
    unsigned int* cnt = lds;
    unsigned int* wsum = lds + F2;
    const GridParams g = gd->g;
    const bool voxel = gd->voxel != 0;
    const unsigned int ncells = (unsigned int)g.ncells;
    const unsigned int b = blockIdx.x;
    const unsigned int cell0 = b * F2;
    if (cell0 > ncells) return;  // (entry [ncells] of cell_start belongs to the bucket that holds it)
    const unsigned int beg = b ? ends[b - 1] : 0u, end = ends[b];
    if (!REFS && beg == end) return;
    for (unsigned int f = threadIdx.x; f < F2; f += MP_T) cnt[f] = 0;
    __syncthreads();
    for (unsigned int j0 = beg + threadIdx.x; j0 < end; j0 += 4 * MP_T) {
        E v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + u * MP_T < end) v[u] = in[j0 + u * MP_T];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + u * MP_T < end) atomicAdd(&cnt[mp_cell(v[u], g, voxel) - cell0], 1u);
    }
    __syncthreads();
    // exclusive scan of the F2 counters: thread t owns a contiguous chunk
    const unsigned int per = (F2 + MP_T - 1) / MP_T;
    const unsigned int f0 = threadIdx.x * per, f1 = min(F2, f0 + per);
    unsigned int s = 0;
    for (unsigned int f = f0; f < f1; ++f) s += cnt[f];
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int inc = wave_incl_scan_add(s);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned int run = beg + inc - s;
    for (unsigned int w = 0; w < wave; ++w) run += wsum[w];
    for (unsigned int f = f0; f < f1; ++f) {
        const unsigned int c = cnt[f];
        cnt[f] = run;  // becomes the cursor of cell f
        if (REFS && cell0 + f <= ncells) cell_start[cell0 + f] = run;
        run += c;
    }
    __syncthreads();
    // A bucket of up to stage_cap elements (the usual case) is placed in LDS and leaves in whole lines; bigger ones
    // scatter their 16-byte (or 4-byte) pieces straight to memory.
    const unsigned int stage_cap = out_pts ? MP_FINE_STAGE_PTS : MP_FINE_STAGE_ORD;
    const bool staged = end - beg <= stage_cap;
    float4* st_pts = reinterpret_cast<float4*>(lds + ((F2 + 4 + 3) & ~3u));
    unsigned int* st_ord = reinterpret_cast<unsigned int*>(st_pts + (out_pts ? MP_FINE_STAGE_PTS : 0));
    for (unsigned int j0 = beg + threadIdx.x; j0 < end; j0 += 4 * MP_T) {
        E v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + u * MP_T < end) v[u] = in[j0 + u * MP_T];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u * MP_T >= end) break;
            const unsigned int pos = atomicAdd(&cnt[mp_cell(v[u], g, voxel) - cell0], 1u);
            if (staged) {
                mp_store_point(out_pts ? st_pts : nullptr, pos - beg, v[u]);
                if (out_order) st_ord[pos - beg] = mp_position(v[u]);
            } else {
                mp_store_point(out_pts, pos, v[u]);  // .w still carries the packed position (the original index)
                if (out_order) out_order[pos] = mp_position(v[u]);
            }
        }
    }
    if (staged) {
        __syncthreads();
        for (unsigned int i = threadIdx.x; i < end - beg; i += MP_T) {
            if (out_pts) out_pts[beg + i] = st_pts[i];
            if (out_order) out_order[beg + i] = st_ord[i];
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
// Same contract as cell_sort(); additionally out_pts may be given for queries (cell-sorted copies, .w = position in
// the query array).  Uses ix->mp_a / mp_b as the two intermediate point buffers.
int cell_sort_mp(pcc_index* ix, const float4* pts, size_t n_pts, bool refs, float4* out_pts, unsigned int* out_order,
                 unsigned int* cell_start, unsigned int** n_sorted_dev, const GridDev* gd_override,
                 unsigned int nc_cap_override) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)n_pts;
    const unsigned int ncap = gd_override ? nc_cap_override : ix->nc_cap;
    const MpPlan p = mp_plan(ncap, n);
    const GridDev* gd = gd_override ? gd_override : ix->d_grid.as<GridDev>();
    const unsigned int np = p.B1 * MP_B2;  // (b1, b2) pairs
    PCC_TRY(ix->mp_a.reserve((size_t)n * sizeof(float4) + 256));
    PCC_TRY(ix->mp_b.reserve((size_t)n * sizeof(float4) + 256));
    const size_t h_elems = (size_t)p.B1 * p.G1 + 1;
    PCC_TRY(ix->scratch_b.reserve(((h_elems + 3) & ~(size_t)3) * sizeof(unsigned int)));
    PCC_TRY(ix->mp_c.reserve(((size_t)np + 8) * sizeof(unsigned int)));
    unsigned int* H = ix->scratch_b.as<unsigned int>();
    // per (b1, b2) bucket: counts -> (scan) first position -> (placement advances it) one past its last point,
    // which is also where the next bucket starts: level 3 reads its range from there
    unsigned int* C = ix->mp_c.as<unsigned int>();
    // (the queries' cells as the pack kernel left them: this cloud, this grid, nothing has moved since -- used once)
    const unsigned int* cells = nullptr;
    if (!refs && !out_pts && !gd_override && pts == ix->q_packed.as<float4>() && ix->q_cells_n == n_pts && n_pts != 0)
        cells = ix->q_cells.as<unsigned int>();
    if (!refs) ix->q_cells_n = 0;
    if (cells)
        hipLaunchKernelGGL(k_mp_hist1_cells, dim3(p.G1), dim3(MP_T), 0, s, cells, n, p.F1, p.B1, p.slice1, H, C, np);
    else
    hipLaunchKernelGGL(k_mp_hist1, dim3(p.G1), dim3(MP_T), 0, s, pts, n, gd, p.F1, p.B1, p.slice1, H, C, np);
    PCC_HIP(hipGetLastError());
    PCC_TRY(launch_exclusive_scan(ix, s, H, h_elems, ix->scratch_a));
    unsigned int* n_valid = H + h_elems - 1;  // grand total == number of valid points
    if (n_sorted_dev) *n_sorted_dev = n_valid;
    // counters + scan words, then the staged output (points and / or order words)
    const size_t lds3 = (((size_t)p.F2 + 4 + 3) & ~(size_t)3) * sizeof(unsigned int) +
                        (out_pts ? (size_t)MP_FINE_STAGE_PTS * (sizeof(float4) + (out_order ? sizeof(unsigned int) : 0))
                                 : (size_t)MP_FINE_STAGE_ORD * sizeof(unsigned int));
    const unsigned int g2c = (n + MP_SLICE2 - 1) / MP_SLICE2;
    if (out_pts) {  // the points travel (reference clouds)
        float4* t1 = ix->mp_a.as<float4>();
        float4* t2 = ix->mp_b.as<float4>();
        const unsigned int g2p = (n + MP_STAGE_BYTES / sizeof(float4) - 1) / (MP_STAGE_BYTES / sizeof(float4));
        if (ix->opt.sort_stage1)
            hipLaunchKernelGGL((k_mp_scatter1_staged<float4>), dim3(p.G1), dim3(MP_T), 0, s, pts, n, gd, p.F1, p.B1, p.slice1, H, t1);
        else
            hipLaunchKernelGGL((k_mp_scatter1<float4>), dim3(p.G1), dim3(MP_T), 0, s, pts, n, gd, p.F1, p.B1, p.slice1, H, t1);
        hipLaunchKernelGGL((k_mp_level2<false, float4>), dim3(g2c), dim3(MP_T), 0, s, t1, n_valid, gd, p.F1, p.F2, C, t2);
        PCC_TRY(launch_exclusive_scan(ix, s, C, np, ix->scratch_a));
        hipLaunchKernelGGL((k_mp_level2<true, float4>), dim3(g2p), dim3(MP_T), 0, s, t1, n_valid, gd, p.F1, p.F2, C, t2);
        if (refs)
            hipLaunchKernelGGL((k_mp_fine<true, float4>), dim3(np), dim3(MP_T), lds3, s, t2, C, gd, p.F2, out_pts, out_order, cell_start);
        else
            hipLaunchKernelGGL((k_mp_fine<false, float4>), dim3(np), dim3(MP_T), lds3, s, t2, C, gd, p.F2, out_pts, out_order, cell_start);
    } else {        // only the order is wanted (query clouds): (cell, position) pairs travel, half the bytes
        uint2* t1 = ix->mp_a.as<uint2>();
        uint2* t2 = ix->mp_b.as<uint2>();
        const unsigned int g2p = (n + MP_STAGE_BYTES / sizeof(uint2) - 1) / (MP_STAGE_BYTES / sizeof(uint2));
        // (the 8-byte pairs lose with the staged form: 10M queries 0.218 vs 0.209 ms for the whole sort -- their pieces are half
        // as long and the tile's LDS round trip costs what the stores save; option value 2 forces it for measurements)
        if (cells)
            hipLaunchKernelGGL(k_mp_scatter1_cells, dim3(p.G1), dim3(MP_T), 0, s, cells, n, p.F1, p.B1, p.slice1, H, t1);
        else if (ix->opt.sort_stage1 == 2)
            hipLaunchKernelGGL((k_mp_scatter1_staged<uint2>), dim3(p.G1), dim3(MP_T), 0, s, pts, n, gd, p.F1, p.B1, p.slice1, H, t1);
        else
            hipLaunchKernelGGL((k_mp_scatter1<uint2>), dim3(p.G1), dim3(MP_T), 0, s, pts, n, gd, p.F1, p.B1, p.slice1, H, t1);
        hipLaunchKernelGGL((k_mp_level2<false, uint2>), dim3(g2c), dim3(MP_T), 0, s, t1, n_valid, gd, p.F1, p.F2, C, t2);
        PCC_TRY(launch_exclusive_scan(ix, s, C, np, ix->scratch_a));
        hipLaunchKernelGGL((k_mp_level2<true, uint2>), dim3(g2p), dim3(MP_T), 0, s, t1, n_valid, gd, p.F1, p.F2, C, t2);
        if (refs)
            hipLaunchKernelGGL((k_mp_fine<true, uint2>), dim3(np), dim3(MP_T), lds3, s, t2, C, gd, p.F2, out_pts, out_order, cell_start);
        else
            hipLaunchKernelGGL((k_mp_fine<false, uint2>), dim3(np), dim3(MP_T), lds3, s, t2, C, gd, p.F2, out_pts, out_order, cell_start);
    }
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

}  // namespace pcc
