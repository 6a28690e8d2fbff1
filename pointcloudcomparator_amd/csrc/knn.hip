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
#include "lane_ops.hpp"
#include <type_traits>
#include <cmath>
#include <cstring>

namespace pcc {
// base + the number of lanes below this one whose bit is set in `mask` (v_mbcnt: two instructions, the base folded in)
__device__ __forceinline__ unsigned int lanes_below(unsigned long long mask, unsigned int base) {
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, base));
}

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
        const unsigned long long k1 = make_key(dist2_nc(qx, qy, qz, r1), r1);
        PCC_PAIR(two);
        if (k0 < worst) knn_insert(list, K, k0, worst);
        if (two && k1 < worst) knn_insert(list, K, k1, worst);
    }
}

__global__ void __launch_bounds__(256)
k_grid_knn(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
           const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned int* __restrict__ order,
           const unsigned int* __restrict__ n_sorted_ptr, unsigned int /*n*/, int K,
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
    float ux, uy, uz;  // the query in the grid's frame (grid_device.hpp): cells and bounds; the distances take (qx, qy, qz)
    grid_frame(g, qx, qy, qz, ux, uy, uz);
    const int cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
    const int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
    const int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
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
        const float lb2 = outside_bound2(ux, uy, uz, x0, x1, y0, y1, z0, z1, g, slack);
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

constexpr unsigned int RAD_FLAT_CAP = 4 * 2048;  // flat candidates the span-end bits cover (4 planes of 64 words)
constexpr int KNN_BOX_MAX = 64;  // half-width (cells) up to which a box is walked (row by row beyond the LDS table); past it: whole grid

// ---- wave-cooperative k-NN (K <= 512) ---------------------------------------------------------
// One WAVE per query.  The 64 lanes read a row's candidates with ONE coalesced load, turn them
// into (d2, position) keys, drop everything not below the current K-th key (tau), and compact
// the survivors into a small LDS staging buffer (ballot + prefix count).  Whenever 64 survivors
// have gathered they are bitonic-sorted across the lanes and merged into the running top list
// (64*KR keys, one/two registers per lane, ascending across lanes) with the classic
// min(L[i], B[63-i]) half-cleaner followed by a 6-stage bitonic merge; tau tightens and after the
// first few batches almost every candidate dies at the compare.  The per-lane list kernel below
// did O(K) global-memory traffic per insertion: 90 ms for 1M points at k = 51.
// value of lane `src`; src must be wave-uniform (v_readlane_b32)
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, src);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long cmpx(unsigned long long v, int m, bool take_min, unsigned int lane) {
    const unsigned long long o = xor_lane_u64(v, m, lane);
    const bool o_less = o < v;
    return (o_less == take_min) ? o : v;
}
__device__ __forceinline__ unsigned long long bitonic_sort64(unsigned long long v, unsigned int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1)
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const bool up = (lane & k2) == 0, lower = (lane & j) == 0;
            v = cmpx(v, j, lower == up, lane);
        }
    return v;  // ascending over the lanes
}
__device__ __forceinline__ unsigned long long bitonic_merge64(unsigned long long v, unsigned int lane) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) v = cmpx(v, j, (lane & j) == 0, lane);
    return v;  // bitonic in -> ascending out
}

// merge a batch of 64 keys into the sorted top list (64 * KR keys, ascending over (register, lane)): the sorted
// batch enters register 0; what each register pushes out (the upper half of a 128-key bitonic split) cascades
// into the next one; the overflow of the last register is dropped
template <int KR>
__device__ __forceinline__ void topk_merge(unsigned long long (&top)[KR], unsigned long long batch, unsigned int lane) {
    unsigned long long carry = bitonic_sort64(batch, lane);
#pragma unroll
    for (int r = 0; r < KR; ++r) {
        const unsigned long long rev = reverse_lanes_u64(carry, lane);
        const unsigned long long lo = rev < top[r] ? rev : top[r];
        const unsigned long long hi = rev < top[r] ? top[r] : rev;
        top[r] = bitonic_merge64(lo, lane);
        if (r + 1 < KR) carry = bitonic_merge64(hi, lane);
    }
}

// where a k-NN row goes: the key array (internal consumers: SOR, normals, region growing) or straight into the caller's
// index / distance arrays (pcc_knn: no key array, no unpack pass over K entries per query)
struct KnnOut {
    unsigned long long* keys;
    int32_t* idx;
    float* d2;
};
__device__ __forceinline__ void knn_emit(const KnnOut& o, size_t p, unsigned long long key) {
    if (o.keys) {
        o.keys[p] = key;
    } else {
        const bool none = key_none(key);
        if (o.idx) o.idx[p] = none ? -1 : (int32_t)(unsigned int)key;
        if (o.d2) o.d2[p] = none ? __builtin_inff() : __uint_as_float((unsigned int)(key >> 32));
    }
}
// rows of non-finite queries (flagged w < 0 by the pack; the searches never see them): "nothing found" throughout
__global__ void __launch_bounds__(256)
k_knn_fill_invalid(const float4* __restrict__ q, unsigned int nq, int K, KnnOut out, const GridDev* __restrict__ gd) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    // (an index without a finite point -- the API refuses it earlier -- would make the wave kernels leave at once: every row then)
    if (i >= nq || (__float_as_int(q[i].w) >= 0 && gd->n_valid != 0u)) return;
    for (int e = 0; e < K; ++e) knn_emit(out, (size_t)i * K + e, ~0ull);
}

template <int KR>
__global__ void __launch_bounds__(256)
k_grid_knn_wave(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
                const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned int* __restrict__ order,
                const unsigned int* __restrict__ n_sorted_ptr, int K, KnnOut out) {
    static_assert(KR == 1 || KR == 2 || KR == 4 || KR == 8, "top list: 64 * KR keys in KR registers per lane");
    __shared__ unsigned long long stage_all[4][128];
    // two spans per row; boxes of up to 11 x 11 rows take the flat walk -- up to the largest cube (19 x 19) for
    // the big-K instantiations, whose sparse neighbourhoods need it
    constexpr int ROWCAP2 = KR <= 2 ? 2 * 11 * 11 : 2 * (2 * GRID_KMAX + 3) * (2 * GRID_KMAX + 3);
    __shared__ unsigned int tab_s_all[4][ROWCAP2], tab_o_all[4][ROWCAP2], win_all[4][64];
    unsigned int* tab_s = tab_s_all[threadIdx.x >> 6];
    unsigned int* tab_o = tab_o_all[threadIdx.x >> 6];
    unsigned int* win = win_all[threadIdx.x >> 6];
    const GridParams g = gd->g;
    const float slack = gd->slack;
    const unsigned int n_valid = gd->n_valid;
    const unsigned int ns = *n_sorted_ptr;
    if (n_valid == 0) return;
    const unsigned int lane = threadIdx.x & 63;
    unsigned long long* stage = stage_all[threadIdx.x >> 6];
    const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));  // (uniform: scalar header loads)
    const unsigned int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int want = (unsigned int)K < n_valid ? K : (int)n_valid;
    for (unsigned int t = wave; t < ns; t += nwaves) {  // wave-uniform
        const unsigned int qi = order[t];
        const float4 qv = q[qi];
        const float qx = qv.x, qy = qv.y, qz = qv.z;
        float ux, uy, uz;  // the query in the grid's frame (grid_device.hpp): cells and bounds; the distances take (qx, qy, qz)
        grid_frame(g, qx, qy, qz, ux, uy, uz);
        const int cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
        const int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
        const int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
        // first pass: the smallest cube that holds at least 2 x `want` points (its K-th key is then an upper
        // bound); the second pass covers what the ball of that bound adds around the cube.  Every pass scans
        // its box MINUS the box already scanned, so no point is seen twice and nothing is rescanned.
        int k = 1;
        for (;; ++k) {
            const int x0 = max(cx - k, 0), x1 = min(cx + k, g.dim[0] - 1);
            const int y0 = max(cy - k, 0), y1 = min(cy + k, g.dim[1] - 1);
            const int z0 = max(cz - k, 0), z1 = min(cz + k, g.dim[2] - 1);
            const int ny = y1 - y0 + 1, nrow = ny * (z1 - z0 + 1);
            unsigned int cnt = 0;
            for (int r = (int)lane; r < nrow; r += 64) {
                const unsigned int row = ((unsigned int)(z0 + r / ny) * g.dim[1] + (y0 + r % ny)) * g.dim[0];
                cnt += cell_start[row + x1 + 1] - cell_start[row + x0];
            }
            cnt = (unsigned int)__builtin_amdgcn_readlane((int)wave_incl_scan_add(cnt), 63);
            // (twice `want`: with barely `want` points in the cube the K-th of them sits in a corner and the
            // ball of that bound is several times the cube -- for large K in sparse regions wider than KMAX cells)
            if (cnt >= 2u * (unsigned int)want || k >= GRID_KMAX) break;
        }
        unsigned long long top[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) top[r] = ~0ull;
        unsigned long long tau = ~0ull;
        unsigned int scnt = 0;  // wave-uniform
        // one batch of <= 64 candidate keys: filter by tau, stage the survivors, merge when 64 have gathered
        auto consume = [&](unsigned long long key) {
            const bool pass = key < tau;
            const unsigned long long mask = __ballot(pass);
            if (mask == 0) return;
            if (pass) stage[lanes_below(mask, scnt)] = key;
            scnt += (unsigned int)__popcll(mask);
            wave_lds_sync();
            if (scnt >= 64) {
                const unsigned long long batch = stage[lane];
                const unsigned int rest = scnt - 64;
                const unsigned long long carry = lane < rest ? stage[64 + lane] : ~0ull;
                wave_lds_sync();
                if (lane < rest) stage[lane] = carry;
                scnt = rest;
                topk_merge<KR>(top, batch, lane);
                tau = shfl_u64(top[(want - 1) >> 6], (want - 1) & 63);
                wave_lds_sync();
            }
        };
        auto flush = [&]() {
            if (scnt) {
                const unsigned long long batch = lane < scnt ? stage[lane] : ~0ull;
                wave_lds_sync();
                topk_merge<KR>(top, batch, lane);
                tau = shfl_u64(top[(want - 1) >> 6], (want - 1) & 63);
                scnt = 0;
                wave_lds_sync();
            }
        };
        // box of this pass and box already scanned (empty at first)
        int x0 = max(cx - k, 0), x1 = min(cx + k, g.dim[0] - 1);
        int y0 = max(cy - k, 0), y1 = min(cy + k, g.dim[1] - 1);
        int z0 = max(cz - k, 0), z1 = min(cz + k, g.dim[2] - 1);
        int ix0 = 1, ix1 = 0, iy0 = 1, iy1 = 0, iz0 = 1, iz1 = 0;
        bool ball_pass = false;  // this pass covers the ball of a valid bound: exact when it ends
        for (;;) {
            const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.dim[0] - 1 && y1 == g.dim[1] - 1 && z1 == g.dim[2] - 1;
            const int ny = y1 - y0 + 1, nrow = ny * (z1 - z0 + 1);
            if (2 * nrow <= ROWCAP2) {
                // FLAT walk: the pass's non-empty spans go into an LDS table (start, running offset) -- lanes
                // over rows, so the bounds cost a few wave loads instead of two dependent loads per row -- and
                // the candidates are then taken 64 at a time across span boundaries.  A row that crosses the
                // box already scanned contributes the part left of it and the part right of it.
                unsigned int nspans = 0, total = 0;  // wave-uniform
                for (int base = 0; base < 2 * nrow; base += 64) {
                    const int v = base + (int)lane, r = v >> 1, side = v & 1;
                    unsigned int s0 = 0, cnt = 0;
                    if (r < nrow) {
                        const int z = z0 + r / ny, y = y0 + r % ny;
                        const bool crosses = y >= iy0 && y <= iy1 && z >= iz0 && z <= iz1;
                        int a = x0, b = x1;  // cells [a, b]
                        if (crosses) { if (side == 0) b = ix0 - 1; else a = ix1 + 1; }
                        else if (side == 1) b = a - 1;
                        if (a <= b) {
                            const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
                            s0 = cell_start[row + a];
                            cnt = cell_start[row + b + 1] - s0;
                        }
                    }
                    const unsigned int incl = wave_incl_scan_add(cnt);
                    const unsigned long long occ = __ballot(cnt != 0);
                    if (cnt) {
                        const unsigned int slot = lanes_below(occ, nspans);
                        tab_s[slot] = s0;
                        tab_o[slot] = total + incl - cnt;
                    }
                    nspans += (unsigned int)__popcll(occ);
                    total += (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
                }
                wave_lds_sync();
                unsigned int next_span = 0, carry_span = 0;  // wave-uniform
                for (unsigned int B = 0; B < total; B += 64) {
                    win[lane] = 0u;
                    wave_lds_sync();
                    const unsigned int r = next_span + lane;
                    const bool starts = r < nspans && tab_o[r] < B + 64;  // offsets are strictly increasing
                    if (starts) win[tab_o[r] - B] = r + 1;
                    next_span += (unsigned int)__popcll(__ballot(starts));
                    wave_lds_sync();
                    unsigned int v = wave_incl_scan_max(win[lane]);
                    v = max(v, carry_span);
                    carry_span = (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
                    const unsigned int c = B + lane;
                    unsigned long long key = ~0ull;
                    if (c < total) {
                        const unsigned int my = v - 1;
                        const float4 r4 = cell_refs[tab_s[my] + (c - tab_o[my])];
                        key = make_key(dist2(qx, qy, qz, r4), r4);
                    }
                    consume(key);
                }
            } else {
                // a box with more rows than the table holds (very sparse neighbourhoods, the whole-grid pass):
                // row by row, from scratch
#pragma unroll
                for (int r = 0; r < KR; ++r) top[r] = ~0ull;
                tau = ~0ull;
                scnt = 0;
                for (int z = z0; z <= z1; ++z)
                    for (int y = y0; y <= y1; ++y) {
                        const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
                        const unsigned int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
                        for (unsigned int p = s; p < e; p += 64) {
                            const unsigned int pp = p + lane;
                            unsigned long long key = ~0ull;
                            if (pp < e) {
                                const float4 r4 = cell_refs[pp];
                                key = make_key(dist2(qx, qy, qz, r4), r4);
                            }
                            consume(key);
                        }
                    }
            }
            flush();
            if (whole || ball_pass) break;
            ix0 = x0; ix1 = x1; iy0 = y0; iy1 = y1; iz0 = z0; iz1 = z1;
            bool go_whole = false;
            if (tau != ~0ull) {
                const float td = __uint_as_float((unsigned int)(tau >> 32));
                const float lb2 = outside_bound2(ux, uy, uz, x0, x1, y0, y1, z0, z1, g, slack);
                if (td < lb2) break;  // nothing outside the scanned box can beat the K-th key
                // cover the ball of the bound (plus what is scanned already, so the subtraction stays a box)
                const float rb = sqrtf(td) * 1.00001f + slack;
                int a0, a1, b0, b1, c0, c1;
                cell_range(ux, rb, g.org[0], g.inv_h, g.dim[0], a0, a1);
                cell_range(uy, rb, g.org[1], g.inv_h, g.dim[1], b0, b1);
                cell_range(uz, rb, g.org[2], g.inv_h, g.dim[2], c0, c1);
                x0 = min(x0, a0); x1 = max(x1, a1); y0 = min(y0, b0); y1 = max(y1, b1); z0 = min(z0, c0); z1 = max(z1, c1);
                ball_pass = rb < __builtin_inff();
                go_whole = !ball_pass;
            } else {
                // fewer than `want` points so far: a bigger cube
                k = k >= 2 ? 2 * k : k + 1;
                if (k > KNN_BOX_MAX) go_whole = true;
                x0 = max(cx - k, 0); x1 = min(cx + k, g.dim[0] - 1);
                y0 = max(cy - k, 0); y1 = min(cy + k, g.dim[1] - 1);
                z0 = max(cz - k, 0); z1 = min(cz + k, g.dim[2] - 1);
            }
            if (!go_whole && (x1 - x0 > 2 * KNN_BOX_MAX || y1 - y0 > 2 * KNN_BOX_MAX || z1 - z0 > 2 * KNN_BOX_MAX)) go_whole = true;
            if (!go_whole && 2 * (y1 - y0 + 1) * (z1 - z0 + 1) > ROWCAP2) {
                ix0 = iy0 = iz0 = 1; ix1 = iy1 = iz1 = 0;  // too many rows for the table: this box row by row, from scratch
            }
            if (go_whole) {  // from scratch over the whole grid (exact, slow, rare)
                x0 = y0 = z0 = 0; x1 = g.dim[0] - 1; y1 = g.dim[1] - 1; z1 = g.dim[2] - 1;
                ix0 = iy0 = iz0 = 1; ix1 = iy1 = iz1 = 0;
                ball_pass = false;
#pragma unroll
                for (int r = 0; r < KR; ++r) top[r] = ~0ull;
                tau = ~0ull;
            }
        }
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = r * 64 + (int)lane;
            if (e < K) knn_emit(out, (size_t)qi * K + e, top[r]);
        }
    }
}

// ---- sorting the rows of a filled radius search ---------------------------------------------------
// pcl::KdTreeFLANN::radiusSearch returns its neighbours ascending by distance.  One WAVE per row: the row's
// keys live in R registers per lane (element e = r * 64 + lane), a bitonic network sorts them -- exchanges at
// distance < 64 cross lanes (DPP / permlane, lane_ops.hpp), larger distances pair registers of the same lane --
// and the row is written back.  The one-lane insertion sort this replaces was O(len^2) global-memory moves:
// 127 ms of a 141 ms search (5M queries, 83 neighbours each).
template <int R>
__device__ __forceinline__ void bitonic_sort_regs(unsigned long long (&v)[R], unsigned int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int jr = j >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int rp = r ^ jr;
                    if (rp > r) {
                        const bool up = ((r * 64) & k) == 0;  // k > j >= 64: decided by the register index
                        const unsigned long long a = v[r], b = v[rp];
                        const unsigned long long lo = a < b ? a : b, hi = a < b ? b : a;
                        v[r] = up ? lo : hi;
                        v[rp] = up ? hi : lo;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool up = (((unsigned int)(r * 64) + lane) & (unsigned int)k) == 0;
                    v[r] = cmpx(v[r], j, ((lane & (unsigned int)j) == 0) == up, lane);
                }
            }
        }
    }
}

// ---- rows in LDS: bucket + rank sort ------------------------------------------------------------------------
// The hits of a radius search all lie below r^2, roughly evenly over [0, r^2) (surface: evenly; volume: ~sqrt): 128
// buckets of equal width in d2 take one or two keys each.  Count (LDS atomics), scan the 128 counts over the lanes,
// scatter the keys to their bucket's range, and give every key its rank among the few of its bucket: position =
// bucket start + number of smaller keys there.  The bucket is a monotone function of d2 alone, so bucket-then-key
// order IS key order.  ~110 VALU instructions for a row of 83 against ~390 of the bitonic network over 128 slots,
// which moves every key through 28 compare-exchange steps wherever it started.  A bucket holding more than
// BUCKET_FULL keys (lattices: few distinct distances) sends the row to the network instead.
constexpr unsigned int BUCKET_ROW_MAX = 256, BUCKET_N = 128, BUCKET_FULL = 24;
constexpr unsigned int ROW_LDS_MAX = 256;  // rows the fused fill keeps on the chip (longer ones: keys in memory, k_sort_rows)
static_assert(ROW_LDS_MAX <= BUCKET_ROW_MAX, "every row kept in LDS can take the bucket sort");
__device__ __forceinline__ unsigned int bucket_of(unsigned long long key, float scale) {
    return (unsigned int)fminf(__uint_as_float((unsigned int)(key >> 32)) * scale, (float)(BUCKET_N - 1));
}
// (nothing is carried in registers from phase to phase: a key is read again and its bucket recomputed -- one DS read and
// three VALU instructions per key and phase against 12 more live registers, which cost the whole kernel a wave per SIMD)
template <int R>
__device__ __forceinline__ bool bucket_sort_lds(unsigned long long* stage, unsigned long long* tmp, unsigned int* bk,
                                                unsigned int have, float scale, unsigned int lane) {
    // bk[0] = 0, bk[1 + b] = count, then fill pointer, then END of bucket b
    bk[1 + 2 * lane] = 0u;
    bk[2 + 2 * lane] = 0u;
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned int e = (unsigned int)(r * 64) + lane;
        if (e < have) atomicAdd(&bk[1 + bucket_of(stage[e], scale)], 1u);
    }
    wave_lds_sync();
    const unsigned int c0 = bk[1 + 2 * lane], c1 = bk[2 + 2 * lane];
    if (__ballot(max(c0, c1) > BUCKET_FULL) != 0ull) return false;
    const unsigned int sum = c0 + c1;
    const unsigned int ex = wave_incl_scan_add(sum) - sum;
    wave_lds_sync();
    if (lane == 0) bk[0] = 0u;
    bk[1 + 2 * lane] = ex;
    bk[2 + 2 * lane] = ex + c0;
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned int e = (unsigned int)(r * 64) + lane;
        if (e < have) {
            const unsigned long long k = stage[e];
            tmp[atomicAdd(&bk[1 + bucket_of(k, scale)], 1u)] = k;
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned int e = (unsigned int)(r * 64) + lane;
        if (e < have) {
            const unsigned long long k = tmp[e];
            const unsigned int bb = bucket_of(k, scale);
            const unsigned int lo = bk[bb], hi = bk[bb + 1];
            unsigned int pos = lo;
            for (unsigned int j = lo; j < hi; ++j) pos += tmp[j] < k ? 1u : 0u;
            stage[pos] = k;
        }
    }
    wave_lds_sync();
    return true;
}

// ---- k-NN by selection (K <= 128) ------------------------------------------------------------------------------
// The merge network above is the whole cost of the kernel (VALU 118 % busy, ~1400 instructions per query at K = 51,
// ~1000 of them compare-exchanges): every 64 survivors are sorted and merged into the running list.  Here nothing is
// merged.  Pass 1 walks the same cube (>= 2K points), keeps every candidate key in LDS and counts d2 into 128 buckets
// over the cube's d2 range; a scan of the counts names the bucket b* that holds the K-th -- so the K nearest all
// have bucket <= b*, and (b* + 1) / scale is an upper bound of the K-th distance.  The kept keys with bucket <= b* are
// compacted in place, pass 2 adds what the ball of that bound holds outside the cube (same filter), and the few more
// than K survivors are ordered by the bucket + rank sort (finer buckets over [0, bound]); the first K leave.  Same
// set, same order as the network: both are the K smallest (d2, index) keys of a candidate set that covers the ball of
// the K-th.  A query that does not fit -- cube beyond the tables, more candidates than LDS holds, more survivors than
// the sort takes (many equal distances), bound in the clamped last bucket (queries far outside the grid), fewer than
// K points around -- goes on a list that k_grid_knn_wave works off afterwards.
// SCAP_: survivors the final sort takes (K + the K-th's bucket + what pass 2 adds); the candidate buffer is twice that (the
// sort's second half).  256 for K <= 128, 384 for K <= 256, 768 for K <= 512 -- those two also take cubes of 13 x 13 rows.
template <int SCAP_, bool STORE>
__global__ void __launch_bounds__(256)
k_grid_knn_sel(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
               const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned int* __restrict__ order,
               const unsigned int* __restrict__ n_sorted_ptr, int K, KnnOut out,
               unsigned int* __restrict__ fb_list, unsigned int* __restrict__ fb_count, unsigned int KNN_RUN) {
    constexpr unsigned int SCAP = SCAP_;
    constexpr int CAP = 2 * SCAP_;
    constexpr bool BIG = SCAP_ > 256;
    constexpr int KSEL = BIG ? 6 : 5;  // largest cube half-width: 11 x 11 or 13 x 13 rows
    constexpr int ROWCAP = (2 * KSEL + 1) * (2 * KSEL + 1), SPANCAP = 2 * ROWCAP;
    constexpr int RL = (ROWCAP + 63) / 64;  // rows of the cube a lane looks after
    // candidates of one pass: one (two) planes of span-end bits; K <= 128: 25 KB of LDS, 6 waves per SIMD
    constexpr unsigned int SEL_FLAT_CAP = BIG ? 4096 : 2048;
    constexpr int PLANES = SEL_FLAT_CAP / 2048;
    struct alignas(8) WaveLds {
        unsigned long long cand[CAP];
        unsigned int bk[BUCKET_N + 2], tab_s[SPANCAP], endb[PLANES][64];
        unsigned short tab_o[SPANCAP];  // (flat offsets stay below SEL_FLAT_CAP)
    };
    __shared__ WaveLds lds_all[4];
    WaveLds& L = lds_all[threadIdx.x >> 6];
    const GridParams g = gd->g;
    const float slack = gd->slack;
    const unsigned int n_valid = gd->n_valid;
    const unsigned int ns = *n_sorted_ptr;
    if (n_valid == 0) return;
    const unsigned int lane = threadIdx.x & 63;
    const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));  // (uniform: scalar header loads)
    const unsigned int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int want = (unsigned int)K < n_valid ? K : (int)n_valid;
    // Round 6: a wave takes RUNS of KNN_RUN consecutive queries of the cell-sorted order (runs dealt out strided over the waves, as
    // single queries were).  The K-th neighbour distance is 1-Lipschitz in the query: r_K(q) <= r_K(p) + |q - p|.  So every query
    // after the first of its run starts from a BOUND -- its predecessor's K-th distance plus their separation -- and needs neither the
    // sizing rounds, nor the bucket histogram that finds a bound, nor the compaction behind it: one table over the box of that ball,
    // one walk that keeps what lies inside, the sort.  Exact whatever the bound: the walk collects EVERY reference with d2 <= bound
    // inside a box that covers the ball; with at least `want` of them their `want` smallest are the answer, with fewer (the bound
    // came from rounded arithmetic, or the predecessor gave up) the query takes the full path below.
    // (KNN_RUN: PCC_OPT_KNN_RUN, default 16; 1 = every query takes the full path, round 5's kernel)
    // (a cap by the survivor buffer -- (1 + ratio)^3 x want <= SCAP / 1.15, i.e. 0.31 for K = 100 -- was measured too: corridor K = 100
    // 1.823 -> 1.796 ms, room scan 1.199 -> 1.229: on surfaces a ball holds fewer points than its volume says.  Not kept.)
    // (with the second stage on the stored keys -- survivors beyond SCAP cut at the K-th's bucket instead of handed to the full path --
    // the ratio was measured again, one handle, run 16 against run 1 per process: 0.45 / 0.7 / 1.0 -> corridor K = 51 0.931 / 0.975 /
    // 1.027, K = 100 0.862 / 0.859 / 0.887; room scan K = 51 0.892 / 0.864 / 0.871, K = 100 0.838 / 0.825 / 0.823.  0.45 stays.)
    constexpr float sep_max_ratio = 0.45f;
    const unsigned int nruns = (ns + KNN_RUN - 1) / KNN_RUN;
    for (unsigned int run = wave; run < nruns; run += nwaves) {  // wave-uniform
    float prev_r = -1.f, pqx = 0.f, pqy = 0.f, pqz = 0.f;  // K-th distance (not squared) and position of the run's previous query; < 0: none
    int prev_k = 1;  // half-width the sizing rounds start from: where the run's previous full-path query ended
    for (unsigned int t = run * KNN_RUN; t < min(ns, (run + 1u) * KNN_RUN); ++t) {
        const unsigned int qi = order[t];
        const float4 qv = q[qi];
        const float qx = qv.x, qy = qv.y, qz = qv.z;
        float ux, uy, uz;  // the query in the grid's frame (grid_device.hpp): cells and bounds; the distances take (qx, qy, qz)
        grid_frame(g, qx, qy, qz, ux, uy, uz);
        const int cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
        const int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
        const int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
        const float run_r = prev_r;  // (the predecessor's bound, if any; whatever happens to this query, the next one needs ITS result)
        prev_r = -1.f;
        auto give_up = [&]() {
            if (lane == 0) fb_list[atomicAdd(fb_count, 1u)] = qi;
        };
        int x0 = 0, x1 = -1, y0 = 0, y1 = -1, z0 = 0, z1 = -1;       // box of the pass in hand (cells)
        int ix0 = 1, ix1 = 0, iy0 = 1, iy1 = 0, iz0 = 1, iz1 = 0;  // box already scanned (none yet)
        // span table: (start in cell_refs, flat offset) per non-empty span, and one bit per span END over the flat candidate
        // positions, transposed as in k_grid_radius_fill_wave
        unsigned int nspans = 0, total = 0;  // wave-uniform
        auto table_reset = [&]() {
            nspans = 0;
            total = 0;
            wave_lds_sync();
#pragma unroll
            for (int p = 0; p < PLANES; ++p) L.endb[p][lane] = 0u;
            wave_lds_sync();
        };
        auto table_add = [&](unsigned int s0, unsigned int c) {
            const unsigned int incl = wave_incl_scan_add(c);
            const unsigned long long occ = __ballot(c != 0);
            if (c) {
                const unsigned int slot = lanes_below(occ, nspans);
                const unsigned int off = total + incl - c;
                L.tab_s[slot] = s0;
                const unsigned int e = off + c - 1;
                if (e < SEL_FLAT_CAP) {
                    L.tab_o[slot] = (unsigned short)off;
                    atomicOr(&L.endb[e >> 11][e & 63], 1u << ((e >> 6) & 31));
                }
            }
            nspans += (unsigned int)__popcll(occ);
            total += (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
        };
        // the box minus the box already scanned: a row crossing it gives its left and its right part
        auto build_table = [&]() {
            const int ny = y1 - y0 + 1, nrow = ny * (z1 - z0 + 1);
            const float inv_ny = __builtin_amdgcn_rcpf((float)ny);  // (1 ulp; (r + 0.5) / ny stays 0.5 / ny away from an integer)
            table_reset();
            for (int base = 0; base < nrow; base += 64) {
                const int r = base + (int)lane;
                unsigned int s0 = 0, c0 = 0, s1 = 0, c1 = 0;
                if (r < nrow) {
                    const int zi = (int)(((float)r + 0.5f) * inv_ny);
                    const int z = z0 + zi, y = y0 + (r - zi * ny);
                    const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
                    if (y >= iy0 && y <= iy1 && z >= iz0 && z <= iz1) {
                        if (x0 < ix0) { s0 = cell_start[row + x0]; c0 = cell_start[row + ix0] - s0; }
                        if (ix1 < x1) { s1 = cell_start[row + ix1 + 1]; c1 = cell_start[row + x1 + 1] - s1; }
                    } else {
                        s0 = cell_start[row + x0];
                        c0 = cell_start[row + x1 + 1] - s0;
                    }
                }
                table_add(s0, c0);
                if (__ballot(c1 != 0) != 0ull) table_add(s1, c1);
            }
            wave_lds_sync();
        };
        // the rows of the box x0..z1 clipped to the BALL of squared radius tau around the query (the bound path): a row whose y / z
        // gaps leave nothing of tau is dropped, the others are cut to the chord -- the same gaps, factors and slack as the radius
        // search's rows (k_grid_radius_fill_wave): conservative, every reference with d2 <= tau lies in a listed span.  A ball
        // fills 52 % of its box: half the candidates of the plain box
        auto build_table_ball = [&](float tau) {
            const int ny = y1 - y0 + 1, nrow = ny * (z1 - z0 + 1);
            const float inv_ny = __builtin_amdgcn_rcpf((float)ny);  // (1 ulp; (r + 0.5) / ny stays 0.5 / ny away from an integer)
            auto gap_of = [&](float v, int c, int dim, float org) {  // (boundary cells of the grid are open-ended)
                return fmaxf(fmaxf((c == 0 ? -__builtin_inff() : org + c * g.h) - v,
                                   v - (c == dim - 1 ? __builtin_inff() : org + (c + 1) * g.h)) - slack, 0.f);
            };
            table_reset();
            for (int base = 0; base < nrow; base += 64) {
                const int r = base + (int)lane;
                unsigned int s0 = 0, c0 = 0;
                if (r < nrow) {
                    const int zi = (int)(((float)r + 0.5f) * inv_ny);
                    const int z = z0 + zi, y = y0 + (r - zi * ny);
                    const float gy = gap_of(uy, y, g.dim[1], g.org[1]), gz = gap_of(uz, z, g.dim[2], g.org[2]);
                    const float rem = tau - (gy * gy + gz * gz) * 0.9999f;
                    if (rem >= 0.f) {
                        int xa, xb;
                        cell_range(ux, __builtin_amdgcn_sqrtf(rem) * 1.00001f + slack, g.org[0], g.inv_h, g.dim[0], xa, xb);
                        xa = max(xa, x0);
                        xb = min(xb, x1);
                        if (xa <= xb) {
                            const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
                            s0 = cell_start[row + xa];
                            c0 = cell_start[row + xb + 1] - s0;
                        }
                    }
                }
                table_add(s0, c0);
            }
            wave_lds_sync();
        };
        // the table's candidates 64 at a time: fn(flat position, key, in range)
        auto walk = [&](auto&& fn) {
            unsigned int before = 0, word = 0;
            // two windows per turn, both loads in flight before the first is used
            for (unsigned int B = 0; B < total; B += 128) {
                unsigned int my[2], c[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned int w = (B >> 6) + (unsigned int)u;
                    if ((w & 31u) == 0u) word = L.endb[(w >> 5) % PLANES][lane];
                    const unsigned long long m = __ballot(((word >> (w & 31u)) & 1u) != 0u);
                    my[u] = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, before));
                    before += (unsigned int)__popcll(m);
                    c[u] = B + 64u * (unsigned int)u + lane;
                }
                float4 r4[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned int cc = min(c[u], total - 1u), mm = min(my[u], nspans - 1u);
                    r4[u] = cell_refs[L.tab_s[mm] + (cc - (unsigned int)L.tab_o[mm])];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u == 1 && B + 64u >= total) break;
                    PCC_PAIR(c[u] < total);
                    fn(c[u], make_key(dist2_nc(qx, qy, qz, r4[u]), r4[u]), c[u] < total);
                }
            }
        };
        // what both paths hand to the sort below: the survivors L.cand[0 .. scnt), every one of them within `bound`
        unsigned int scnt = 0;
        float bound = 0.f;
        bool by_bound = false;
        // a window's keys below `tkey` appended to the survivors
        unsigned long long tkey = 0ull;
        auto keep_below = [&](unsigned int, unsigned long long key, bool in) {
            const bool pass = in && key < tkey;
            const unsigned long long mask = __ballot(pass);
            const unsigned int slot = lanes_below(mask, scnt);
            if (pass && slot < (unsigned int)CAP) L.cand[slot] = key;
            scnt += (unsigned int)__popcll(mask);
        };
        // the bucket that holds the `want`-th of the keys counted into L.bk: first one whose running count reaches it; s1 = keys
        // with a bucket <= b* (at least `want` keys were counted, so some lane reaches it)
        auto find_bstar = [&](unsigned int& bstar, unsigned int& s1) {
            const unsigned int c0 = L.bk[1 + 2 * lane], c1 = L.bk[2 + 2 * lane];
            const unsigned int incl = wave_incl_scan_add(c0 + c1);
            const unsigned long long reached = __ballot(incl >= (unsigned int)want);
            const int fl = __builtin_ctzll(reached);
            const unsigned int i_fl = (unsigned int)__builtin_amdgcn_readlane((int)incl, fl);
            const unsigned int c1_fl = (unsigned int)__builtin_amdgcn_readlane((int)c1, fl);
            const bool first_half = i_fl - c1_fl >= (unsigned int)want;
            bstar = 2u * (unsigned int)fl + (first_half ? 0u : 1u);
            s1 = first_half ? i_fl - c1_fl : i_fl;
        };
        // the keys L.cand[0 .. count) with bucket <= b*, compacted to the front of the buffer (in place: a window is read whole
        // before its survivors are written, and they land at or before their own position); scnt = how many
        auto compact_stored = [&](unsigned int count, float scale1, unsigned int bstar) {
            scnt = 0;
            for (unsigned int B = 0; B < count; B += 64) {
                const unsigned int c = B + lane;
                const unsigned long long key = c < count ? L.cand[c] : ~0ull;
                const bool pass = c < count && bucket_of(key, scale1) <= bstar;
                const unsigned long long mask = __ballot(pass);
                wave_lds_sync();
                if (pass) L.cand[lanes_below(mask, scnt)] = key;
                scnt += (unsigned int)__popcll(mask);
                wave_lds_sync();
            }
        };
        // ---- the bound path (every query of a run but the first)
        if (run_r >= 0.f) {
            const float sx = qx - pqx, sy = qy - pqy, sz = qz - pqz;
            const float sep = __builtin_amdgcn_sqrtf((sx * sx + sy * sy) + sz * sz);  // (1 ulp: the factor below covers it)
            const float rb = (run_r + sep) * 1.00001f + slack;  // radius that holds `want` references
            const float tau = rb * rb;
            const float rbox = rb * 1.00001f + slack;  // (cells that can hold a reference with d2 <= tau: as the ball pass below)
            // Only where the bound is TIGHT: the ball of r + sep holds (1 + sep / r)^3 times the K references wanted.  Inside a
            // dense object the K-th neighbour is a cell away and so is the next query of the cell order (sep / r ~ 0.7: five times
            // the references, more than the survivor buffer takes -- the walk is wasted and the full path follows); in sparse
            // regions and on surfaces sep / r is 0.1-0.5.  Measured on one box, K = 51 at 1M, search kernels us, ratio 0.15 / 0.3 / 0.45 /
            // 0.6 against every query on its own: corridor 1210 / 1216 / 1212 / 1238 (1200), room scan 1068 / 1001 / 991 / 976 (1075);
            // K = 100 corridor 1838 / 1804 / 1811 / 1883 (1932).  Without the test (every query of a run on the bound path): corridor
            // 1442, room 968.
            if (tau < 3.0e38f && sep <= sep_max_ratio * run_r) {
                cell_range(ux, rbox, g.org[0], g.inv_h, g.dim[0], x0, x1);
                cell_range(uy, rbox, g.org[1], g.inv_h, g.dim[1], y0, y1);
                cell_range(uz, rbox, g.org[2], g.inv_h, g.dim[2], z0, z1);
                if ((y1 - y0 + 1) * (z1 - z0 + 1) <= ROWCAP) {
                    build_table_ball(tau);
                    if (total >= (unsigned int)want && total <= SEL_FLAT_CAP) {
                        tkey = ((unsigned long long)__float_as_uint(tau) + 1ull) << 32;  // every key with d2 <= tau
                        walk(keep_below);
                        wave_lds_sync();
                        if (scnt >= (unsigned int)want && scnt <= SCAP) {
                            bound = tau;
                            by_bound = true;
                        } else if (scnt > SCAP && scnt <= (unsigned int)CAP) {
                            // more than the sort takes, all of them kept (the buffer's second half is free until the sort): the
                            // buckets of the full path's pass 1 over [0, tau] on the stored keys cut them at the `want`-th's bucket
                            const float sc = (float)BUCKET_N * __builtin_amdgcn_rcpf(tau);  // (any scale serves, the bound below takes the same one)
                            L.bk[1 + 2 * lane] = 0u;
                            L.bk[2 + 2 * lane] = 0u;
                            wave_lds_sync();
                            for (unsigned int c = lane; c < scnt; c += 64) atomicAdd(&L.bk[1 + bucket_of(L.cand[c], sc)], 1u);
                            wave_lds_sync();
                            unsigned int bstar, s1;
                            find_bstar(bstar, s1);
                            if (s1 <= SCAP) {
                                compact_stored(scnt, sc, bstar);
                                bound = fminf(tau, (float)(bstar + 1) / sc * 1.00001f);
                                by_bound = true;
                            }
                        }
                    }
                }
            }
            scnt = by_bound ? scnt : 0u;
        }
        if (!by_bound) {
        // the smallest cube that holds at least 2 x `want` points (as k_grid_knn_wave).  A lane looks after rows `lane` and
        // `lane + 64` of the cube (11 x 11 rows at most): the bounds it reads for the count ARE the spans of pass 1
        // The rounds start from the half-width the run's previous full-path query ended with (one round instead of two or three; a
        // cube that came out too large for the buffers starts over from 3 x 3 x 3)
        constexpr unsigned int CUBE_MAX = STORE ? (unsigned int)CAP : SEL_FLAT_CAP;
        int k = prev_k;
        unsigned int cnt = 0, rs0[RL], rc[RL];
        for (bool from_prev = k > 1;; ++k) {
            const int x0 = max(cx - k, 0), x1 = min(cx + k, g.dim[0] - 1);
            const int y0 = max(cy - k, 0), y1 = min(cy + k, g.dim[1] - 1);
            const int z0 = max(cz - k, 0), z1 = min(cz + k, g.dim[2] - 1);
            const int ny = y1 - y0 + 1, nrow = ny * (z1 - z0 + 1);
            const float inv_ny = __builtin_amdgcn_rcpf((float)ny);  // (1 ulp; (r + 0.5) / ny stays 0.5 / ny away from an integer)
            unsigned int mine = 0;
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                const int r = (int)lane + 64 * i;
                rs0[i] = 0;
                rc[i] = 0;
                if (r < nrow) {
                    const int zi = (int)(((float)r + 0.5f) * inv_ny);  // r / ny (exact: r < 192, ny <= 13)
                    const unsigned int row = ((unsigned int)(z0 + zi) * g.dim[1] + (y0 + (r - zi * ny))) * g.dim[0];
                    rs0[i] = cell_start[row + x0];
                    rc[i] = cell_start[row + x1 + 1] - rs0[i];
                }
                mine += rc[i];
            }
            cnt = (unsigned int)__builtin_amdgcn_readlane((int)wave_incl_scan_add(mine), 63);
            if (from_prev && cnt > CUBE_MAX) {
                from_prev = false;
                k = 0;
                continue;
            }
            from_prev = false;
            if (cnt >= 2u * (unsigned int)want || k >= KSEL) break;  // (the table's rows)
        }
        // (a cube with several times the points it needs: the next query tries one smaller)
        prev_k = cnt >= 6u * (unsigned int)want && k > 1 ? k - 1 : k;
        if (cnt < (unsigned int)want || cnt > CUBE_MAX) { give_up(); continue; }
        x0 = max(cx - k, 0); x1 = min(cx + k, g.dim[0] - 1);
        y0 = max(cy - k, 0); y1 = min(cy + k, g.dim[1] - 1);
        z0 = max(cz - k, 0); z1 = min(cz + k, g.dim[2] - 1);
        ix0 = iy0 = iz0 = 1; ix1 = iy1 = iz1 = 0;
        // ---- pass 1: every candidate of the cube kept, d2 counted into buckets over the cube's d2 range
        {
            const float reach = (float)(k + 1) * g.h;
            const float span1 = 3.03f * reach * reach, scale1 = (float)BUCKET_N / span1;
            table_reset();
#pragma unroll
            for (int i = 0; i < RL; ++i)
                if (i == 0 || (2 * k + 1) * (2 * k + 1) > 64 * i) table_add(rs0[i], rc[i]);
            L.bk[1 + 2 * lane] = 0u;
            L.bk[2 + 2 * lane] = 0u;
            wave_lds_sync();
            // (STORE: the keys are kept for the compaction below; otherwise -- K > 64, cubes of up to a thousand points, whose
            // keys would cost half the resident waves -- the cube is walked a second time)
            walk([&](unsigned int c, unsigned long long key, bool in) {
                if (in) {
                    if constexpr (STORE) L.cand[c] = key;
                    atomicAdd(&L.bk[1 + bucket_of(key, scale1)], 1u);
                }
            });
            wave_lds_sync();
            unsigned int bstar, s1;  // the bucket that holds the `want`-th; cube candidates with bucket <= b*
            find_bstar(bstar, s1);
            if (bstar >= BUCKET_N - 1 || s1 > SCAP) { give_up(); continue; }
            // every d2 >= bound has a bucket > b* (span1 / BUCKET_N is 1 / scale1 to an ulp: the factor covers it, no division)
            bound = (float)(bstar + 1) * (span1 * (1.0f / (float)BUCKET_N)) * 1.00001f;
            // the kept keys with bucket <= b* to the front of the buffer
            scnt = 0;
            // a window's keys with bucket <= b* appended to the survivors
            auto keep = [&](unsigned int, unsigned long long key, bool in) {
                const bool pass = in && bucket_of(key, scale1) <= bstar;
                const unsigned long long mask = __ballot(pass);
                const unsigned int slot = lanes_below(mask, scnt);
                if (pass && slot < SCAP) L.cand[slot] = key;
                scnt += (unsigned int)__popcll(mask);
            };
            if constexpr (STORE) {
                compact_stored(total, scale1, bstar);
            } else {
                walk(keep);  // (s1 <= SCAP: they all fit)
                wave_lds_sync();
            }
            // ---- pass 2: what the ball of the bound holds outside the cube
            const float lb2 = outside_bound2(ux, uy, uz, x0, x1, y0, y1, z0, z1, g, slack);
            bool fits = true;
            if (!(bound < lb2)) {
                // (hardware sqrt, 1 ulp, no denormals: a bound below 1e-30 is taken as 1e-30 -- larger, so still covering)
                const float rb = __builtin_amdgcn_sqrtf(fmaxf(bound, 1.0e-30f)) * 1.00001f + slack;
                int a0, a1, b0, b1, e0, e1;
                cell_range(ux, rb, g.org[0], g.inv_h, g.dim[0], a0, a1);
                cell_range(uy, rb, g.org[1], g.inv_h, g.dim[1], b0, b1);
                cell_range(uz, rb, g.org[2], g.inv_h, g.dim[2], e0, e1);
                ix0 = x0; ix1 = x1; iy0 = y0; iy1 = y1; iz0 = z0; iz1 = z1;
                x0 = min(x0, a0); x1 = max(x1, a1); y0 = min(y0, b0); y1 = max(y1, b1); z0 = min(z0, e0); z1 = max(z1, e1);
                if ((y1 - y0 + 1) * (z1 - z0 + 1) > ROWCAP) fits = false;
                if (fits) {
                    build_table();
                    if (total > SEL_FLAT_CAP) fits = false;
                }
                if (fits) {
                    walk(keep);
                    wave_lds_sync();
                    if (scnt > SCAP) fits = false;
                }
            }
            if (!fits) { give_up(); continue; }
        }
        }  // (the full path)
        {
            // ---- the survivors in order: finer buckets over [0, bound]; a crowded bucket (equal distances) -> the network
            const float scale2 = (float)BUCKET_N * __builtin_amdgcn_rcpf(bound);  // (an ordering only: any scale serves)
            unsigned long long* tmp = L.cand + SCAP;
            bool in_order = scnt < 2;
            if (!in_order) {
                if (scnt <= 64) in_order = bucket_sort_lds<1>(L.cand, tmp, L.bk, scnt, scale2, lane);
                else if (scnt <= 128) in_order = bucket_sort_lds<2>(L.cand, tmp, L.bk, scnt, scale2, lane);
                else if (scnt <= 256) in_order = bucket_sort_lds<4>(L.cand, tmp, L.bk, scnt, scale2, lane);
                else if constexpr (BIG) in_order = bucket_sort_lds<(int)SCAP / 64>(L.cand, tmp, L.bk, scnt, scale2, lane);
            }
            // (entries past `want` -- K beyond the number of valid references -- are "nothing found")
            const size_t row0 = (size_t)qi * K;
            unsigned long long kth = ~0ull;  // the `want`-th key of the finished row (wave-uniform)
            if (in_order) {
                for (unsigned int e = lane; e < (unsigned int)K; e += 64) knn_emit(out, row0 + e, e < (unsigned int)want ? L.cand[e] : ~0ull);
                kth = L.cand[want - 1];
            } else if constexpr (BIG) {
                give_up();  // (a crowded bucket among more keys than four registers per lane hold: the merge kernel's)
                continue;
            } else {
                unsigned long long v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (unsigned int)(r * 64) + lane < scnt ? L.cand[r * 64 + lane] : ~0ull;
                bitonic_sort_regs<4>(v, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned int e = (unsigned int)(r * 64) + lane;
                    if (e < (unsigned int)K) knn_emit(out, row0 + e, e < (unsigned int)want ? v[r] : ~0ull);
                    if (r == ((want - 1) >> 6)) kth = shfl_u64(v[r], (want - 1) & 63);
                }
            }
            // the next query of the run starts from this one's K-th distance (a full row only: `want` == K real neighbours)
            if (want == K && !key_none(kth)) {
                prev_r = __builtin_amdgcn_sqrtf(__uint_as_float((unsigned int)(kth >> 32)));
                pqx = qx; pqy = qy; pqz = qz;
            }
        }
    }
    }  // (runs)
}

// keys: the rows as search keys; or keys == nullptr and idx_out / d2_out (either may be null): the rows delivered as
// indices and squared distances (K <= 512 only -- check with grid_knn_delivers)
bool grid_knn_delivers(int K) { return K <= 512; }
int grid_knn(pcc_index* ix, const float4* q, size_t nq, int K, unsigned long long* keys, int32_t* idx_out, float* d2_out) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)nq;
    const KnnOut out{keys, keys ? nullptr : idx_out, keys ? nullptr : d2_out};
    if (K > 512) {
        if (!keys) { set_error("grid_knn: K > 512 needs the key array"); return PCC_ERR_INVALID; }
        PCC_HIP(hipMemsetAsync(keys, 0xff, nq * (size_t)K * sizeof(unsigned long long), s));
    } else {
        // (the wave kernels write every entry of every valid query's row; only the rows of non-finite queries are left)
        hipLaunchKernelGGL(k_knn_fill_invalid, dim3((n + 255) / 256), dim3(256), 0, s, q, n, K, out, ix->d_grid.as<GridDev>());
        PCC_HIP(hipGetLastError());
    }
    unsigned int *order = nullptr, *n_sorted = nullptr;
    PCC_TRY(grid_sort_queries(ix, q, nq, &order, &n_sorted));
    ev_mark(ix, EV_MAIN0);
    if (K <= 512) {
        unsigned int gw = (n + 3) / 4;  // one wave per query, 4 waves per workgroup, waves loop
        if (gw > 8192) gw = 8192;
        // selection kernel first; the queries it hands back are worked off by the merge kernel below
        const unsigned int* work = order;
        const unsigned int* n_work = n_sorted;
        if (ix->opt.knn_kernel != 0) {
            // (K > 64: keeping a thousand keys per wave leaves 3 waves per SIMD and the VALU 60 % busy -- 2.66 ms at 1M x K = 100;
            // the 512-key form first and only its returns through a 1024-key one: 3.18 ms -- half the cubes hold more than 512)
            PCC_TRY(ix->knn_fb.reserve((nq + 1) * sizeof(unsigned int)));
            unsigned int* fb = ix->knn_fb.as<unsigned int>();
            PCC_HIP(hipMemsetAsync(fb + nq, 0, sizeof(unsigned int), s));
#define PCC_LAUNCH_SEL(SCAP_, STORE_)                                                                                     \
    hipLaunchKernelGGL((k_grid_knn_sel<SCAP_, STORE_>), dim3(gw), dim3(256), 0, s, ix->cell_refs.as<float4>(),            \
                       ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, order, n_sorted, K, out, fb, fb + nq, \
                       (unsigned int)ix->opt.knn_run)
            if (K <= 64) PCC_LAUNCH_SEL(256, true);
            else if (K <= 128) PCC_LAUNCH_SEL(256, false);
            else if (K <= 256) PCC_LAUNCH_SEL(384, false);
            else PCC_LAUNCH_SEL(768, false);
#undef PCC_LAUNCH_SEL
            PCC_HIP(hipGetLastError());
            work = fb;
            n_work = fb + nq;
        }
#define PCC_LAUNCH_KNN(KR)                                                                                         \
    hipLaunchKernelGGL((k_grid_knn_wave<KR>), dim3(gw), dim3(256), 0, s, ix->cell_refs.as<float4>(),               \
                       ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, work, n_work, K, out)
        if (K <= 64) PCC_LAUNCH_KNN(1);
        else if (K <= 128) PCC_LAUNCH_KNN(2);
        else if (K <= 256) PCC_LAUNCH_KNN(4);
        else PCC_LAUNCH_KNN(8);
#undef PCC_LAUNCH_KNN
    } else
    hipLaunchKernelGGL(k_grid_knn, dim3((n + 255) / 256), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                       ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, order, n_sorted, n, K, keys);
    PCC_HIP(hipGetLastError());
    ev_mark(ix, EV_MAIN1);
    return PCC_OK;
}

constexpr unsigned int ROW_SORT_MAX = 512;  // rows up to this length are sorted by k_sort_rows (8 registers per lane)

// ---- radius search ---------------------------------------------------------------------------
// FILL = false: counts[i] = #refs with d2 < r2 (strict).
// FILL = true : keys written at offsets[i]; with sorted the row is then ordered by (d2, position)
//               with an in-place insertion sort (rows are short: tens to a few hundred entries).
// which count kernel serves a radius: a wave per query pays when the ball's box holds hundreds of candidates -- cells of
// the box x the filling of an OCCUPIED cell (n_valid / non-empty cells: a scan's points pile up on surfaces, an average
// over the bounding box says nothing about where the queries are).  Decided on the device -- cell size and cell count are
// computed there and the host does not wait for them: both kernels are launched, one of them returns at once.
// (5M object-layer points, r = 0.05, 83 hits: 2.9 against 3.5 ms; 1M corridor, r = 0.2, 210 hits: 0.75 against 1.13;
// 1M corridor, r = 0.05, 5.5 hits: 0.40 against 0.25 -- the lane form keeps those)
__device__ __forceinline__ bool radius_count_by_wave(const GridDev* gd, float r, const unsigned int* occupied) {
    const float c = 2.f * r * gd->g.inv_h + 1.f;
    const unsigned int occ = *occupied;
    return c * c * c * ((float)gd->n_valid / (float)(occ ? occ : 1u)) >= 200.f;
}
__global__ void __launch_bounds__(256)
k_count_occupied(const unsigned int* __restrict__ cell_start, const GridDev* __restrict__ gd, unsigned int* __restrict__ out) {
    const unsigned int nc = (unsigned int)gd->g.ncells;
    unsigned int mine = 0;
    for (unsigned int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x)
        mine += cell_start[c + 1] > cell_start[c] ? 1u : 0u;
    mine = wave_incl_scan_add(mine);
    if ((threadIdx.x & 63) == 63 && mine) atomicAdd(out, mine);
}
template <bool FILL>
__global__ void __launch_bounds__(256)
k_grid_radius(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
              const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned int* __restrict__ order,
              const unsigned int* __restrict__ n_sorted_ptr, unsigned int /*n*/, float r, float r2,
              int32_t* __restrict__ counts, const int64_t* __restrict__ offsets,
              unsigned long long* __restrict__ keys, int /*sorted*/, const unsigned int* __restrict__ occupied) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *n_sorted_ptr) return;
    if (!FILL && radius_count_by_wave(gd, r, occupied)) return;  // (k_grid_radius_fill_wave<true> counts)
    const unsigned int qi = order[t];
    const float4 qv = q[qi];
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    float ux, uy, uz;  // the query in the grid's frame: cells; the distances take (qx, qy, qz)
    grid_frame(g, qx, qy, qz, ux, uy, uz);
    int x0, x1, y0, y1, z0, z1;
    const float rr = r + slack;  // cells that can hold a point within r (conservative)
    cell_range(ux, rr, g.org[0], g.inv_h, g.dim[0], x0, x1);
    cell_range(uy, rr, g.org[1], g.inv_h, g.dim[1], y0, y1);
    cell_range(uz, rr, g.org[2], g.inv_h, g.dim[2], z0, z1);
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
    }
}

// ---- first point within a radius (reference src/comparator.cpp:696-713) -----------------------------
// lowest position whose double-precision distance (float differences, double squares/sum/sqrt) is
// < radius.  The cells are chosen with the float radius plus slack (a superset); the double test
// decides.
__global__ void __launch_bounds__(256)
k_grid_first_within(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
                    const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned int* __restrict__ order,
                    const unsigned int* __restrict__ n_sorted_ptr, double radius, int32_t* __restrict__ idx) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *n_sorted_ptr) return;
    const unsigned int qi = order[t];
    const float4 qv = q[qi];
    int x0, x1, y0, y1, z0, z1;
    const float rr = (float)radius * 1.000001f + slack;
    float ux, uy, uz;  // the query in the grid's frame
    grid_frame(g, qv.x, qv.y, qv.z, ux, uy, uz);
    cell_range(ux, rr, g.org[0], g.inv_h, g.dim[0], x0, x1);
    cell_range(uy, rr, g.org[1], g.inv_h, g.dim[1], y0, y1);
    cell_range(uz, rr, g.org[2], g.inv_h, g.dim[2], z0, z1);
    unsigned int first = 0xffffffffu;
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y) {
            const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
            const unsigned int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
            for (unsigned int p = s; p < e; ++p) {
                const float4 rp = cell_refs[p];
                const unsigned int pos = (unsigned int)__float_as_int(rp.w);
                if (pos >= first) continue;
                const double dx = (double)(qv.x - rp.x), dy = (double)(qv.y - rp.y), dz = (double)(qv.z - rp.z);
                if (sqrt(dx * dx + dy * dy + dz * dz) < radius) first = pos;
            }
        }
    idx[qi] = first == 0xffffffffu ? -1 : (int32_t)first;
}

int grid_first_within(pcc_index* ix, const float4* q, size_t nq, double radius, int32_t* idx) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)nq;
    unsigned int *order = nullptr, *n_sorted = nullptr;
    PCC_TRY(grid_sort_queries(ix, q, nq, &order, &n_sorted));
    PCC_HIP(hipMemsetAsync(idx, 0xff, nq * sizeof(int32_t), s));  // non-finite queries: -1
    ev_mark(ix, EV_MAIN0);
    hipLaunchKernelGGL(k_grid_first_within, dim3((n + 255) / 256), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                       ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, order, n_sorted, radius, idx);
    PCC_HIP(hipGetLastError());
    ev_mark(ix, EV_MAIN1);
    return PCC_OK;
}

template <int R>
__device__ __forceinline__ void sort_row(unsigned long long* __restrict__ row, unsigned int len, unsigned int lane) {
    unsigned long long v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = (unsigned int)(r * 64) + lane < len ? row[r * 64 + lane] : ~0ull;
    bitonic_sort_regs<R>(v, lane);
#pragma unroll
    for (int r = 0; r < R; ++r)
        if ((unsigned int)(r * 64) + lane < len) row[r * 64 + lane] = v[r];
}

// ---- rows longer than the registers hold ---------------------------------------------------------------------
// A bitonic network over the row in global memory, in place, by the row's wave: O(len log^2 len) against the O(len^2)
// moves of the one-lane insertion sort this replaces (a query whose ball holds 150k references -- a radius as large
// as the cloud -- did not come back within minutes; tools/fuzz_gpu.py found it).  The all-ascending form of the
// network (every merge level starts with a MIRRORED compare, i against block_end - i, then plain half-cleaners) lets a
// row of any length be treated as padded with +inf up to a power of two without storing the padding: a compare whose
// upper index lies beyond the row is skipped.  Steps with a stride below 512 work inside aligned chunks of 512 keys, where
// the register network sorts the chunk outright.  Accesses bypass the L1 (agent-scope relaxed atomics), a fence
// separates the steps: lanes exchange data through memory.
__device__ __forceinline__ unsigned long long row_ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void row_st(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void row_cx(unsigned long long* row, unsigned int i, unsigned int p) {
    const unsigned long long a = row_ld(row + i), b = row_ld(row + p);
    if (b < a) { row_st(row + i, b); row_st(row + p, a); }
}
__device__ __forceinline__ void sort_chunks_512(unsigned long long* row, unsigned int len, unsigned int lane) {
    for (unsigned int c0 = 0; c0 < len; c0 += 512) {
        const unsigned int cl = min(512u, len - c0);
        unsigned long long v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (unsigned int)(r * 64) + lane < cl ? row_ld(row + c0 + r * 64 + lane) : ~0ull;
        bitonic_sort_regs<8>(v, lane);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if ((unsigned int)(r * 64) + lane < cl) row_st(row + c0 + r * 64 + lane, v[r]);
    }
    __threadfence();
}
__device__ __noinline__ void sort_long_row(unsigned long long* row, unsigned int len, unsigned int lane) {
    unsigned long long n2 = 1024;
    while (n2 < len) n2 <<= 1;
    __threadfence();  // (the fill kernel's keys are in memory; start from L2)
    sort_chunks_512(row, len, lane);
    for (unsigned long long k = 1024; k <= n2; k <<= 1) {
        const unsigned long long half = k >> 1;
        for (unsigned long long t = lane; t < n2 / 2; t += 64) {  // mirrored compare across the two halves of a block
            const unsigned long long blk = t / half, o = t - blk * half;
            const unsigned long long i = blk * k + o, p = blk * k + (k - 1 - o);
            if (p < len) row_cx(row, (unsigned int)i, (unsigned int)p);
        }
        __threadfence();
        for (unsigned long long j = k >> 2; j >= 512; j >>= 1) {  // half-cleaners down to the chunk size
            for (unsigned long long t = lane; t < n2 / 2; t += 64) {
                const unsigned long long i = (t / j) * 2 * j + (t % j), p = i + j;
                if (p < len) row_cx(row, (unsigned int)i, (unsigned int)p);
            }
            __threadfence();
        }
        sort_chunks_512(row, len, lane);
    }
}

// ---- wave-cooperative fill ------------------------------------------------------------------------
// One WAVE per query for the fill pass: the rows of cells the r-ball touches go into an LDS table (lanes over
// rows), their points are taken 64 at a time across row boundaries (as in k_grid_knn_wave), tested, and the hits
// are written with a ballot-compacted, coalesced store.  One lane per query walked hundreds of candidates
// alone and scattered 8-byte stores over 64 different rows per instruction: 10 of the 13.7 ms of an unsorted
// 5M x 83 search.
constexpr int RAD_ROWCAP = 11 * 11;
// (8 waves per SIMD: the kernel waits on dependent loads most of its time -- 5.6 -> 4.9 ms at 5M x 83 against 6 waves; the
// rows kept in LDS were halved to 256 and three registers spill to make room)
// COUNT: the same walk, hits only counted (counts[query]; no offsets, no rows) -- the count pass of a search whose balls
// hold hundreds of candidates, where one lane per query walks them alone
template <bool COUNT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_grid_radius_fill_wave(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
                        const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned int* __restrict__ order,
                        const unsigned int* __restrict__ n_sorted_ptr, float r, float r2, const int64_t* __restrict__ offsets,
                        unsigned long long* __restrict__ keys, int32_t* __restrict__ idx_out, float* __restrict__ d2_out,
                        int sorted, unsigned int* __restrict__ long_list, unsigned int* __restrict__ long_count,
                        int32_t* __restrict__ counts, const unsigned int* __restrict__ occupied) {
    // FUSED (idx_out / d2_out given): a row of up to ROW_LDS_MAX neighbours never leaves the chip between the search and
    // the caller's arrays -- its hits gather in LDS, are sorted in registers (PCL's sorted results) and go out as index
    // and squared distance in two coalesced stores.  Before: keys to memory (8 B), sorted in place by a second kernel
    // (16 B), unpacked by a third (16 B).  Longer rows keep that route (k_sort_rows).
    // per wave: the row tables of the search (dead once the last window is through) double as the bucket sort's second
    // buffer, and its 130 counters sit in the upper half of the stage (rows it sorts fill at most the lower half)
    struct alignas(8) Tables { unsigned int tab_s[RAD_ROWCAP + 1], tab_o[RAD_ROWCAP + 1], win[64], endb[4][64]; };
    static_assert(sizeof(Tables) >= BUCKET_ROW_MAX * sizeof(unsigned long long), "the tables must hold the sort's second buffer");
    __shared__ unsigned long long stage_all[4][ROW_LDS_MAX + (BUCKET_N + 2) / 2 + 1];
    __shared__ Tables tables_all[4];
    Tables& tb = tables_all[threadIdx.x >> 6];
    unsigned int (*endb)[64] = tb.endb;
    unsigned long long* stage = stage_all[threadIdx.x >> 6];
    const bool fused = !COUNT && (idx_out != nullptr || d2_out != nullptr);
    unsigned int* tab_s = tb.tab_s;
    unsigned int* tab_o = tb.tab_o;
    unsigned int* win = tb.win;
    const GridParams g = gd->g;
    const float slack = gd->slack;
    const unsigned int ns = *n_sorted_ptr;
    if (COUNT && !radius_count_by_wave(gd, r, occupied)) return;  // (k_grid_radius<false> counts)
    const unsigned int lane = threadIdx.x & 63;
    // (tried and measured no gain, 5M x 83: a launch of exactly the resident workgroups so that the waves form a band
    // marching through the cell-sorted queries, with or without one eighth of the order per XCD; 2 / 8 / 16 waves per
    // workgroup for more L1 sharing between neighbouring queries; streaming stores)
    // (wave-uniform by construction; said to the compiler, the per-query header -- order, query, offsets -- goes through
    // scalar loads and scalar address arithmetic)
    const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const unsigned int nwaves = (gridDim.x * blockDim.x) >> 6;
    const float bscale = (float)BUCKET_N / r2;
    // The per-query header (query index, packed query, two row offsets) is a chain of two dependent loads at the head of a
    // chain of five more (bounds, windows) -- and the kernel waits most of its time.  It is fetched one query ahead instead:
    // at the top of a query's turn ONE vector load picks up the next query's header (lanes 0-3 its packed coordinates, 4-7
    // its two offsets) and, in lane 8, the query index two turns ahead; it lands while this query's rows are walked and is
    // read out with v_readlane at the end of the turn.  Being older than every load of the turn it also never waits behind
    // the previous row's stores (gfx9: loads and stores share vmcnt, in order).
    unsigned int t = wave;
    if (t >= ns) return;
    unsigned int qi = order[t];
    unsigned int qi_n = order[min(t + nwaves, ns - 1u)];
    float qx, qy, qz;
    int64_t row_beg = 0, row_end = 0;
    {
        const float4 qv = q[qi];
        qx = qv.x; qy = qv.y; qz = qv.z;
        if constexpr (!COUNT) {
            row_beg = offsets[qi];
            row_end = offsets[qi + 1];
        }
    }
    for (;;) {  // wave-uniform
        unsigned int hv = 0u;
        {
            const unsigned int t2 = min(t + 2u * nwaves, ns - 1u);
            const unsigned int* hp = lane < 4 ? reinterpret_cast<const unsigned int*>(q + qi_n) + lane
                                   : lane < 8 ? reinterpret_cast<const unsigned int*>(offsets + qi_n) + (lane - 4)
                                              : order + t2;
            if (lane < 9 && !(COUNT && lane >= 4 && lane < 8)) hv = *hp;
        }
        int x0, x1, y0, y1, z0, z1;
        const float rr = r + slack;
        float ux, uy, uz;  // this turn's query in the grid's frame: cells, gaps, chords; the distances take (qx, qy, qz)
        grid_frame(g, qx, qy, qz, ux, uy, uz);
        cell_range(ux, rr, g.org[0], g.inv_h, g.dim[0], x0, x1);
        cell_range(uy, rr, g.org[1], g.inv_h, g.dim[1], y0, y1);
        cell_range(uz, rr, g.org[2], g.inv_h, g.dim[2], z0, z1);
        const unsigned int row_len = (unsigned int)(row_end - row_beg);
        const bool in_lds = fused && row_len <= ROW_LDS_MAX;
        if (fused && !in_lds && lane == 0) long_list[atomicAdd(long_count, 1u)] = qi;  // left as keys for k_sort_rows
        unsigned long long* row_out = in_lds ? stage : keys + row_beg;
        unsigned int written = 0;  // wave-uniform
        const int ny = y1 - y0 + 1, nrow = ny * (z1 - z0 + 1);
        // the rows in chunks of at most RAD_ROWCAP (one chunk unless the radius spans more than 11 cells)
        for (int rbase = 0; rbase < nrow; rbase += RAD_ROWCAP) {
            const int rchunk = min(RAD_ROWCAP, nrow - rbase);
            unsigned int nspans = 0, total = 0;
            wave_lds_sync();
            if (lane < 64) { endb[0][lane] = 0u; endb[1][lane] = 0u; endb[2][lane] = 0u; endb[3][lane] = 0u; }
            wave_lds_sync();
            // a row's cells the BALL reaches, given the squares of its y and z gaps: none if they leave nothing of r^2,
            // else clipped to the chord (same slack as the cell range above).  (v_sqrt_f32 itself, 1 ulp: the factor
            // 1.00001 and the slack absorb it; the libm form spends 20 instructions on the last bit)
            auto span_of = [&](bool valid, int y, int z, float g2sum, unsigned int& s0, unsigned int& cnt) {
                s0 = 0;
                cnt = 0;
                const float rem = rr * rr - g2sum * 0.9999f;
                if (valid && rem >= 0.f) {
                    int xa, xb;
                    cell_range(ux, __builtin_amdgcn_sqrtf(rem) * 1.00001f + slack, g.org[0], g.inv_h, g.dim[0], xa, xb);
                    xa = max(xa, x0);
                    xb = min(xb, x1);
                    if (xa <= xb) {
                        const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
                        s0 = cell_start[row + xa];
                        cnt = cell_start[row + xb + 1] - s0;
                    }
                }
            };
            auto record = [&](unsigned int s0, unsigned int cnt) {
                const unsigned int incl = wave_incl_scan_add(cnt);
                const unsigned long long occ = __ballot(cnt != 0);
                if (cnt) {
                    const unsigned int slot = lanes_below(occ, nspans);
                    const unsigned int off = total + incl - cnt;
                    tab_s[slot] = s0;
                    tab_o[slot] = off;
                    // one bit per span END over the flat candidate positions, transposed (grid.hip, k_grid_nn1_flat2): word
                    // (p mod 64) of plane (p / 2048), bit (p / 64) mod 32 -- a candidate's span = ends before it
                    const unsigned int e = off + cnt - 1;
                    if (e < RAD_FLAT_CAP) atomicOr(&endb[e >> 11][e & 63], 1u << ((e >> 6) & 31));
                }
                nspans += (unsigned int)__popcll(occ);
                total += (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
            };
            // gap of the coordinate v to cell c of an axis, shrunk by the slack (boundary cells of the grid are open-ended)
            auto gap_of = [&](float v, int c, int dim, float org) {
                return fmaxf(fmaxf((c == 0 ? -__builtin_inff() : org + c * g.h) - v,
                                   v - (c == dim - 1 ? __builtin_inff() : org + (c + 1) * g.h)) - slack, 0.f);
            };
            if (ny <= 8 && nrow <= 8 * ny) {
                // up to 8 x 8 rows (the usual radius: a few cells): lane = 8 * z + y, no division; the gaps depend on one
                // axis each, so lanes 0-7 work out the y gaps, lanes 8-15 the z gaps -- one evaluation -- and every lane
                // picks up its two (ds_bpermute).  Same values, same row order as the general form below.
                const bool zlane = (lane & 8u) != 0u;
                const float gp = gap_of(zlane ? uz : uy, (zlane ? z0 : y0) + (int)(lane & 7u), zlane ? g.dim[2] : g.dim[1],
                                        zlane ? g.org[2] : g.org[1]);
                const int g2 = __float_as_int(gp * gp);
                const float gy2 = __int_as_float(__builtin_amdgcn_ds_bpermute((int)((lane & 7u) << 2), g2));
                const float gz2 = __int_as_float(__builtin_amdgcn_ds_bpermute((int)((8u + (lane >> 3)) << 2), g2));
                const int yi = (int)(lane & 7u), zi = (int)(lane >> 3);
                unsigned int s0, cnt;
                span_of(yi < ny && zi * ny < nrow, y0 + yi, z0 + zi, gy2 + gz2, s0, cnt);
                record(s0, cnt);
            } else
            for (int base = 0; base < rchunk; base += 64) {
                const int rr_i = base + (int)lane;
                const int rrow = rbase + rr_i;
                const int z = z0 + rrow / ny, y = y0 + rrow % ny;
                const float gy = gap_of(uy, y, g.dim[1], g.org[1]), gz = gap_of(uz, z, g.dim[2], g.org[2]);
                unsigned int s0, cnt;
                span_of(rr_i < rchunk, y, z, gy * gy + gz * gz, s0, cnt);
                record(s0, cnt);
            }
            wave_lds_sync();
            if (total <= RAD_FLAT_CAP) {
                // two windows per turn, both loads in flight before the first is used (four measured slower: registers)
                unsigned int before = 0, word = 0;
                for (unsigned int B = 0; B < total; B += 128) {
                    unsigned int my[2], c[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const unsigned int w = (B >> 6) + (unsigned int)u;
                        if ((w & 31u) == 0u) word = endb[(w >> 5) & 3u][lane];
                        const unsigned long long m = __ballot(((word >> (w & 31u)) & 1u) != 0u);
                        my[u] = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, before));
                        before += (unsigned int)__popcll(m);
                        c[u] = B + 64u * (unsigned int)u + lane;
                    }
                    float4 r4[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const unsigned int cc = min(c[u], total - 1u), mm = min(my[u], nspans - 1u);
                        r4[u] = cell_refs[tab_s[mm] + (cc - tab_o[mm])];
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (u == 1 && B + 64u >= total) break;
                        const float d = dist2_nc(qx, qy, qz, r4[u]);
                        PCC_PAIR(c[u] < total);
                        const bool hit = c[u] < total && d < r2;
                        const unsigned long long key = make_key(d, r4[u]);
                        const unsigned long long mask = __ballot(hit);
                        const unsigned int slot = lanes_below(mask, written);
                        if (!COUNT && hit) {
                            if (in_lds) { if (slot < ROW_LDS_MAX) stage[slot] = key; }
                            else if (slot < row_len) keys[row_beg + slot] = key;
                        }
                        written += (unsigned int)__popcll(mask);
                    }
                }
                continue;
            }
            unsigned int next_span = 0, carry_span = 0;
            for (unsigned int B = 0; B < total; B += 64) {
                win[lane] = 0u;
                wave_lds_sync();
                const unsigned int sp = next_span + lane;
                const bool starts = sp < nspans && tab_o[sp] < B + 64;
                if (starts) win[tab_o[sp] - B] = sp + 1;
                next_span += (unsigned int)__popcll(__ballot(starts));
                wave_lds_sync();
                unsigned int v = wave_incl_scan_max(win[lane]);
                v = max(v, carry_span);
                carry_span = (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
                const unsigned int c = B + lane;
                bool hit = false;
                unsigned long long key = 0;
                if (c < total) {
                    const unsigned int my = v - 1;
                    const float4 r4 = cell_refs[tab_s[my] + (c - tab_o[my])];
                    const float d = dist2(qx, qy, qz, r4);
                    hit = d < r2;
                    key = make_key(d, r4);
                }
                const unsigned long long mask = __ballot(hit);
                // (a fill can only find what the count found -- same arithmetic --; the bound is belt and braces)
                const unsigned int slot = lanes_below(mask, written);
                if (!COUNT && hit && slot < (in_lds ? ROW_LDS_MAX : row_len)) row_out[slot] = key;
                written += (unsigned int)__popcll(mask);
            }
        }
        // a row left as keys in memory (beyond the LDS stage) whose offsets promise more than the ball holds: the
        // rest reads "nothing found" (no memset of the key array precedes a fused fill).  No trip with exact offsets.
        if (!COUNT && !in_lds)
            for (unsigned int e = written + lane; e < row_len; e += 64) keys[row_beg + e] = ~0ull;
        if (in_lds) {
            wave_lds_sync();
            const unsigned int have = written < row_len ? written : row_len;
            auto emit = [&](auto rc) {
                constexpr int R = decltype(rc)::value;
                bool in_order = !sorted || have < 2;
                if constexpr (R * 64 <= (int)BUCKET_ROW_MAX) {
                    if (!in_order)
                        in_order = bucket_sort_lds<R>(stage, reinterpret_cast<unsigned long long*>(&tb), reinterpret_cast<unsigned int*>(stage + ROW_LDS_MAX), have, bscale, lane);
                }
                unsigned long long v[R];
#pragma unroll
                for (int rr_ = 0; rr_ < R; ++rr_) v[rr_] = (unsigned int)(rr_ * 64) + lane < have ? stage[rr_ * 64 + lane] : ~0ull;
                if (!in_order) bitonic_sort_regs<R>(v, lane);
#pragma unroll
                for (int rr_ = 0; rr_ < R; ++rr_) {
                    const unsigned int e = (unsigned int)(rr_ * 64) + lane;
                    if (e < row_len) {
                        const bool none = key_none(v[rr_]);
                        if (idx_out) idx_out[row_beg + e] = none ? -1 : (int32_t)(unsigned int)v[rr_];
                        if (d2_out) d2_out[row_beg + e] = none ? __builtin_inff() : __uint_as_float((unsigned int)(v[rr_] >> 32));
                    }
                }
            };
            if (row_len <= 64) emit(std::integral_constant<int, 1>{});
            else if (row_len <= 128) emit(std::integral_constant<int, 2>{});
            else emit(std::integral_constant<int, 4>{});
            wave_lds_sync();
        }
        if (COUNT && lane == 0) counts[qi] = (int32_t)written;
        t += nwaves;
        if (t >= ns) break;
        qi = qi_n;
        qx = __int_as_float(__builtin_amdgcn_readlane((int)hv, 0));
        qy = __int_as_float(__builtin_amdgcn_readlane((int)hv, 1));
        qz = __int_as_float(__builtin_amdgcn_readlane((int)hv, 2));
        row_beg = (int64_t)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)hv, 5) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)hv, 4));
        row_end = (int64_t)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)hv, 7) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)hv, 6));
        qi_n = (unsigned int)__builtin_amdgcn_readlane((int)hv, 8);
    }
}

// min_len: rows shorter than that are someone else's (the fused fill has delivered them).  idx_out / d2_out: the
// rows handled here are also unpacked into the caller's arrays (the long rows of a fused fill).
__global__ void __launch_bounds__(256)
k_sort_rows(const int64_t* __restrict__ offsets, unsigned int nq, unsigned long long* __restrict__ keys, unsigned int min_len,
            int sorted, int32_t* __restrict__ idx_out, float* __restrict__ d2_out, const unsigned int* __restrict__ list,
            const unsigned int* __restrict__ list_count) {
    // list (nullable): the rows to look at (the fused fill names the rows it left as keys: a wave per row just to skip
    // 5M short ones took 0.44 ms)
    const unsigned int lane = threadIdx.x & 63;
    const unsigned int nwaves = (gridDim.x * blockDim.x) >> 6;
    if (list) nq = *list_count;
    for (unsigned int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; j < nq; j += nwaves) {  // wave-uniform
        const unsigned int i = list ? list[j] : j;
        const int64_t beg = offsets[i];
        const unsigned int len = (unsigned int)(offsets[i + 1] - beg);
        unsigned long long* row = keys + beg;
        if (len < min_len) continue;
        if (sorted && len > 1) {
            if (len > ROW_SORT_MAX) sort_long_row(row, len, lane);
            else if (len <= 64) sort_row<1>(row, len, lane);
            else if (len <= 128) sort_row<2>(row, len, lane);
            else if (len <= 256) sort_row<4>(row, len, lane);
            else sort_row<8>(row, len, lane);
        }
        if (idx_out || d2_out) {
            if (sorted && len > ROW_SORT_MAX) __threadfence();
            for (unsigned int e = lane; e < len; e += 64) {
                const unsigned long long k = len > ROW_SORT_MAX ? row_ld(row + e) : row[e];
                const bool none = key_none(k);
                if (idx_out) idx_out[beg + e] = none ? -1 : (int32_t)(unsigned int)k;
                if (d2_out) d2_out[beg + e] = none ? __builtin_inff() : __uint_as_float((unsigned int)(k >> 32));
            }
        }
    }
}

// idx_out / d2_out (fill only, nullable): the caller's result arrays (device).  When the fill takes the wave-per-query
// route it delivers them itself and *delivered is set: the caller skips its unpack pass.
int grid_radius(pcc_index* ix, const float4* q, size_t nq, float r, float r2, int32_t* counts,
                const int64_t* offsets, unsigned long long* keys, int sorted, size_t total, int32_t* idx_out,
                float* d2_out, bool* delivered) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)nq;
    unsigned int *order = nullptr, *n_sorted = nullptr;
    PCC_TRY(grid_sort_queries(ix, q, nq, &order, &n_sorted));
    const GridDev* gd = ix->d_grid.as<GridDev>();
    if (delivered) *delivered = false;
    ev_mark(ix, EV_MAIN0);
    unsigned int gw = (n + 3) / 4;  // 4 waves per workgroup, waves loop
    if (gw > 8192) gw = 8192;
    if (keys && total >= 24 * nq) {  // long rows: a wave per query (short rows leave most of its lanes idle)
        const bool fused = delivered != nullptr && (idx_out || d2_out);
        unsigned int *long_list = nullptr, *long_count = nullptr;
        if (fused) {
            PCC_TRY(ix->knn_fb.reserve((nq + 1) * sizeof(unsigned int)));
            long_list = ix->knn_fb.as<unsigned int>();
            long_count = long_list + nq;
            PCC_HIP(hipMemsetAsync(long_count, 0, sizeof(unsigned int), s));
        }
        hipLaunchKernelGGL((k_grid_radius_fill_wave<false>), dim3(gw), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), gd, q, order, n_sorted, r, r2, offsets, keys,
                           fused ? idx_out : nullptr, fused ? d2_out : nullptr, sorted, long_list, long_count, (int32_t*)nullptr, (const unsigned int*)nullptr);
        PCC_HIP(hipGetLastError());
        if (fused) {  // what is left: rows beyond the LDS stage, still as keys in memory
            hipLaunchKernelGGL(k_sort_rows, dim3(1024), dim3(256), 0, s, offsets, n, keys, ROW_LDS_MAX + 1, sorted, idx_out, d2_out,
                               (const unsigned int*)long_list, (const unsigned int*)long_count);
            PCC_HIP(hipGetLastError());
            *delivered = true;
        } else if (sorted) {
            hipLaunchKernelGGL(k_sort_rows, dim3(gw), dim3(256), 0, s, offsets, n, keys, 0u, 1, (int32_t*)nullptr, (float*)nullptr,
                               (const unsigned int*)nullptr, (const unsigned int*)nullptr);
            PCC_HIP(hipGetLastError());
        }
        ev_mark(ix, EV_MAIN1);
        return PCC_OK;
    }
    const unsigned int* occupied = nullptr;
    if (!keys) {  // count pass, wave form (returns at once unless the balls hold hundreds of candidates)
        PCC_TRY(ix->occ.reserve(sizeof(unsigned int)));
        if (!ix->occ_valid) {  // non-empty cells of this grid: once per indexed cloud
            PCC_HIP(hipMemsetAsync(ix->occ.p, 0, sizeof(unsigned int), s));
            hipLaunchKernelGGL(k_count_occupied, dim3(1024), dim3(256), 0, s, ix->cell_start.as<unsigned int>(), gd, ix->occ.as<unsigned int>());
            PCC_HIP(hipGetLastError());
            ix->occ_valid = true;
        }
        occupied = ix->occ.as<unsigned int>();
        hipLaunchKernelGGL((k_grid_radius_fill_wave<true>), dim3(gw), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), gd, q, order, n_sorted, r, r2, (const int64_t*)nullptr,
                           (unsigned long long*)nullptr, (int32_t*)nullptr, (float*)nullptr, 0, (unsigned int*)nullptr,
                           (unsigned int*)nullptr, counts, occupied);
        PCC_HIP(hipGetLastError());
    }
    if (keys)
        hipLaunchKernelGGL((k_grid_radius<true>), dim3((n + 255) / 256), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), gd, q, order, n_sorted, n, r, r2, counts,
                           offsets, keys, sorted, occupied);
    else
        hipLaunchKernelGGL((k_grid_radius<false>), dim3((n + 255) / 256), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), gd, q, order, n_sorted, n, r, r2, counts,
                           offsets, keys, sorted, occupied);
    PCC_HIP(hipGetLastError());
    if (keys && sorted) {
        hipLaunchKernelGGL(k_sort_rows, dim3(gw), dim3(256), 0, s, offsets, n, keys, 0u, 1, (int32_t*)nullptr, (float*)nullptr,
                           (const unsigned int*)nullptr, (const unsigned int*)nullptr);
        PCC_HIP(hipGetLastError());
    }
    ev_mark(ix, EV_MAIN1);
    return PCC_OK;
}

PCC_PAIRS_TAKE(knn)

}  // namespace pcc
