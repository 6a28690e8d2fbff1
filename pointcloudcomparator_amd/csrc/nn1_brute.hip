// nn1_brute.hip -- exhaustive tiled k=1 nearest neighbour for gfx950 (MI355X).
//
// Replaces N sequential pcl::KdTreeFLANN::nearestKSearch(pt, 1, ...) calls
// (reference src/comparator.cpp:571-577; ICP correspondences reached from :1096).
//
// Shape (wave64, 256-thread workgroups):
//   * Q queries per lane live in VGPRs (x, y, z, running min, winning chunk);
//     a lane owns its queries, so the arg-min needs no cross-lane traffic.
//   * reference points are staged HBM -> registers -> LDS in tiles of 1024
//     float4 (16 KiB, double buffered); all 64 lanes read the same LDS address,
//     which is a conflict-free broadcast ds_read_b128.
//   * inner loop per (query, ref) pair: 3 v_sub, 3 v_mul, 2 v_add (unfused:
//     this file is built with -ffp-contract=off so the bits equal FLANN's
//     L2_Simple), and the running min folded two pairs at a time by v_min3.
//     The index is NOT tracked per pair: only "which 32-point chunk improved
//     the min" (3 VALU per 32 pairs); the winning chunk is re-scanned once at
//     the end to recover the lowest index with d2 == min.
//   * few queries / many references: the reference range is split over
//     blockIdx.y and partial results merge through one 64-bit atomicMin of
//     (d2_bits << 32 | index) -- non-negative floats order like unsigned ints,
//     and the low word gives the lowest-index tie-break for free.
// Bound: fp32 VALU (8.5 non-FMA ops per pair), not HBM, not MFMA -- DESIGN.md.
#include "pcc_internal.hpp"

namespace pcc {

constexpr int BR_T = 256;
constexpr int BR_TILE = 1024;
constexpr int BR_CH = 32;
constexpr int BR_LD = BR_TILE / BR_T;  // float4 loads per thread per tile

__device__ __forceinline__ float dist2(float qx, float qy, float qz, float rx, float ry, float rz) {
    // FLANN L2_Simple: result = 0; result += diff*diff per dimension, each op rounded
    float dx = qx - rx, dy = qy - ry, dz = qz - rz;
    float d = dx * dx;
    d = d + dy * dy;
    d = d + dz * dz;
    return d;
}
__device__ __forceinline__ float dist2(float qx, float qy, float qz, const float4& r) {
    return dist2(qx, qy, qz, r.x, r.y, r.z);
}
__device__ __forceinline__ float fmin3(float a, float b, float c) {
    float r;  // one VALU op for two pairs; operands are never NaN here
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int Q>
__global__ void __launch_bounds__(BR_T)
k_nn1_brute(const float4* __restrict__ refs, unsigned int m, unsigned int per_split, unsigned int splits,
            const float4* __restrict__ q, unsigned int n, unsigned long long* __restrict__ out,
            const unsigned int* __restrict__ qlist, const unsigned int* __restrict__ qcount_dev, int idx_from_w) {
    __shared__ __attribute__((aligned(16))) float tile[2][3][BR_TILE];  // SoA: x[], y[], z[]
    const unsigned int nq = qlist ? min(*qcount_dev, n) : n;
    const unsigned int nqb = (nq + BR_T * Q - 1) / (BR_T * Q);
    const unsigned int total = nqb * splits;
    // work item = (query block, reference split).  Consecutive items share the split, so
    // workgroups running together stream the same reference tiles (L2 reuse per XCD).
    // In list mode the grid is fixed and the item count comes from the device counter.
    for (unsigned int item = blockIdx.x; item < total; item += gridDim.x) {
    const unsigned int qbase = (item % nqb) * (BR_T * Q);
    const unsigned int rbeg = (item / nqb) * per_split;
    const unsigned int rend = min(m, rbeg + per_split);
    if (rbeg >= rend) continue;  // uniform over the workgroup
    __syncthreads();             // previous item's LDS reads are done

    float qx[Q], qy[Q], qz[Q], best[Q];
    int bchunk[Q];
    unsigned int qi[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        unsigned int t = qbase + k * BR_T + threadIdx.x;
        qi[k] = 0xffffffffu;
        qx[k] = qy[k] = qz[k] = 0.f;
        if (t < nq) {
            unsigned int id = qlist ? qlist[t] : t;
            float4 v = q[id];
            if (__float_as_int(v.w) >= 0) { qi[k] = id; qx[k] = v.x; qy[k] = v.y; qz[k] = v.z; }
        }
        best[k] = __builtin_inff();
        bchunk[k] = -1;
    }

    const unsigned int ntiles = (rend - rbeg + BR_TILE - 1) / BR_TILE;
    float sx[BR_LD], sy[BR_LD], sz[BR_LD];
    // tail of the last tile is padded with +inf coordinates: d2 = +inf never wins
#define PCC_STAGE_LOAD(TILE_NO)                                                      \
    _Pragma("unroll") for (int l = 0; l < BR_LD; ++l) {                              \
        unsigned int j = rbeg + (TILE_NO) * BR_TILE + l * BR_T + threadIdx.x;        \
        float4 v = refs[min(j, rend - 1)];                                           \
        bool in = j < rend && __float_as_int(v.w) >= 0; /* non-finite refs never win */ \
        sx[l] = in ? v.x : __builtin_inff();                                         \
        sy[l] = in ? v.y : __builtin_inff();                                         \
        sz[l] = in ? v.z : __builtin_inff();                                         \
    }
    PCC_STAGE_LOAD(0u)
    for (unsigned int t = 0; t < ntiles; ++t) {
        float(*cur)[BR_TILE] = tile[t & 1];
#pragma unroll
        for (int l = 0; l < BR_LD; ++l) {
            cur[0][l * BR_T + threadIdx.x] = sx[l];
            cur[1][l * BR_T + threadIdx.x] = sy[l];
            cur[2][l * BR_T + threadIdx.x] = sz[l];
        }
        __syncthreads();
        if (t + 1 < ntiles) {  // next tile's loads fly under this tile's arithmetic
            PCC_STAGE_LOAD(t + 1)
        }
#pragma unroll 1
        for (int c = 0; c < BR_TILE / BR_CH; ++c) {
            float cm[Q];
#pragma unroll
            for (int k = 0; k < Q; ++k) cm[k] = __builtin_inff();
#pragma unroll
            for (int j = 0; j < BR_CH; j += 4) {
                // wave-uniform addresses: three broadcast ds_read_b128 feed 4 refs x Q queries
                const float4 rx = *reinterpret_cast<const float4*>(&cur[0][c * BR_CH + j]);
                const float4 ry = *reinterpret_cast<const float4*>(&cur[1][c * BR_CH + j]);
                const float4 rz = *reinterpret_cast<const float4*>(&cur[2][c * BR_CH + j]);
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                    float d0 = dist2(qx[k], qy[k], qz[k], rx.x, ry.x, rz.x);
                    float d1 = dist2(qx[k], qy[k], qz[k], rx.y, ry.y, rz.y);
                    float d2_ = dist2(qx[k], qy[k], qz[k], rx.z, ry.z, rz.z);
                    float d3 = dist2(qx[k], qy[k], qz[k], rx.w, ry.w, rz.w);
                    cm[k] = fmin3(cm[k], d0, d1);
                    cm[k] = fmin3(cm[k], d2_, d3);
                }
            }
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                if (cm[k] < best[k]) {  // strict: the earliest chunk holding the min wins
                    best[k] = cm[k];
                    bchunk[k] = (int)(t * (BR_TILE / BR_CH) + c);
                }
            }
        }
    }

    // recover the lowest index with d2 == best inside the winning chunk
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        if (qi[k] == 0xffffffffu) continue;
        unsigned int widx = 0xffffffffu;
        if (bchunk[k] < 0) {
            // every distance overflowed to +inf: the oracle keeps the first valid reference
            for (unsigned int p = rbeg; p < rend; ++p)
                if (__float_as_int(refs[p].w) >= 0) { widx = idx_from_w ? (unsigned int)__float_as_int(refs[p].w) : p; break; }
            if (widx == 0xffffffffu) continue;  // no valid reference in this split
        } else {
            unsigned int base = rbeg + (unsigned int)bchunk[k] * BR_CH;
            for (int j = 0; j < BR_CH; ++j) {
                unsigned int p = base + j;
                if (p >= rend) break;
                float4 r = refs[p];
                if (__float_as_int(r.w) >= 0 && dist2(qx[k], qy[k], qz[k], r) == best[k]) {
                    widx = idx_from_w ? (unsigned int)__float_as_int(r.w) : p;  // position == original index
                    break;
                }
            }
        }
        unsigned long long pk = ((unsigned long long)__float_as_uint(best[k]) << 32) | widx;
        atomicMin(&out[qi[k]], pk);
    }
    }  // item loop
}

template <int Q>
static int launch_q(hipStream_t s, const float4* refs, size_t m, const float4* q, size_t n,
                    unsigned long long* out, const unsigned int* qlist,
                    const unsigned int* qcount_dev, unsigned int splits, unsigned int max_grid, int idx_from_w = 0) {
    unsigned int qblocks = (unsigned int)((n + (size_t)BR_T * Q - 1) / ((size_t)BR_T * Q));
    unsigned int per_split = (unsigned int)((m + splits - 1) / splits);
    per_split = (per_split + BR_TILE - 1) / BR_TILE * BR_TILE;
    splits = (unsigned int)((m + per_split - 1) / per_split);
    unsigned long long total = (unsigned long long)qblocks * splits;
    unsigned int grid = (unsigned int)(total < max_grid ? total : max_grid);
    hipLaunchKernelGGL(k_nn1_brute<Q>, dim3(grid), dim3(BR_T), 0, s, refs, (unsigned int)m, per_split, splits,
                       q, (unsigned int)n, out, qlist, qcount_dev, idx_from_w);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

int launch_nn1_brute(hipStream_t s, const float4* refs, size_t m, const float4* q, size_t n,
                     unsigned long long* out, const unsigned int* qlist,
                     const unsigned int* qcount_dev, size_t qcount_max, bool idx_from_w) {
    size_t nq = qlist ? qcount_max : n;
    if (nq == 0 || m == 0) return PCC_OK;
    if (m >= (1ull << 31) || n >= (1ull << 32)) { set_error("cloud too large for 32-bit indices"); return PCC_ERR_UNSUPPORTED; }
    if (qlist) {
        // GRID fallback list: the count lives on the device, usually tiny.  Fixed grid of
        // 2048 workgroups looping over (query block, split) items; many splits so that a
        // handful of queries still spreads over the whole chip.
        // one tile per split: the list is usually short (the seed scan of an ICP pass: ~10k queries against
        // 31k seeds was 154 items = 0.6 workgroups per CU with four tiles per split, 240 us)
        size_t splits = (m + BR_TILE - 1) / BR_TILE;
        if (splits < 1) splits = 1;
        if (splits > 1024) splits = 1024;
        return launch_q<2>(s, refs, m, q, qcount_max, out, qlist, qcount_dev, (unsigned)splits, 2048, idx_from_w ? 1 : 0);
    }
    // queries per lane: as many as still leave >= 1024 workgroups (4 per CU) in flight
    int Q = 1;
    if (nq >= (size_t)BR_T * 8 * 1024) Q = 8;
    else if (nq >= (size_t)BR_T * 4 * 512) Q = 4;
    else if (nq >= (size_t)BR_T * 2 * 256) Q = 2;
    size_t qblocks = (nq + (size_t)BR_T * Q - 1) / ((size_t)BR_T * Q);
    size_t splits = (2048 + qblocks - 1) / qblocks;
    size_t max_splits = (m + 4 * BR_TILE - 1) / (4 * BR_TILE);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const unsigned int max_grid = 1u << 30;
    switch (Q) {
        case 8: return launch_q<8>(s, refs, m, q, n, out, nullptr, nullptr, (unsigned)splits, max_grid);
        case 4: return launch_q<4>(s, refs, m, q, n, out, nullptr, nullptr, (unsigned)splits, max_grid);
        case 2: return launch_q<2>(s, refs, m, q, n, out, nullptr, nullptr, (unsigned)splits, max_grid);
        default: return launch_q<1>(s, refs, m, q, n, out, nullptr, nullptr, (unsigned)splits, max_grid);
    }
}

}  // namespace pcc
