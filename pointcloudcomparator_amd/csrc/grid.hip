// grid.hip -- GRID engine of libpcc_nn (gfx950): cell-sorted references and exact,
// conservatively bounded ring search.  Plays the role the kd-tree plays inside
// pcl::KdTreeFLANN (reference src/comparator.cpp:564-577, src/segmentation.cpp:120-131):
// prune the exhaustive scan without changing a single result bit.
//
// Index layout in HBM:
//   cell_refs  float4[n_valid]   (x, y, z, bits(position in the packed original-order array)), sorted by linear
//                                cell id with x fastest, so one row of cells along x
//                                is ONE contiguous span of points
//   cell_start uint32[ncells+1]  CSR starts
// Search of one query: scan the cube of cells [c-k, c+k]^3 row by row, keep
// min (d2, index) as one 64-bit key, then bound everything outside the cube from
// below by the distance to the cube's open faces; if the bound cannot exclude a
// closer (or equal, lower-index) point the cube grows, and past KMAX the query is
// handed to the exhaustive kernel.  Distances use the same unfused fp32 arithmetic
// as nn1_brute.hip (-ffp-contract=off), so both engines return identical bits.
#include "pcc_internal.hpp"
#include "grid_device.hpp"
#include <cmath>
#include <cstring>
#include <algorithm>

namespace pcc {

constexpr float GRID_TARGET_PPC = 0.5f; // mean points per cell (over the bounding box) the cell size aims for;
                                         // measured optimum on the corridor scene at 1M and 10M points
constexpr unsigned int GRID_MAX_CELLS = 1u << 26;

// ---- counting sort by cell ----------------------------------------------------------
// The counting pass hands every point its rank inside its cell (the value the atomic
// returns), so the scatter pass needs no second round of atomics: pos = start[cell] + rank.
__global__ void __launch_bounds__(256)
k_cell_count(const float4* __restrict__ p, unsigned int n, GridParams g, unsigned int* __restrict__ count,
             uint2* __restrict__ cell_rank) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 v = p[i];
        if (__float_as_int(v.w) < 0) { cell_rank[i] = make_uint2(0xffffffffu, 0u); continue; }  // invalid query
        unsigned int c = cell_id(v, g);
        cell_rank[i] = make_uint2(c, atomicAdd(&count[c], 1u));
    }
}
__global__ void __launch_bounds__(256)
k_cell_scatter_refs(const float4* __restrict__ p, unsigned int n, const uint2* __restrict__ cell_rank,
                    const unsigned int* __restrict__ start, float4* __restrict__ out) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint2 cr = cell_rank[i];
        float4 v = p[i];
        v.w = __int_as_float((int)i);  // cell-sorted copies carry the packed POSITION, not the original index
        out[start[cr.x] + cr.y] = v;
    }
}
__global__ void __launch_bounds__(256)
k_cell_scatter_ids(unsigned int n, const uint2* __restrict__ cell_rank, const unsigned int* __restrict__ start,
                   unsigned int* __restrict__ order) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint2 cr = cell_rank[i];
        if (cr.x != 0xffffffffu) order[start[cr.x] + cr.y] = i;
    }
}

static inline int grid1d(size_t n) {
    size_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

int grid_build(pcc_index* ix, const float lo_in[3], const float hi_in[3]) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)ix->n_valid;
    const float4* refs = ix->refs.as<float4>();
    float lo[3], hi[3], ext[3];
    float maxext = 0.f, maxabs = 0.f;
    for (int a = 0; a < 3; ++a) {
        lo[a] = lo_in[a];
        hi[a] = hi_in[a];
        ext[a] = hi[a] - lo[a];
        if (!(ext[a] >= 0.f) || !std::isfinite(ext[a])) ext[a] = 0.f;  // overflowed extents fall back to one cell
        maxext = std::max(maxext, ext[a]);
        maxabs = std::max(maxabs, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    }
    // 2. cell size: GRID_TARGET_PPC points per cell on average over the non-flat dimensions
    GridParams g;
    int nd = 0;
    double vol = 1.0;
    for (int a = 0; a < 3; ++a)
        if (ext[a] > 1e-6f * maxext && ext[a] > 0.f) { vol *= ext[a]; ++nd; }
    double cells_wanted = std::max(1.0, (double)n / GRID_TARGET_PPC);
    const char* env = getenv("PCC_GRID_PPC");
    if (env && atof(env) > 0) cells_wanted = std::max(1.0, (double)n / atof(env));
    if (cells_wanted > GRID_MAX_CELLS) cells_wanted = GRID_MAX_CELLS;
    double hcell = nd ? std::pow(vol / cells_wanted, 1.0 / nd) : 1.0;
    if (!(hcell > 0) || !std::isfinite(hcell)) hcell = 1.0;
    for (int iter = 0; iter < 64; ++iter) {  // grow h until the cell count fits
        double tot = 1;
        for (int a = 0; a < 3; ++a) tot *= std::floor(ext[a] / hcell) + 1;
        if (tot <= (double)GRID_MAX_CELLS) break;
        hcell *= 1.26;
    }
    g.h = (float)hcell;
    g.inv_h = 1.0f / g.h;
    if (!std::isfinite(g.inv_h) || g.inv_h <= 0) { g.h = 1.f; g.inv_h = 1.f; }
    double tot = 1;
    for (int a = 0; a < 3; ++a) {
        g.org[a] = lo[a];
        double d = std::floor((double)ext[a] * g.inv_h) + 1;
        if (d > 1 << 20) d = 1 << 20;
        g.dim[a] = (int)d;
        tot *= d;
    }
    if (tot > (double)GRID_MAX_CELLS * 2) { set_error("grid sizing failed"); return PCC_ERR_INVALID; }
    g.ncells = g.dim[0] * g.dim[1] * g.dim[2];
    ix->grid = g;
    ix->stats[3] = (uint64_t)g.ncells;
    // 3. sort by cell
    size_t cs_bytes = ((size_t)g.ncells + 1 + 3) / 4 * 4 * sizeof(unsigned int);
    PCC_TRY(ix->cell_start.reserve(cs_bytes));
    PCC_TRY(ix->cell_refs.reserve((size_t)n * sizeof(float4)));
    static const bool lds_sort = !(getenv("PCC_SORT") && !strcmp(getenv("PCC_SORT"), "atomic"));
    if (lds_sort) {
        PCC_TRY(cell_sort(ix, refs, n, true, ix->cell_refs.as<float4>(), nullptr, ix->cell_start.as<unsigned int>(), nullptr));
        ix->has_grid = true;
        return PCC_OK;
    }
    // first-generation path (one returning device-scope atomic per point), kept for A/B runs
    PCC_TRY(ix->scratch_c.reserve((size_t)n * sizeof(uint2) + 256));
    unsigned int* cstart = ix->cell_start.as<unsigned int>();
    uint2* cell_rank = ix->scratch_c.as<uint2>();
    PCC_HIP(hipMemsetAsync(cstart, 0, cs_bytes, s));
    hipLaunchKernelGGL(k_cell_count, dim3(grid1d(n)), dim3(256), 0, s, refs, n, g, cstart, cell_rank);
    PCC_HIP(hipGetLastError());
    PCC_TRY(launch_exclusive_scan(s, cstart, (size_t)g.ncells + 1, ix->scratch_a));
    hipLaunchKernelGGL(k_cell_scatter_refs, dim3(grid1d(n)), dim3(256), 0, s, refs, n, cell_rank, cstart,
                       ix->cell_refs.as<float4>());
    PCC_HIP(hipGetLastError());
    ix->has_grid = true;
    return PCC_OK;
}

// ---- k = 1 search ------------------------------------------------------------------------
// one candidate folded into the running (d2, index) key
__device__ __forceinline__ unsigned long long fold(unsigned long long best, float qx, float qy, float qz,
                                                   const float4& r) {
    const float d = dist2(qx, qy, qz, r);
    const unsigned long long key =
        ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)__float_as_int(r.w);
    return key < best ? key : best;
}

// scan the contiguous span [s, e) of cell-sorted references.  Four independent 16-byte loads
// are in flight per lane (the loop is latency-bound otherwise); the tail re-reads the last
// point of the span, which cannot change the minimum.
template <int U>
__device__ __forceinline__ unsigned long long scan_span(const float4* __restrict__ cell_refs, unsigned int s,
                                                        unsigned int e, float qx, float qy, float qz,
                                                        unsigned long long best) {
    if (s >= e) return best;
    const unsigned int last = e - 1;
    for (unsigned int p = s; p < e; p += U) {
        float4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = cell_refs[min(p + u, last)];
#pragma unroll
        for (int u = 0; u < U; ++u) best = fold(best, qx, qy, qz, r[u]);
    }
    return best;
}

// scan every row of the cell box; row bounds are fetched four rows at a time (8 independent
// loads) before the spans are streamed, instead of paying two dependent latencies per row
template <int U>
__device__ __forceinline__ unsigned long long scan_box(const float4* __restrict__ cell_refs,
                                                       const unsigned int* __restrict__ cell_start, const GridParams& g,
                                                       int x0, int x1, int y0, int y1, int z0, int z1, float qx, float qy,
                                                       float qz, unsigned long long best) {
    for (int z = z0; z <= z1; ++z) {
        for (int yb = y0; yb <= y1; yb += 4) {
            unsigned int rs[4], re[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = yb + i <= y1;
                const unsigned int row = ((unsigned int)z * g.dim[1] + (ok ? yb + i : yb)) * g.dim[0];
                rs[i] = ok ? cell_start[row + x0] : 0u;
                re[i] = ok ? cell_start[row + x1 + 1] : 0u;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) best = scan_span<U>(cell_refs, rs[i], re[i], qx, qy, qz, best);
        }
    }
    return best;
}

template <int U>
__global__ void __launch_bounds__(256)
k_grid_nn1(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start, GridParams g,
           float slack, const float4* __restrict__ q, const unsigned int* __restrict__ order,
           const unsigned int* __restrict__ n_sorted_ptr, unsigned int n,
           unsigned long long* __restrict__ out, unsigned int* __restrict__ fb_list,
           unsigned int* __restrict__ fb_count) {
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int ns = n_sorted_ptr ? *n_sorted_ptr : n;
    if (t >= ns) return;
    const unsigned int qi = order ? order[t] : t;
    const float4 qv = q[qi];
    if (__float_as_int(qv.w) < 0) return;
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    const int cx = cell_coord(qx, g.org[0], g.inv_h, g.dim[0]);
    const int cy = cell_coord(qy, g.org[1], g.inv_h, g.dim[1]);
    const int cz = cell_coord(qz, g.org[2], g.inv_h, g.dim[2]);
    unsigned long long best = ~0ull;  // (d2 bits << 32) | original index: u64 min == (d2, idx) lexicographic
    bool resolved = false;
    // ---- phase 1: the 3x3x3 cube.  Bounds of all 9 rows first (18 independent loads, one
    // latency), then the rows are streamed.
    {
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
        unsigned int rs[9], re[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int z = cz + i / 3 - 1, y = cy + i % 3 - 1;
            const bool ok = z >= 0 && z < g.dim[2] && y >= 0 && y < g.dim[1];
            const unsigned int row = ((unsigned int)(ok ? z : 0) * g.dim[1] + (ok ? y : 0)) * g.dim[0];
            rs[i] = ok ? cell_start[row + x0] : 0u;
            re[i] = ok ? cell_start[row + x1 + 1] : 0u;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) best = scan_span<U>(cell_refs, rs[i], re[i], qx, qy, qz, best);
        const int y0 = max(cy - 1, 0), y1 = min(cy + 1, g.dim[1] - 1);
        const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.dim[2] - 1);
        const float bd = __uint_as_float((unsigned int)(best >> 32));
        const float lb2 = outside_bound2(qx, qy, qz, x0, x1, y0, y1, z0, z1, g, slack);
        if (best != ~0ull && (bd < lb2 || lb2 == __builtin_inff())) resolved = true;
    }
    if (!resolved) {
        // ---- phase 1b: nothing within the 3x3x3 cube -> double the cube until a point shows up
        bool give_up = false;
        int k = 1;
        while (best == ~0ull) {
            if (k >= GRID_KMAX) { give_up = true; break; }
            k = min(2 * k, GRID_KMAX);
            const int x0 = max(cx - k, 0), x1 = min(cx + k, g.dim[0] - 1);
            const int y0 = max(cy - k, 0), y1 = min(cy + k, g.dim[1] - 1);
            const int z0 = max(cz - k, 0), z1 = min(cz + k, g.dim[2] - 1);
            best = scan_box<U>(cell_refs, cell_start, g, x0, x1, y0, y1, z0, z1, qx, qy, qz, best);
        }
        // ---- phase 2: cover the ball of radius sqrt(best) -- every cell a closer (or equal,
        // lower-index) point could live in.  Usually one extra slab of cells on one or two sides,
        // far fewer rows than the next bigger cube.  Exact by construction: no bound test after it.
        if (!give_up) {
            const float rb = sqrtf(__uint_as_float((unsigned int)(best >> 32))) * 1.00001f + slack;
            int x0, x1, y0, y1, z0, z1;
            cell_range(qx, rb, g.org[0], g.inv_h, g.dim[0], x0, x1);
            cell_range(qy, rb, g.org[1], g.inv_h, g.dim[1], y0, y1);
            cell_range(qz, rb, g.org[2], g.inv_h, g.dim[2], z0, z1);
            const int span = 2 * GRID_KMAX + 1;
            if (!(rb < __builtin_inff()) || x1 - x0 >= span || y1 - y0 >= span || z1 - z0 >= span) {
                give_up = true;  // the ball is too big for a cell walk: exhaustive fallback
            } else {
                best = scan_box<U>(cell_refs, cell_start, g, x0, x1, y0, y1, z0, z1, qx, qy, qz, best);
                resolved = true;
            }
        }
    }
    if (resolved) {
        out[qi] = best;
    } else {
        unsigned int slot = atomicAdd(fb_count, 1u);
        fb_list[slot] = qi;
    }
}

float grid_slack(const GridParams& g) {
    float maxabs = 0.f;
    for (int a = 0; a < 3; ++a)
        maxabs = std::max(maxabs, std::max(std::fabs(g.org[a]), std::fabs(g.org[a] + g.dim[a] * g.h)));
    return 4e-6f * maxabs + 1e-6f * g.h;
}

// sort the queries by reference-grid cell so neighbouring lanes walk the same rows
int grid_sort_queries(pcc_index* ix, const float4* q, size_t nq, unsigned int** order_dev,
                      unsigned int** n_sorted_dev) {
    hipStream_t s = ix->stream;
    const GridParams g = ix->grid;
    const unsigned int n = (unsigned int)nq;
    size_t cs_bytes = ((size_t)g.ncells + 1 + 3) / 4 * 4 * sizeof(unsigned int);
    PCC_TRY(ix->scratch_b.reserve(cs_bytes));
    PCC_TRY(ix->scratch_c.reserve((size_t)n * sizeof(uint2) + 256));
    PCC_TRY(ix->scratch_e.reserve((size_t)n * sizeof(unsigned int) + 256));
    unsigned int* qcell = ix->scratch_b.as<unsigned int>();
    uint2* cell_rank = ix->scratch_c.as<uint2>();
    unsigned int* order = ix->scratch_e.as<unsigned int>();
    unsigned int* n_sorted = ix->small.as<unsigned int>() + 36;
    static const bool lds_sort = !(getenv("PCC_SORT") && !strcmp(getenv("PCC_SORT"), "atomic"));
    if (lds_sort) {
        ev_mark(ix, EV_SORT0);
        PCC_TRY(ix->scratch_g.reserve((size_t)n * sizeof(unsigned int) + 256));
        unsigned int* ord = ix->scratch_g.as<unsigned int>();
        PCC_TRY(cell_sort(ix, q, nq, false, nullptr, ord, nullptr, n_sorted));
        ev_mark(ix, EV_SORT1);
        *order_dev = ord;
        *n_sorted_dev = n_sorted;
        return PCC_OK;
    }
    ev_mark(ix, EV_SORT0);
    PCC_HIP(hipMemsetAsync(qcell, 0, cs_bytes, s));
    hipLaunchKernelGGL(k_cell_count, dim3(grid1d(n)), dim3(256), 0, s, q, n, g, qcell, cell_rank);
    PCC_HIP(hipGetLastError());
    PCC_TRY(launch_exclusive_scan(s, qcell, (size_t)g.ncells + 1, ix->scratch_a));
    // qcell[ncells] == number of valid queries; keep it on the device for the search kernels
    PCC_HIP(hipMemcpyAsync(n_sorted, qcell + g.ncells, 4, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_cell_scatter_ids, dim3(grid1d(n)), dim3(256), 0, s, n, cell_rank, qcell, order);
    PCC_HIP(hipGetLastError());
    ev_mark(ix, EV_SORT1);
    *order_dev = order;
    *n_sorted_dev = n_sorted;
    return PCC_OK;
}

int grid_nn1(pcc_index* ix, const float4* q, size_t nq, unsigned long long* out) {
    hipStream_t s = ix->stream;
    const GridParams g = ix->grid;
    const unsigned int n = (unsigned int)nq;
    PCC_TRY(ix->scratch_d.reserve((size_t)n * sizeof(unsigned int) + 256));
    unsigned int* fb_list = ix->scratch_d.as<unsigned int>();
    unsigned int* fb_count = ix->small.as<unsigned int>() + 32;
    PCC_HIP(hipMemsetAsync(fb_count, 0, 4, s));
    unsigned int *order = nullptr, *n_sorted = nullptr;
    PCC_TRY(grid_sort_queries(ix, q, nq, &order, &n_sorted));
    const float slack = grid_slack(g);
    ev_mark(ix, EV_MAIN0);
    static const int U = getenv("PCC_GRID_UNROLL") ? atoi(getenv("PCC_GRID_UNROLL")) : 4;
    static const int BS = getenv("PCC_GRID_BLOCK") ? atoi(getenv("PCC_GRID_BLOCK")) : 256;
#define PCC_LAUNCH_NN1(UU)                                                                                       \
    hipLaunchKernelGGL((k_grid_nn1<UU>), dim3((n + BS - 1) / BS), dim3(BS), 0, s, ix->cell_refs.as<float4>(),     \
                       ix->cell_start.as<unsigned int>(), g, slack, q, order, n_sorted, n, out, fb_list, fb_count)
    if (U == 1) PCC_LAUNCH_NN1(1);
    else if (U == 2) PCC_LAUNCH_NN1(2);
    else if (U == 8) PCC_LAUNCH_NN1(8);
    else PCC_LAUNCH_NN1(4);
    PCC_HIP(hipGetLastError());
    ev_mark(ix, EV_MAIN1);
    // queries the cubes could not resolve: exhaustive scan over the original-order references
    ev_mark(ix, EV_FB0);
    PCC_TRY(launch_nn1_brute(s, ix->refs.as<float4>(), ix->n_valid, q, n, out, fb_list, fb_count, n));
    ev_mark(ix, EV_FB1);
    return PCC_OK;
}

}  // namespace pcc
