// grid.hip -- GRID engine of libpcc_nn (gfx950): cell-sorted references and exact,
// conservatively bounded ring search.  Plays the role the kd-tree plays inside
// pcl::KdTreeFLANN (reference src/comparator.cpp:564-577, src/segmentation.cpp:120-131):
// prune the exhaustive scan without changing a single result bit.
//
// Index layout in HBM:
//   cell_refs  float4[n_valid]   (x, y, z, bits(position in refs == original index)), sorted by linear
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
#include "lane_ops.hpp"
#include "grid_params_device.hpp"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <type_traits>

namespace pcc {

constexpr float GRID_TARGET_PPC = 0.75f; // mean points per cell (over the bounding box) the cell size aims for (Options::grid_ppc);
                                         // measured optimum on the corridor scene at 1M and 10M points
constexpr unsigned int GRID_MAX_CELLS = 1u << 26;

unsigned int grid_nc_cap(size_t n, double ppc) {
    // cells the grid may use: twice the target (thin clouds round up a lot per dimension)
    double c = 2.0 * (double)n / ppc + 4096.0;
    if (c > (double)GRID_MAX_CELLS) c = (double)GRID_MAX_CELLS;
    return (unsigned int)c;
}

// ---- grid sizing on the device: grid_params_device.hpp; here as a launch of its own (PCC_OPT_FUSE_PARAMS = 0) ------------
__global__ void __launch_bounds__(1024)
k_grid_params(const float* __restrict__ blk, int nblk, unsigned int n, float ppc, unsigned int nc_cap, int trim_k, int axes,
              GridDev* __restrict__ out, GridDev* __restrict__ host_mirror) {
    grid_params_block<false>(blk, nblk, n, ppc, nc_cap, trim_k, axes, out, host_mirror);
}

int grid_params(pcc_index* ix, const float* blk_stats_dev, int n_blocks) {
    const float ppc = (float)ix->opt.grid_ppc;
    const int trim = ix->opt.grid_trim;
    PCC_TRY(ix->d_grid.reserve(sizeof(GridDev)));
    ix->nc_cap = grid_nc_cap(ix->n_orig, ix->opt.grid_ppc);
    // (trimming needs enough rows to tell an outlier from the scene: 128 pack workgroups = 64k points)
    hipLaunchKernelGGL(k_grid_params, dim3(1), dim3(1024), 0, ix->stream, blk_stats_dev, n_blocks, (unsigned int)ix->n_orig,
                       ppc, ix->nc_cap, n_blocks >= 128 ? trim : 0, ix->opt.grid_axes, ix->d_grid.as<GridDev>(), ix->h_grid);
    PCC_HIP(hipGetLastError());
    ix->info_pending = true;
    return PCC_OK;
}

// [52] of pcc_index::small: the ticket word of the fused form (zeroed with the handle, reset by the kernel itself)
int grid_params_fused(pcc_index* ix, PackGrid* pg) {
    PCC_TRY(ix->d_grid.reserve(sizeof(GridDev)));
    ix->nc_cap = grid_nc_cap(ix->n_orig, ix->opt.grid_ppc);
    *pg = PackGrid{ix->small.as<unsigned int>() + 52, (float)ix->opt.grid_ppc, ix->nc_cap, ix->opt.grid_trim, ix->opt.grid_axes,
                   ix->d_grid.as<GridDev>(), ix->h_grid};
    ix->info_pending = true;
    return PCC_OK;
}

int sync_info(pcc_index* ix) {
    if (!ix->info_pending) return PCC_OK;
    PCC_HIP(hipStreamSynchronize(ix->stream));
    ix->info_pending = false;
    ix->grid = ix->h_grid->g;
    ix->n_valid = ix->h_grid->n_valid;
    for (int a = 0; a < 3; ++a) { ix->bbox_lo[a] = ix->h_grid->lo[a]; ix->bbox_hi[a] = ix->h_grid->hi[a]; }
    ix->stats[2] = ix->n_valid;
    ix->stats[3] = (uint64_t)ix->grid.ncells;
    return PCC_OK;
}

constexpr int SEED_STRIDE = PCC_SEED_STRIDE;
constexpr int FAR_SPAN = 1024;

// cell-sort the references (asynchronous; launch sizes come from n_orig and nc_cap)
int grid_build(pcc_index* ix) {
    const size_t n = ix->n_orig;
    size_t cs_bytes = ((size_t)ix->nc_cap + 1 + 3 + 4) / 4 * 4 * sizeof(unsigned int);  // + pad for 16-byte bound loads
    PCC_TRY(ix->cell_start.reserve(cs_bytes));
    PCC_TRY(ix->cell_refs.reserve(n * sizeof(float4) + 64));
    PCC_TRY(cell_sort(ix, ix->refs.as<float4>(), n, true, ix->cell_refs.as<float4>(), nullptr,
                      ix->cell_start.as<unsigned int>(), nullptr));
    ix->has_grid = true;  // (the seed subset for far queries was written by the pack kernel)
    ix->q_cells_n = 0;    // (query cells staged against an earlier grid are void)
    return PCC_OK;
}

// one candidate folded into the running (d2, index) key
__device__ __forceinline__ unsigned long long fold(unsigned long long best, float qx, float qy, float qz,
                                                   const float4& r) {
    const float d = dist2(qx, qy, qz, r);
    const unsigned long long key =
        ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)__float_as_int(r.w);
    return key < best ? key : best;
}

// scan the contiguous span [s, e) of cell-sorted references, U independent 16-byte loads in
// flight per lane.  The kernel is bound by the L1/TA gather rate (a wave64 dwordx4 gather costs
// ~16 clk whatever it fetches), so tail slots are PREDICATED OFF rather than clamped to a
// duplicate address: an inactive lane costs the texture-address unit nothing.
template <int U>
__device__ __forceinline__ unsigned long long scan_span(const float4* __restrict__ cell_refs, unsigned int s,
                                                        unsigned int e, float qx, float qy, float qz,
                                                        unsigned long long best) {
    // one pointer per span, advanced by U; the U loads use immediate offsets and the tail predicate is a
    // compare of the remaining count against a constant: ~4 VALU fewer per candidate slot than indexing
    const float4* ptr = cell_refs + s;
    for (int rem = (int)(e - s); rem > 0; rem -= U, ptr += U) {
        float4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u == 0 || rem > u) r[u] = ptr[u];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u == 0 || rem > u) best = fold(best, qx, qy, qz, r[u]);
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

// Phase 2: the cells the ball of the best distance touches OUTSIDE the cube [cx0..cx1] x [cy0..cy1] x [cz0..cz1] that
// has been dealt with (what the clipped walk of phase 1 skipped inside the cube was farther than the best of that
// moment, hence than today's).  Rows whose y/z gap exceeds the best distance are skipped, the others clipped to
// the ball's chord; a row that runs through the cube contributes only what sticks out on either side.  Bounds of four
// rows are fetched together, as scan_box does.
template <int U>
__device__ __forceinline__ unsigned long long scan_ball_outside(const float4* __restrict__ cell_refs,
                                                                const unsigned int* __restrict__ cell_start,
                                                                const GridParams& g, float slack, int x0, int x1, int y0,
                                                                int y1, int z0, int z1, int cx0, int cx1, int cy0, int cy1,
                                                                int cz0, int cz1, float qx, float qy, float qz,
                                                                float ux, float uy, float uz, unsigned long long best) {
    // (qx, qy, qz): the query, for the distances; (ux, uy, uz): the same point in the grid's frame, for the cells
    for (int z = z0; z <= z1; ++z) {
        const float gz = fmaxf(fmaxf((z == 0 ? -__builtin_inff() : g.org[2] + z * g.h) - uz,
                                     uz - (z == g.dim[2] - 1 ? __builtin_inff() : g.org[2] + (z + 1) * g.h)) - slack, 0.f);
        for (int yb = y0; yb <= y1; yb += 4) {
            unsigned int as[4], ae[4], bs[4], be[4];  // per row: the part left of the cube (or the whole chord), the part right of it
            const float bd = __uint_as_float((unsigned int)(best >> 32));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = yb + i;
                as[i] = ae[i] = bs[i] = be[i] = 0u;
                if (y > y1) continue;
                const float gy = fmaxf(fmaxf((y == 0 ? -__builtin_inff() : g.org[1] + y * g.h) - uy,
                                             uy - (y == g.dim[1] - 1 ? __builtin_inff() : g.org[1] + (y + 1) * g.h)) - slack, 0.f);
                const float rem = bd - (gy * gy + gz * gz) * 0.9999f;
                if (!(rem >= 0.f)) continue;
                int xa, xb;
                cell_range(ux, sqrtf(rem) * 1.00001f + slack, g.org[0], g.inv_h, g.dim[0], xa, xb);
                xa = max(xa, x0);
                xb = min(xb, x1);
                if (xa > xb) continue;
                const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
                if (y >= cy0 && y <= cy1 && z >= cz0 && z <= cz1) {  // through the cube: only what sticks out
                    if (xa < cx0) { as[i] = cell_start[row + xa]; ae[i] = cell_start[row + cx0]; }
                    if (xb > cx1) { bs[i] = cell_start[row + cx1 + 1]; be[i] = cell_start[row + xb + 1]; }
                } else {
                    as[i] = cell_start[row + xa];
                    ae[i] = cell_start[row + xb + 1];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                best = scan_span<U>(cell_refs, as[i], ae[i], qx, qy, qz, best);
                best = scan_span<U>(cell_refs, bs[i], be[i], qx, qy, qz, best);
            }
        }
    }
    return best;
}

// What phase 1 of k_grid_nn1 leaves open, for one query: `best` is its best key over the 3x3x3 cube (~0: empty).
template <int U>
__device__ __forceinline__ void nn1_finish(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
                                           const GridDev* __restrict__ gd, const GridParams& g, float slack, float qx, float qy, float qz, unsigned int qi,
                                           unsigned long long best, unsigned long long* __restrict__ out,
                                           unsigned int* __restrict__ fb_list, unsigned int* __restrict__ fb_count, bool ball,
                                           const float4* __restrict__ warm_refs) {
    // Warm start (ICP passes after the first): out[] still holds every query's neighbour of the previous pass.  A query
    // with nothing in its cube has moved a little since, that reference is most likely still its neighbour or next to
    // it: its distance NOW is an upper bound held by a real point -- no cube doubling, and a query beyond the cell walk
    // takes the bound along to the far walk instead of needing the seed scan.  (Folding it in before phase 1, for every
    // lane, bought nothing there -- at 0.5 references per cell a tight ball still touches most of the cube -- and cost
    // a 16-byte gather per query.)
    if (warm_refs && best == ~0ull) {
        const unsigned long long pk = out[qi];
        if (pk != ~0ull) best = fold(best, qx, qy, qz, warm_refs[(unsigned int)pk]);
    }
    float ux, uy, uz;  // the query in the grid's frame: everything about CELLS below; the distances take (qx, qy, qz)
    grid_frame(g, qx, qy, qz, ux, uy, uz);
    const int cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
    const int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
    const int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
    // ---- phase 1b: nothing within the 3x3x3 cube -> double the cube until a point shows up
    bool give_up = false, done = false;
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
        // (the ball cut down to what the cloud's bounding box leaves of it: grid_device.hpp)
        const BallBox bb = ball_box(ux, uy, uz, __uint_as_float((unsigned int)(best >> 32)), gd, g, slack);
        const int x0 = bb.x0, x1 = bb.x1, y0 = bb.y0, y1 = bb.y1, z0 = bb.z0, z1 = bb.z1;
        const int span = 2 * GRID_KMAX + 1;
        if (!bb.finite || x1 - x0 >= span || y1 - y0 >= span || z1 - z0 >= span) {
            give_up = true;  // the ball is too big for a cell walk: exhaustive fallback
        } else {
            // the cube of half-width k around the query's cell is done (phase 1, or the last doubling)
            if (ball) best = scan_ball_outside<U>(cell_refs, cell_start, g, slack, x0, x1, y0, y1, z0, z1, max(cx - k, 0),
                                        min(cx + k, g.dim[0] - 1), max(cy - k, 0), min(cy + k, g.dim[1] - 1),
                                        max(cz - k, 0), min(cz + k, g.dim[2] - 1), qx, qy, qz, ux, uy, uz, best);
            else best = scan_box<U>(cell_refs, cell_start, g, x0, x1, y0, y1, z0, z1, qx, qy, qz, best);
            done = true;
        }
    }
    if (done) {
        out[qi] = best;
    } else {
        out[qi] = best;  // a real point if there is one (~0 otherwise): the far walk starts from it, the passes merge with atomicMin
        fb_list[atomicAdd(fb_count, 1u)] = qi;
    }
}

template <int U>
__global__ void __launch_bounds__(256)
k_grid_nn1(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
           const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned int* __restrict__ order,
           const unsigned int* __restrict__ n_sorted_ptr, unsigned int n,
           unsigned long long* __restrict__ out, unsigned int* __restrict__ fb_list,
           unsigned int* __restrict__ fb_count, unsigned int xcd_run, bool ball_walk,
           const float4* __restrict__ warm_refs) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Runs of `xcd_run` consecutive
    // groups of sorted queries -- neighbouring rows of cells -- are steered to the SAME XCD, so each L2 sees one
    // eighth of the index instead of all of it (time-neutral here, the Infinity Cache hides the misses, but the
    // fabric traffic drops towards the algorithmic bytes).
    unsigned int bid = blockIdx.x;
    if (xcd_run > 1) {
        const unsigned int per = 8u * xcd_run, full = (gridDim.x / per) * per;
        if (bid < full) {
            const unsigned int base = bid / per * per, in = bid - base;
            bid = base + (in & 7u) * xcd_run + (in >> 3);
        }
    }
    // open queries of the workgroup (phase 1 did not decide them), packed: finished by its first lanes (below)
    __shared__ float4 open_q[256];               // x, y, z, bits(position in q)
    __shared__ unsigned long long open_best[256];
    __shared__ unsigned int open_count;
    if (threadIdx.x == 0) open_count = 0;
    __syncthreads();
    const unsigned int t = bid * blockDim.x + threadIdx.x;
    const unsigned int ns = n_sorted_ptr ? *n_sorted_ptr : n;
    unsigned int qi = 0;
    float4 qv = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    if (t < ns) { qi = order ? order[t] : t; qv = q[qi]; }
    const bool active = __float_as_int(qv.w) >= 0;
    float qx = qv.x, qy = qv.y, qz = qv.z;
    float ux, uy, uz;  // the query in the grid's frame (cells, gaps); distances take (qx, qy, qz)
    grid_frame(g, qx, qy, qz, ux, uy, uz);
    int cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
    int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
    int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
    unsigned long long best = ~0ull;  // (d2 bits << 32) | original index: u64 min == (d2, idx) lexicographic
    bool resolved = false;
    // ---- phase 1: the 3x3x3 cube.  Bounds of all 9 rows first (9 independent 16-byte loads, one latency), then the
    // rows are streamed.  The kernel is VALU-issue bound (PMC: 3000 VALU instructions per wave, the SIMDs 70 % busy),
    // so everything per row is kept to a handful of instructions.
    if (active) {
        const int x0 = max(cx - 1, 0);
        const bool shifted = cx == 0;                // the 16-byte load then starts at the own cell, not at its left neighbour
        const bool has_right = cx + 1 < g.dim[0];
        // per row the four bounds L <= A <= B <= R: [L,A) left cell, [A,B) own cell, [B,R) right cell -- all of them with
        // ONE gather: <= 3 cells, so they sit in one (unaligned) 16-byte load (cell_start is padded by 4 entries)
        auto row_bounds = [&](int i, bool wanted, unsigned int& L, unsigned int& A, unsigned int& B, unsigned int& R) {
            const int z = cz + i / 3 - 1, y = cy + i % 3 - 1;
            const bool ok = wanted && z >= 0 && z < g.dim[2] && y >= 0 && y < g.dim[1];
            const unsigned int row = ((unsigned int)(ok ? z : 0) * g.dim[1] + (ok ? y : 0)) * g.dim[0];
            uint4 b4 = make_uint4(0u, 0u, 0u, 0u);
            if (ok) b4 = *reinterpret_cast<const uint4*>(cell_start + row + x0);
            L = b4.x;
            A = shifted ? b4.x : b4.y;
            B = shifted ? b4.y : b4.z;
            const unsigned int r3 = shifted ? b4.z : b4.w;
            R = has_right ? r3 : B;
        };
        // Own row first; every other row only where the ball of the best distance so far reaches it: rows whose
        // y/z gap already exceeds it are skipped, and the left / right cell of a row is dropped when the gap to the own
        // cell's face plus the row's gap exceeds it (squared gaps against the squared distance: no square root, no cell
        // arithmetic -- ~8 instructions per row).  Gaps are shrunk by the slack and by 1e-4 relative, as the cube bound is;
        // with nothing found yet the best distance is NaN and every comparison below keeps the row whole.
        // All lanes still walk the rows in the same order (lines stay shared); a lane inside a dense region looks
        // at ~60 candidates instead of 230, and so does a lane NEXT to one once its first full row has given it a
        // bound.  C2 140 -> 121 us, C3 1290 -> 1070 us; uniform clouds unchanged.
        const float fx = g.org[0] + cx * g.h, fy = g.org[1] + cy * g.h, fz = g.org[2] + cz * g.h;
        const float gxl = fmaxf((ux - fx) - slack, 0.f), gxr = fmaxf(((fx + g.h) - ux) - slack, 0.f);
        const float gyl = fmaxf((uy - fy) - slack, 0.f), gyr = fmaxf(((fy + g.h) - uy) - slack, 0.f);
        const float gzl = fmaxf((uz - fz) - slack, 0.f), gzr = fmaxf(((fz + g.h) - uz) - slack, 0.f);
        const float gxl2 = gxl * gxl * 0.9999f, gxr2 = gxr * gxr * 0.9999f;
        const float gy2[3] = {gyl * gyl * 0.9999f, 0.f, gyr * gyr * 0.9999f};
        const float gz2[3] = {gzl * gzl * 0.9999f, 0.f, gzr * gzr * 0.9999f};
        // Two rounds of bound gathers: the own row and its four face neighbours, then the four diagonal rows -- by then
        // the best distance is tight and a lane fetches the bounds of a diagonal row only if its ball reaches it.  20
        // bound registers live instead of 36: 8 waves per SIMD instead of 7.
        constexpr int round_rows[2][5] = {{4, 3, 5, 1, 7}, {0, 2, 6, 8, -1}};
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            unsigned int rL[5], rA[5], rB[5], rR[5];
            const float bd0 = __uint_as_float((unsigned int)(best >> 32));
#pragma unroll
            for (int o = 0; o < 5; ++o) {
                const int i = round_rows[rd][o];
                if (i < 0) continue;
                const bool wanted = rd == 0 || !(bd0 - (gy2[i % 3] + gz2[i / 3]) < 0.f);
                row_bounds(i, wanted, rL[o], rA[o], rB[o], rR[o]);
            }
#pragma unroll
            for (int o = 0; o < 5; ++o) {
                const int i = round_rows[rd][o];
                if (i < 0) continue;
                if (rL[o] == rR[o]) continue;
                const float bd = __uint_as_float((unsigned int)(best >> 32));
                const float rem = bd - (gy2[i % 3] + gz2[i / 3]);  // (own row: the gaps are 0; a NaN distance keeps the row whole)
                if (rem < 0.f) continue;  // every point of the row is strictly farther than the best
                const unsigned int s0 = gxl2 > rem ? rA[o] : rL[o], e0 = gxr2 > rem ? rB[o] : rR[o];
                best = scan_span<U>(cell_refs, s0, e0, qx, qy, qz, best);
            }
        }
        const int x1 = min(cx + 1, g.dim[0] - 1);
        const int y0 = max(cy - 1, 0), y1 = min(cy + 1, g.dim[1] - 1);
        const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.dim[2] - 1);
        const float bd = __uint_as_float((unsigned int)(best >> 32));
        const float lb2 = outside_bound2(ux, uy, uz, x0, x1, y0, y1, z0, z1, g, slack);
        if (best != ~0ull && (bd < lb2 || lb2 == __builtin_inff())) resolved = true;
    }
    // The lanes phase 1 leaves open (about one in ten in sparse regions, spread over every wave) used to be finished
    // in place: every wave then walked the ball cover for a handful of lanes -- 37 of the kernel's 120 us on the
    // corridor scene.  They are packed through LDS instead and the workgroup's FIRST lanes finish them: one wave with
    // (nearly) all lanes busy instead of four with a few each.  Still the same launch, so the occupancy that hides the
    // walk's dependent loads is kept (a separate second kernel over a compacted list lost for lack of it).  Worth
    // 3-4 % at 10M x 10M (1070 -> 1030 us), nothing at 1M x 1M: the walk's cost is the LINES its lanes touch, which
    // packing does not change.  Letting the last wave to arrive finish them instead of a barrier lost (143 us at 1M).
    if (active && resolved) out[qi] = best;
    if (active && !resolved) {
        const unsigned int slot = atomicAdd(&open_count, 1u);
        open_q[slot] = make_float4(qx, qy, qz, __uint_as_float(qi));
        open_best[slot] = best;
    }
    __syncthreads();
    const unsigned int n_open = open_count;
    for (unsigned int j = threadIdx.x; j < n_open; j += blockDim.x) {
        const float4 oq = open_q[j];
        qx = oq.x; qy = oq.y; qz = oq.z;
        qi = __float_as_uint(oq.w);
        best = open_best[j];
        nn1_finish<2>(cell_refs, cell_start, gd, g, slack, qx, qy, qz, qi, best, out, fb_list, fb_count, ball_walk, warm_refs);
    }
}

// ---- k_grid_nn1_flat2: the rows drained with lanes over CANDIDATES ---------------------------------------------
// k_grid_nn1 gives every query a lane, and a wave's walk over a row of cells lasts as long as the LONGEST of its 64
// spans: 20 of 64 lanes active per VALU instruction at 10M x 10M (profiles/r02_nn1_counters.json) -- a row of three
// cells holds 1.5 references on average in sparse regions, and in dense ones the rows are clipped to the ball of the
// best distance, so most lanes skip most rows while the wave still walks all nine at full length.
// Here the spans of a pass (the same rows, bounds, clipping and slack as k_grid_nn1: the candidate set covers every
// reference that kernel looks at, and a reference more never changes a minimum) are DRAINED FLAT: laid end to end, and
// the wave takes the candidates 64 at a time across span boundaries.  Consecutive lanes read consecutive cell_refs
// entries, fetch their query from LDS and fold (d2 bits, index) into the query's LDS slot with one ds_min_u64 -- order
// free, hence exact.  Which span a candidate belongs to: one bit per span END over the flat index space; the rank of a
// candidate's span = ends before it = ends of earlier windows (a scalar count) + v_mbcnt of its window's bits.  The bits
// are stored TRANSPOSED -- word (p mod 64) of plane (p / 2048) holds flat position p at bit (p / 64) mod 32 -- so the
// ends of neighbouring spans, a few positions apart, go to different words (the atomic ORs that set them do not
// collide: with one word per 64 positions half of the LDS cycles were conflicts), lane l's word serves 32 consecutive
// windows, and a window's mask is a ballot of one bit test.
// (The ablation builds behind profiles/r05_nn1_ablation.txt -- parts of this kernel compiled out to price them -- are a patch
// applied to a scratch copy of this file: tools/exp_ablate.sh + tools/exp_ablate.patch.  None of it lives here.)
constexpr int FLAT_PLANES = 4;
constexpr int FLAT_CAP = FLAT_PLANES * 2048;  // candidates one flat pass can hold; larger passes fall back to the lane walk

__device__ __forceinline__ void flat_sync() { wave_lds_sync(); }  // (lane_ops.hpp)

// Two passes.  The spans of a pass are laid out lane by lane (a lane's rows next to each other), so ONE wave scan
// gives every lane its flat offset and its first record slot
// (candidate total and non-empty-span count travel in one 32-bit word); records are one word, (first reference - flat
// offset) and the query slot packed.  Pass 0 gives every query a bound: the own row where the wave is sparse, the own CELL
// where it is dense (>= 4 references per cell on average over the wave: there the nearest neighbour is almost always in
// the own cell, and the other two cells of the row cost 17 candidates per query).  Pass 1 takes everything else -- the
// rest of the own row and the eight neighbour rows, each skipped or clipped against the bound exactly as in k_grid_nn1.
constexpr int F2_R = 10;
template <int SR>
struct alignas(16) FlatWaveT {
    float4 q[64];                        // the wave's queries (x, y, z, bits of the best d2 when the pass began)
    unsigned long long best[64];         // running (d2 bits << 32 | index) per query
    unsigned int ends[FLAT_PLANES][64];  // span-end bits, transposed (k_grid_nn1_flat)
    unsigned int span[SR * 64];          // non-empty spans in flat order: (first reference - flat offset + FLAT_CAP) << 6 | query slot
};
using FlatWave2 = FlatWaveT<F2_R>;
static_assert(sizeof(FlatWave2) == 1024 + 512 + FLAT_PLANES * 256 + F2_R * 256, "LDS budget of k_grid_nn1_flat2");
constexpr size_t F2_MAX_REFS = (1u << 26) - 2 * FLAT_CAP;  // the packed record holds 26 bits of reference position

template <int R, int U, int B, bool LIVE, class FW>
__device__ __forceinline__ void flat2_pass(FW& fw, const float4* __restrict__ cell_refs, const unsigned int (&s)[R],
                                           const unsigned int (&len)[R], unsigned int lane) {
    static_assert(B == 1 || B == 2 || B == 4 || B == 8, "a batch of windows never straddles a plane of 32");
    unsigned int tot = 0, cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) { tot += len[r]; cnt += len[r] != 0 ? 1u : 0u; }
    // (a lane total past 16383 means the pass is past FLAT_CAP anyway; clamped so that the wave sum stays below 2^20)
    const unsigned int v = min(tot, 16383u) | (cnt << 20);
    const unsigned int incl = wave_incl_scan_add(v);
    const unsigned int last = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
    const unsigned int T = last & 0xfffffu, nspan = last >> 20;  // wave-uniform
    if (T == 0) return;
    if (T > (unsigned int)FLAT_CAP) {  // (piles of duplicates, very coarse grids: rare) the lane walk of k_grid_nn1
        unsigned long long b = ~0ull;
        const float4 qv = fw.q[lane];  // (the lane's own query: not kept in registers across the passes for this rare case)
#pragma unroll
        for (int r = 0; r < R; ++r) b = scan_span<U>(cell_refs, s[r], s[r] + len[r], qv.x, qv.y, qv.z, b);
        if (b < fw.best[lane]) fw.best[lane] = b;  // (only this lane touches its slot outside a flat drain)
        flat_sync();
        return;
    }
    unsigned int off = (incl - v) & 0xfffffu, rank = (incl - v) >> 20;
    const unsigned int nwin = (T + 63) >> 6;
#pragma unroll
    for (int p = 0; p < FLAT_PLANES; ++p)
        if ((unsigned int)p * 2048u < T) fw.ends[p][lane] = 0u;
    flat_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (len[r]) {
            fw.span[rank] = ((s[r] - off + (unsigned int)FLAT_CAP) << 6) | lane;
            const unsigned int e = off + len[r] - 1;  // last flat position of the span
            // (the end of the very last span is marked too: only lanes past the end could count it, and they are masked)
            atomicOr(&fw.ends[e >> 11][e & 63], 1u << ((e >> 6) & 31));
            off += len[r];
            ++rank;
        }
    }
    flat_sync();
    const unsigned int last_span = nspan - 1;
    unsigned int before = 0;  // span ends in earlier windows (scalar)
    unsigned int word = 0;
    auto batch = [&](unsigned int w0, auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        unsigned int rec[B], cc[B];
        bool ok[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const unsigned int w = w0 + b;
            const unsigned long long m = __ballot(((word >> (w & 31)) & 1u) != 0u);
            unsigned int rk = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, before));
            before += (unsigned int)__popcll(m);
            cc[b] = (w << 6) + lane;
            ok[b] = true;
            if (TAIL) {
                ok[b] = cc[b] < T;
                cc[b] = ok[b] ? cc[b] : T - 1;
                rk = ok[b] ? rk : last_span;
            }
            rec[b] = fw.span[rk];
        }
        float4 r4[B];
#pragma unroll
        for (int b = 0; b < B; ++b) r4[b] = cell_refs[(rec[b] >> 6) + cc[b] - (unsigned int)FLAT_CAP];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const unsigned int slot = rec[b] & 63u;
            const float4 qv = fw.q[slot];
            const float d = dist2_nc(qv.x, qv.y, qv.z, r4[b]);
            PCC_PAIR(ok[b]);
            const unsigned int db = __float_as_uint(d);
            const unsigned int cur = LIVE ? reinterpret_cast<const unsigned int*>(&fw.best[slot])[1] : __float_as_uint(qv.w);
            const unsigned int ri = (unsigned int)__float_as_int(r4[b].w);  // (never ~0: keeps the candidate ONE 16-byte load)
            if (ok[b] && db <= cur && ri != 0xffffffffu) atomicMin(&fw.best[slot], ((unsigned long long)db << 32) | ri);
        }
    };
    const unsigned int nfull = (T >> 6) / B * B;
    for (unsigned int p0 = 0; p0 < nwin; p0 += 32) {
        word = fw.ends[p0 >> 5][lane];
        const unsigned int pend = min(p0 + 32u, nwin);
        unsigned int w0 = p0;
        for (; w0 + B <= min(pend, nfull); w0 += B) batch(w0, std::false_type{});
        for (; w0 < pend; w0 += B) batch(w0, std::true_type{});
    }
    flat_sync();
}

template <int U, int B, int NW, bool OPENK>
__global__ void __launch_bounds__(NW * 64)
k_grid_nn1_flat2(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
                 const GridDev* __restrict__ gd, const float4* q, const unsigned int* __restrict__ order,
                 const unsigned int* __restrict__ n_sorted_ptr, unsigned int n,
                 unsigned long long* __restrict__ out, unsigned int* __restrict__ fb_list,
                 unsigned int* __restrict__ fb_count, unsigned int xcd_run, bool ball_walk,
                 const float4* __restrict__ warm_refs, unsigned int dense_min,
                 unsigned int* __restrict__ open_list, unsigned long long* __restrict__ open_keys,
                 unsigned int* __restrict__ open_total, const float* __restrict__ pre_T, float4* q_rw) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    unsigned int bid = blockIdx.x;  // XCD-aware order of the workgroups, as in k_grid_nn1
    if (xcd_run > 1) {
        const unsigned int per = 8u * xcd_run, full = (gridDim.x / per) * per;
        if (bid < full) {
            const unsigned int base = bid / per * per, in = bid - base;
            bid = base + (in & 7u) * xcd_run + (in >> 3);
        }
    }
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[NW * sizeof(FlatWave2)];
    __shared__ unsigned int open_count;
    static_assert(sizeof(FlatWave2) >= 64 * (sizeof(float4) + sizeof(unsigned long long)), "open-lane list fits");
    FlatWave2& fw = reinterpret_cast<FlatWave2*>(lds_raw)[threadIdx.x >> 6];
    float4* open_q = reinterpret_cast<float4*>(lds_raw);
    unsigned long long* open_best = reinterpret_cast<unsigned long long*>(lds_raw + NW * 64 * sizeof(float4));
    if (threadIdx.x == 0) open_count = 0;
    const unsigned int lane = threadIdx.x & 63;
    const unsigned int t = bid * blockDim.x + threadIdx.x;
    const unsigned int ns = n_sorted_ptr ? *n_sorted_ptr : n;
    unsigned int qi = 0;
    float4 qv = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    if (t < ns) {
        qi = order ? order[t] : t;
        qv = q[qi];
    }
    const bool active = __float_as_int(qv.w) >= 0;
    // ICP passes in cell order (pcc_index::pre_transform): the previous pass's rigid motion is applied HERE, to the query the lane
    // has just read, and written back for the kernels that follow (open lanes, far walk, sums) -- pcl::transformPointCloud's
    // rounding, ((m0 x + m1 y) + m2 z) + m3, exactly as k_transform does it; the pass then has no transform kernel of its own
    if (pre_T && active) {
        const float x = qv.x, y = qv.y, z = qv.z;
        qv.x = ((pre_T[0] * x + pre_T[1] * y) + pre_T[2] * z) + pre_T[3];
        qv.y = ((pre_T[4] * x + pre_T[5] * y) + pre_T[6] * z) + pre_T[7];
        qv.z = ((pre_T[8] * x + pre_T[9] * y) + pre_T[10] * z) + pre_T[11];
        q_rw[qi] = qv;
    }
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    float ux, uy, uz;  // the query in the grid's frame (cells, gaps); the drain's distances take fw.q = (qx, qy, qz)
    grid_frame(g, qx, qy, qz, ux, uy, uz);
    const int cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
    const int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
    const int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
    fw.q[lane] = make_float4(qx, qy, qz, __uint_as_float(0xffffffffu));
    fw.best[lane] = ~0ull;
    flat_sync();
    unsigned long long best = ~0ull;
    bool resolved = false;
    {
        const int x0 = max(cx - 1, 0);
        const bool shifted = cx == 0;
        const bool has_right = cx + 1 < g.dim[0];
        const float fx = g.org[0] + cx * g.h, fy = g.org[1] + cy * g.h, fz = g.org[2] + cz * g.h;
        const float gxl = fmaxf((ux - fx) - slack, 0.f), gxr = fmaxf(((fx + g.h) - ux) - slack, 0.f);
        const float gyl = fmaxf((uy - fy) - slack, 0.f), gyr = fmaxf(((fy + g.h) - uy) - slack, 0.f);
        const float gzl = fmaxf((uz - fz) - slack, 0.f), gzr = fmaxf(((fz + g.h) - uz) - slack, 0.f);
        const float gxl2 = gxl * gxl * 0.9999f, gxr2 = gxr * gxr * 0.9999f;
        const float gy2[3] = {gyl * gyl * 0.9999f, 0.f, gyr * gyr * 0.9999f};
        const float gz2[3] = {gzl * gzl * 0.9999f, 0.f, gzr * gzr * 0.9999f};
        // the four bounds L <= A <= Bc <= Rr of row i (0..8: z = cz + i / 3 - 1, y = cy + i % 3 - 1): [L, A) left cell,
        // [A, Bc) own column's cell, [Bc, Rr) right cell -- one unaligned 16-byte gather, as in k_grid_nn1
        const unsigned int rowc = ((unsigned int)cz * g.dim[1] + cy) * g.dim[0] + x0;
        const bool ok_y[3] = {cy > 0, true, cy + 1 < g.dim[1]}, ok_z[3] = {cz > 0, true, cz + 1 < g.dim[2]};
        // (two steps, so that the gathers of all rows of a pass are in flight together: the loads first, nothing else
        // inside their predicate; the selects afterwards)
        auto row_load = [&](int i, bool wanted) -> uint4 {
            uint4 b4 = make_uint4(0u, 0u, 0u, 0u);
            if (wanted & active & ok_y[i % 3] & ok_z[i / 3]) {  // (one predicate, one branch)
                const int delta = ((i / 3 - 1) * g.dim[1] + (i % 3 - 1)) * g.dim[0];
                b4 = *reinterpret_cast<const uint4*>(cell_start + (rowc + (unsigned int)delta));
            }
            return b4;
        };
        auto row_bounds = [&](const uint4& b4, unsigned int& L, unsigned int& A, unsigned int& Bc, unsigned int& Rr) {
            L = b4.x;
            A = shifted ? b4.x : b4.y;
            Bc = shifted ? b4.y : b4.z;
            const unsigned int r3 = shifted ? b4.z : b4.w;
            Rr = has_right ? r3 : Bc;
        };
        // the bound of everything OUTSIDE the 3x3x3 cube, taken now: it depends on the query alone, and the grid-frame
        // coordinates die here instead of living through both passes
        float lb2;
        {
            const int x1 = min(cx + 1, g.dim[0] - 1);
            const int y0 = max(cy - 1, 0), y1 = min(cy + 1, g.dim[1] - 1);
            const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.dim[2] - 1);
            lb2 = outside_bound2(ux, uy, uz, x0, x1, y0, y1, z0, z1, g, slack);
            asm volatile("" : "+v"(lb2));  // (HERE: left alone the compiler sinks the whole bound to its use behind the passes, with its six inputs)
        }
        unsigned int oL, oA, oB, oR;
        row_bounds(row_load(4, true), oL, oA, oB, oR);
        // dense wave: >= dense_min references per own cell on average
        const unsigned int own_cells = (unsigned int)__builtin_amdgcn_readlane((int)wave_incl_scan_add(oB - oA), 63);
        const bool dense = own_cells >= 64u * dense_min;
        {
            unsigned int s1[1] = {dense ? oA : oL}, l1[1] = {dense ? oB - oA : oR - oL};
            flat2_pass<1, U, B, true>(fw, cell_refs, s1, l1, lane);
        }
        {
            const float bd = __uint_as_float((unsigned int)(fw.best[lane] >> 32));  // (NaN while nothing is found: every test below keeps its span)
            fw.q[lane].w = bd;  // the flat drain's pre-filter (read with the query, no extra LDS access)
            unsigned int sp[F2_R], ln[F2_R];
            // the rest of the own row (dense waves): a side cell is dropped when the gap to it exceeds the bound
            sp[0] = oL; ln[0] = dense && !(gxl2 > bd) ? oA - oL : 0u;
            sp[1] = oB; ln[1] = dense && !(gxr2 > bd) ? oR - oB : 0u;
            constexpr int rows[8] = {3, 5, 1, 7, 0, 2, 6, 8};  // face neighbours, then the diagonal rows
            uint4 b4[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const int i = rows[o];
                b4[o] = row_load(i, !(bd - (gy2[i % 3] + gz2[i / 3]) < 0.f));
            }
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const int i = rows[o];
                const float rem = bd - (gy2[i % 3] + gz2[i / 3]);
                unsigned int L, A, Bc, Rr;
                row_bounds(b4[o], L, A, Bc, Rr);
                const unsigned int s0 = gxl2 > rem ? A : L, e0 = gxr2 > rem ? Bc : Rr;
                sp[2 + o] = s0;
                ln[2 + o] = e0 - s0;
            }
            flat2_pass<F2_R, U, B, false>(fw, cell_refs, sp, ln, lane);
        }
        best = fw.best[lane];
        if (active) {
            const float bd = __uint_as_float((unsigned int)(best >> 32));
            if (best != ~0ull && (bd < lb2 || lb2 == __builtin_inff())) resolved = true;
        }
    }
    if (OPENK) {
        // the lanes phase 1 leaves open go to a device-wide list, finished by k_nn1_open with every lane busy (in place,
        // a workgroup's dozen open lanes ran the ball walk at a fifth of the lanes: 45 % of the kernel's instructions)
        // (a NON-TEMPORAL store: the key is scattered through the sort order -- 64 sectors per wave, none of them written again
        // by this kernel -- and a plain store leaves 10M partly written lines to age in the L2s beside the rows the search
        // re-reads.  Round 5, same box, two runs each: 696 / 691 -> 668 / 671 us.  The scatter stays the most expensive single
        // thing the kernel does: with the key stored at out[t] instead the kernels take 447 us, profiles/r05_nn1_ablation.txt)
        if (active && resolved) __builtin_nontemporal_store(best, &out[qi]);
        const bool open = active && !resolved;
        const unsigned long long om = __ballot(open);
        if (om != 0ull) {
            // (one returning atomic per wave; the counters are sharded by workgroup so that no single word has to take them
            // all: every shard owns the slice of the list its workgroups could fill)
            const unsigned int shard = blockIdx.x % PCC_OPEN_SHARDS;
            const unsigned int cap = (gridDim.x + PCC_OPEN_SHARDS - 1) / PCC_OPEN_SHARDS * (NW * 64);
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(open_total + shard * PCC_OPEN_CTR_STRIDE, (unsigned int)__popcll(om));
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            if (open) {
                const unsigned int k = shard * cap + base + __builtin_amdgcn_mbcnt_hi((unsigned int)(om >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)om, 0u));
                open_list[k] = qi;
                open_keys[k] = best;
            }
        }
        return;
    }
    __syncthreads();  // every wave is done with its block: the open-lane list may overwrite them
    if (active && resolved) out[qi] = best;
    if (active && !resolved) {
        const unsigned int slot = atomicAdd(&open_count, 1u);
        open_q[slot] = make_float4(qv.x, qv.y, qv.z, __uint_as_float(qi));
        open_best[slot] = best;
    }
    __syncthreads();
    const unsigned int n_open = open_count;
    for (unsigned int j = threadIdx.x; j < n_open; j += blockDim.x) {
        const float4 oq = open_q[j];
        nn1_finish<2>(cell_refs, cell_start, gd, g, slack, oq.x, oq.y, oq.z, __float_as_uint(oq.w), open_best[j], out, fb_list,
                      fb_count, ball_walk, warm_refs);
    }
}

// the open lanes of k_grid_nn1_flat2, compacted: one lane per listed query, every lane busy
__global__ void __launch_bounds__(256)
k_nn1_open(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start, const GridDev* __restrict__ gd,
           const float4* __restrict__ q, const unsigned int* __restrict__ open_list,
           const unsigned long long* __restrict__ open_keys, const unsigned int* __restrict__ open_total,
           unsigned long long* __restrict__ out, unsigned int* __restrict__ fb_list, unsigned int* __restrict__ fb_count,
           bool ball_walk, const float4* __restrict__ warm_refs, unsigned int shard_cap) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    // block b works on shard b % PCC_OPEN_SHARDS, as chunk b / PCC_OPEN_SHARDS of gridDim / PCC_OPEN_SHARDS
    const unsigned int shard = blockIdx.x % PCC_OPEN_SHARDS, chunk = blockIdx.x / PCC_OPEN_SHARDS, nchunk = gridDim.x / PCC_OPEN_SHARDS;
    const unsigned int cnt = open_total[shard * PCC_OPEN_CTR_STRIDE];
    for (unsigned int j = chunk * blockDim.x + threadIdx.x; j < cnt; j += nchunk * blockDim.x) {
        const unsigned int k = shard * shard_cap + j;
        const unsigned int qi = open_list[k];
        const float4 qv = q[qi];
        nn1_finish<2>(cell_refs, cell_start, gd, g, slack, qv.x, qv.y, qv.z, qi, open_keys[k], out, fb_list, fb_count, ball_walk,
                      warm_refs);
    }
}

// distance from coordinate v to the interval of cell c along one axis (0 inside), shrunk by slack
// (the boundary cells are open-ended: they also hold the references that lie beyond the grid's box)
__device__ __forceinline__ float axis_gap(float v, int c, int dim, float org, float h, float slack) {
    const float lo = c == 0 ? -__builtin_inff() : org + c * h, hi = c == dim - 1 ? __builtin_inff() : org + (c + 1) * h;
    return fmaxf(fmaxf(lo - v, v - hi) - slack, 0.f);
}

// The open lanes again, DRAINED FLAT (round 4).  k_nn1_open gives every listed query a lane that walks the box of its
// ball row by row: two dependent round trips per row and a wave as slow as its widest box -- 26 of 64 lanes active, and in
// the ICP configuration (2M x 2M) 150 us a pass against 91 us for the search proper.  Here the lanes only LAY OUT the
// rows: the ball's rows are taken OR at a time (one z-layer of cells, OR rows of it), every lane clips its rows to the
// chord of the ball of its CURRENT best distance (same gaps, slack and factors as scan_ball_outside) and the spans of all
// 64 queries are drained with lanes over candidates by flat2_pass -- the bounds of a layer's rows in flight together, the
// candidates coalesced, the bound shrinking from layer to layer.  Exact for the same reason the ball walk is: every cell
// that can hold a reference at or below the best distance is looked at.  Queries without any bound (empty cube, no warm
// start) or with a ball wider than the cell walk allows keep the lane walk (nn1_finish), which also feeds the far list.
constexpr int OPEN_NY = 4, OPEN_NZ = 4, OPEN_R = OPEN_NY * OPEN_NZ;  // rows of one flat pass: a 4 x 4 window of the ball's box
using FlatWaveOpen = FlatWaveT<OPEN_R>;
template <int U, int B, int NW>
__global__ void __launch_bounds__(NW * 64)
k_nn1_open_flat(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start, const GridDev* __restrict__ gd,
                const float4* __restrict__ q, const unsigned int* __restrict__ open_list,
                const unsigned long long* __restrict__ open_keys, const unsigned int* __restrict__ open_total,
                unsigned long long* __restrict__ out, unsigned int* __restrict__ fb_list, unsigned int* __restrict__ fb_count,
                bool ball_walk, const float4* __restrict__ warm_refs, unsigned int shard_cap) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[NW * sizeof(FlatWaveOpen)];
    FlatWaveOpen& fw = reinterpret_cast<FlatWaveOpen*>(lds_raw)[threadIdx.x >> 6];
    const unsigned int lane = threadIdx.x & 63;
    const unsigned int shard = blockIdx.x % PCC_OPEN_SHARDS, chunk = blockIdx.x / PCC_OPEN_SHARDS, nchunk = gridDim.x / PCC_OPEN_SHARDS;
    const unsigned int cnt = open_total[shard * PCC_OPEN_CTR_STRIDE];
    for (unsigned int j0 = chunk * blockDim.x; j0 < cnt; j0 += nchunk * blockDim.x) {  // (block-uniform)
        const unsigned int j = j0 + threadIdx.x;
        const bool valid = j < cnt;
        const unsigned int k = shard * shard_cap + (valid ? j : 0u);
        const unsigned int qi = valid ? open_list[k] : 0u;
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned long long key = ~0ull;
        if (valid) { qv = q[qi]; key = open_keys[k]; }
        const float qx = qv.x, qy = qv.y, qz = qv.z;
        if (valid && warm_refs && key == ~0ull) {  // (ICP passes after the first: the previous neighbour bounds the ball)
            const unsigned long long pk = out[qi];
            if (pk != ~0ull) key = fold(key, qx, qy, qz, warm_refs[(unsigned int)pk]);
        }
        float ux, uy, uz;  // the query in the grid's frame (cells, gaps, chords); distances take (qx, qy, qz)
        grid_frame(g, qx, qy, qz, ux, uy, uz);
        int x0 = 0, x1 = -1, y0 = 0, y1 = -1, z0 = 0, z1 = -1;
        float ex2 = 0.f;
        bool flat = valid && key != ~0ull;
        if (flat) {
            const BallBox bb = ball_box(ux, uy, uz, __uint_as_float((unsigned int)(key >> 32)), gd, g, slack);
            x0 = bb.x0; x1 = bb.x1; y0 = bb.y0; y1 = bb.y1; z0 = bb.z0; z1 = bb.z1;
            ex2 = bb.ex2;
            const int span = 2 * GRID_KMAX + 1;  // (beyond it the lane walk hands the query to the far list)
            flat = bb.finite && x1 - x0 < span && y1 - y0 < span && z1 - z0 < span;
        }
        const int cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
        const int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
        const int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
        const bool cube_done = true;  // (every listed query comes from k_grid_nn1_flat2, which has scanned the cube around its cell)
        const int ny = flat ? y1 - y0 + 1 : 0, nz = flat ? z1 - z0 + 1 : 0;
        const int max_ny = __builtin_amdgcn_readlane((int)wave_incl_scan_max((unsigned int)ny), 63);
        const int max_nz = __builtin_amdgcn_readlane((int)wave_incl_scan_max((unsigned int)nz), 63);
        fw.q[lane] = make_float4(qx, qy, qz, 0.f);
        fw.best[lane] = key;
        flat_sync();
        // One pass takes a 4 x 4 window of rows -- the whole box of a ball up to two cells in radius, i.e. of nearly every
        // open query of an aligned cloud -- so that a wave's chain is two round trips (bounds, candidates) whatever the ball:
        // with one z-layer per pass the kernel waited for ten (4.8 waves per SIMD, the VALUs 27 % busy: 152 us at 10M x 10M
        // against 103 us for the lane walk).  Wider boxes take further windows.
        for (int zb = 0; zb < max_nz; zb += OPEN_NZ) {  // wave-uniform loops
            for (int yb = 0; yb < max_ny; yb += OPEN_NY) {
                const float bd = __uint_as_float(reinterpret_cast<const unsigned int*>(&fw.best[lane])[1]);
                float gy2[OPEN_NY], gz2[OPEN_NZ];
#pragma unroll
                for (int a = 0; a < OPEN_NY; ++a) {
                    const float gy = axis_gap(uy, min(y0 + yb + a, g.dim[1] - 1), g.dim[1], g.org[1], g.h, slack);
                    gy2[a] = yb + a < ny ? gy * gy : __builtin_inff();  // (+inf: a row the lane does not have)
                }
#pragma unroll
                for (int a = 0; a < OPEN_NZ; ++a) {
                    const float gz = axis_gap(uz, min(z0 + zb + a, g.dim[2] - 1), g.dim[2], g.org[2], g.h, slack);
                    gz2[a] = zb + a < nz ? gz * gz : __builtin_inff();
                }
                bool iny[OPEN_NY], inz[OPEN_NZ];  // the row runs through the cube around the query's cell
#pragma unroll
                for (int a = 0; a < OPEN_NY; ++a) iny[a] = cube_done && abs(y0 + yb + a - cy) <= 1;
#pragma unroll
                for (int a = 0; a < OPEN_NZ; ++a) inz[a] = abs(z0 + zb + a - cz) <= 1;
                unsigned int sp[OPEN_R], ln[OPEN_R], ea[OPEN_R];
                int xa[OPEN_R], xb[OPEN_R];
                bool ok[OPEN_R];
#pragma unroll
                for (int r = 0; r < OPEN_R; ++r) {
                    const float rem = bd - (gy2[r % OPEN_NY] + gz2[r / OPEN_NY]) * 0.9999f;
                    // (else every reference of the row is strictly farther than the best: the chord does not reach the
                    // cloud's bounding box, let alone the row; -inf for rows the lane does not have)
                    ok[r] = rem >= ex2;
                    cell_range(ux, __builtin_amdgcn_sqrtf(fmaxf(rem, 0.f)) * 1.00001f + slack, g.org[0], g.inv_h, g.dim[0], xa[r], xb[r]);
                    xa[r] = max(xa[r], x0);
                    xb[r] = min(xb[r], x1);
                    // a row through the 3x3x3 cube the search kernel has dealt with (what it skipped there was farther than its
                    // best of that moment, hence than today's): only what sticks out of the cube on ONE side -- the usual open
                    // query's ball leaves the cube through one face, its cap is a handful of cells; a chord that sticks out on
                    // both sides (a ball of more than 1.5 cells) is taken whole
                    if (iny[r % OPEN_NY] && inz[r / OPEN_NY]) {
                        const bool left = xa[r] < cx - 1, right = xb[r] > cx + 1;
                        if (left && !right) xb[r] = cx - 2;
                        if (right && !left) xa[r] = cx + 2;
                        ok[r] = ok[r] && (left || right);
                    }
                    ok[r] = ok[r] && xa[r] <= xb[r];
                }
#pragma unroll
                for (int r = 0; r < OPEN_R; ++r) {  // (the loads alone inside their predicate: all of them in flight together)
                    sp[r] = 0u;
                    ea[r] = 0u;
                    if (ok[r]) {
                        const unsigned int row = ((unsigned int)(z0 + zb + r / OPEN_NY) * g.dim[1] + (unsigned int)(y0 + yb + r % OPEN_NY)) * g.dim[0];
                        sp[r] = cell_start[row + xa[r]];
                        ea[r] = cell_start[row + xb[r] + 1];
                    }
                }
#pragma unroll
                for (int r = 0; r < OPEN_R; ++r) ln[r] = ea[r] - sp[r];
                flat2_pass<OPEN_R, U, B, true>(fw, cell_refs, sp, ln, lane);
            }
        }
        if (flat) out[qi] = fw.best[lane];
        else if (valid)
            nn1_finish<2>(cell_refs, cell_start, gd, g, slack, qx, qy, qz, qi, key == ~0ull ? open_keys[k] : key, out, fb_list, fb_count,
                          ball_walk, warm_refs);
        flat_sync();  // (the next chunk overwrites the wave's block)
    }
}

// ---- far queries -------------------------------------------------------------------------------
// A query whose neighbourhood is empty for KMAX cells, or whose nearest point lies further than
// KMAX cells (a source cloud that is still misaligned in ICP, points outside the reference's
// bounding box), used to go straight to the exhaustive kernel: 11k such queries out of 2M cost
// 3.4 ms against 0.34 ms for the other 1.99M.  Now they take three steps:
//   1. exhaustive scan of the SEEDS only (every 64th reference): an upper bound on the NN distance
//   2. k_grid_far: walk the cells the ball of that radius touches, pruning whole rows by their
//      y/z distance and clipping every row's x-range to the chord of the (shrinking) ball
//   3. only if the ball spans more than FAR_SPAN cells per axis: the exhaustive kernel

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int lo = __shfl_xor((unsigned int)v, off, 64);
        const unsigned int hi = __shfl_xor((unsigned int)(v >> 32), off, 64);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

// One WAVE per far query.  Rows of cells are visited ring by ring around the query's (clamped)
// row, one row per lane; after every ring the lanes share their best key, so the ball -- and with
// it each row's x-chord and the number of rings still needed -- shrinks as fast as points are
// found.  Ring rho is skipped entirely once (rho-1) cell edges exceed the current best distance.
template <int U>
__global__ void __launch_bounds__(256)
k_grid_far(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
           const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned int* __restrict__ list,
           const unsigned int* __restrict__ count, unsigned long long* __restrict__ out,
           unsigned int* __restrict__ list2, unsigned int* __restrict__ count2) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    const unsigned int cnt = *count;
    const unsigned int lane = threadIdx.x & 63;
    const unsigned int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned int t = wave; t < cnt; t += nwaves) {  // wave-uniform loop
        const unsigned int qi = list[t];
        const float4 qv = q[qi];
        const float qx = qv.x, qy = qv.y, qz = qv.z;
        float ux, uy, uz;  // the query in the grid's frame (cells, gaps, chords); distances take (qx, qy, qz)
        grid_frame(g, qx, qy, qz, ux, uy, uz);
        unsigned long long best = out[qi];  // from the seed scan: a real point, hence a valid upper bound
        bool exhaustive = best == ~0ull;
        const float exg = fmaxf(fmaxf(gd->glo[0] - ux, ux - gd->ghi[0]) - slack, 0.f);  // gap to the cloud's bounding box along the rows
        const float ex2 = exg * exg * 0.9999f;
        const float rb0 = sqrtf(__uint_as_float((unsigned int)(best >> 32))) * 1.00001f + slack;
        if (!exhaustive && !(rb0 * g.inv_h < (float)FAR_SPAN)) exhaustive = true;  // ball too large for a cell walk
        if (!exhaustive) {
            const int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
            const int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
            const int rho_max = max(max(cy, g.dim[1] - 1 - cy), max(cz, g.dim[2] - 1 - cz));
            for (int rho = 0; rho <= rho_max; ++rho) {
                const float bd = __uint_as_float((unsigned int)(best >> 32));
                if (rho >= 1) {
                    const float reach = fmaxf((float)(rho - 1) * g.h - slack, 0.f);
                    if (reach * reach >= bd * 1.00002f) break;  // every remaining ring is farther than the best
                }
                const int nring = rho == 0 ? 1 : 8 * rho;
                for (int r = (int)lane; r < nring; r += 64) {
                    int dy, dz;
                    if (rho == 0) { dy = 0; dz = 0; }
                    else if (r < 2 * rho + 1) { dz = -rho; dy = r - rho; }
                    else if (r < 2 * (2 * rho + 1)) { dz = rho; dy = r - (2 * rho + 1) - rho; }
                    else if (r < 2 * (2 * rho + 1) + (2 * rho - 1)) { dy = -rho; dz = r - 2 * (2 * rho + 1) - rho + 1; }
                    else { dy = rho; dz = r - 2 * (2 * rho + 1) - (2 * rho - 1) - rho + 1; }
                    const int y = cy + dy, z = cz + dz;
                    if (y < 0 || y >= g.dim[1] || z < 0 || z >= g.dim[2]) continue;
                    const float gy = axis_gap(uy, y, g.dim[1], g.org[1], g.h, slack), gz = axis_gap(uz, z, g.dim[2], g.org[2], g.h, slack);
                    const float lbd = __uint_as_float((unsigned int)(best >> 32));
                    const float rem = lbd * 1.00002f - (gz * gz + gy * gy);
                    // the whole row is at least as far as the current best -- or its chord ends short of the cloud's bounding box
                    if (!(rem > 0.f) || rem < ex2) continue;
                    int rx0, rx1;
                    cell_range(ux, sqrtf(rem) + slack, g.org[0], g.inv_h, g.dim[0], rx0, rx1);
                    const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
                    best = scan_span<U>(cell_refs, cell_start[row + rx0], cell_start[row + rx1 + 1], qx, qy, qz, best);
                }
                best = wave_min_u64(best);
            }
            if (lane == 0) out[qi] = best;
        } else if (lane == 0) {
            out[qi] = ~0ull;
            list2[atomicAdd(count2, 1u)] = qi;
        }
    }
}

// order[t] = original position of the t-th point of the cell-sorted references
__global__ void __launch_bounds__(256)
k_order_of_cell_refs(const float4* __restrict__ cell_refs, const GridDev* __restrict__ gd, unsigned int* __restrict__ order) {
    const unsigned int nv = gd->n_valid;
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < nv; t += gridDim.x * blockDim.x)
        order[t] = (unsigned int)__float_as_int(cell_refs[t].w);
}

// sort the queries by reference-grid cell so neighbouring lanes walk the same rows
int grid_sort_queries(pcc_index* ix, const float4* q, size_t nq, unsigned int** order_dev,
                      unsigned int** n_sorted_dev) {
    const unsigned int n = (unsigned int)nq;
    unsigned int* n_sorted = nullptr;  // points at the sort's grand total (stays valid until the next sort)
    ev_mark(ix, EV_SORT0);
    PCC_TRY(ix->scratch_g.reserve((size_t)n * sizeof(unsigned int) + 256));
    unsigned int* ord = ix->scratch_g.as<unsigned int>();
    if (q == ix->refs.as<float4>() && nq == ix->n_orig && ix->has_grid) {
        // a SELF query (SOR, normals, region growing: the indexed cloud asks about itself): the index already holds these points in
        // cell order -- their positions are the order, no sort (45-60 us of a 1.2-ms k-NN call at 1M points)
        unsigned int g = (n + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(k_order_of_cell_refs, dim3(g), dim3(256), 0, ix->stream, ix->cell_refs.as<float4>(), ix->d_grid.as<GridDev>(), ord);
        PCC_HIP(hipGetLastError());
        ev_mark(ix, EV_SORT1);
        *order_dev = ord;
        *n_sorted_dev = &ix->d_grid.as<GridDev>()->n_valid;
        return PCC_OK;
    }
    PCC_TRY(cell_sort(ix, q, nq, false, nullptr, ord, nullptr, &n_sorted));
    ev_mark(ix, EV_SORT1);
    *order_dev = ord;
    *n_sorted_dev = n_sorted;
    return PCC_OK;
}

// can the k = 1 search of this handle apply the ICP loop's transform itself (flat kernel form)?
bool grid_nn1_takes_transform(const pcc_index* ix) { return ix->opt.nn1_kernel != 0 && ix->n_orig <= F2_MAX_REFS; }

int grid_nn1(pcc_index* ix, const float4* q, size_t nq, unsigned long long* out) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)nq;
    PCC_TRY(ix->scratch_d.reserve(((size_t)n * 2 + 128) * sizeof(unsigned int) + 256));
    unsigned int* fb_list = ix->scratch_d.as<unsigned int>();
    unsigned int* fb_count = ix->small.as<unsigned int>() + 32;
    // zeroed by the query pack kernel of this call; searches that re-use packed queries (the ICP
    // loop transforms them in place) have no pack and zero it here
    if (!ix->fb_zeroed) {  // fallback and far counters, the sharded open-lane counters
        PCC_HIP(hipMemsetAsync(fb_count, 0, 8, s));
        PCC_HIP(hipMemsetAsync(ix->small.as<unsigned int>() + PCC_OPEN_CTR0, 0, PCC_OPEN_SHARDS * PCC_OPEN_CTR_STRIDE * 4, s));
    }
    ix->fb_zeroed = false;
    unsigned int *order = nullptr, *n_sorted = nullptr;
    if (ix->pre_order && ix->pre_order_nq == nq && !ix->keep_order) {
        order = ix->pre_order;  // sorted beside the index build by the caller (api.hip: PrepOverlap), for this search only
        n_sorted = ix->pre_nsorted;
        ix->pre_order = nullptr;
    } else if (ix->keep_order && ix->order_valid && ix->order_nq == nq) {
        order = ix->order_ptr;  // (any permutation of the valid queries is correct; this one is still coherent)
        n_sorted = ix->order_nsorted;
    } else {
        PCC_TRY(grid_sort_queries(ix, q, nq, &order, &n_sorted));
        ix->order_valid = ix->keep_order;
        ix->order_nq = nq;
        ix->order_ptr = order;
        ix->order_nsorted = n_sorted;
    }
    ev_mark(ix, EV_MAIN0);
    const int BS = 256;  // 128 and 512 measured 9-12 % slower (fewer lanes to pack / longer wait at the barrier)
    // consecutive workgroups per XCD (see k_grid_nn1; PCC_OPT_XCD_RUN) -- at most a sixteenth of the launch, so that every XCD gets
    // at least two runs (the kernels leave the last, partial round of runs unmapped: a launch smaller than 8 runs would lose the
    // steering altogether)
    const unsigned int xcd_run = std::max(1u, std::min((unsigned int)ix->opt.xcd_run, ((n + 127u) / 128u) / 16u));
    // 4 candidate loads in flight per lane: 2 and 8 measured 153 and 151 us against 142 at 1M x 1M
    // phase 2 as the ball outside the finished cube, except in ICP passes: while the source is still misaligned the
    // balls are several cells wide and the per-row chord arithmetic costs more than the rows it drops
    // (2M x 2M, 50 passes: 26.5 ms with the plain box, 29.1 ms with the ball; 10M x 10M sorted: 1047 vs 1030 us; again with the
    // open lanes in their own kernel, round 3: 19.5 vs 21.1 ms)
    const bool ball_walk = !ix->keep_order;
    const bool warm = ix->warm_start && ix->keep_order;  // out[] holds the previous pass's keys of the SAME queries (pcc_icp_align)
    // PCC_OPT_NN1_KERNEL: 0 one lane per query (k_grid_nn1); 1 rows drained flat (k_grid_nn1_flat2), the lanes it leaves open
    // finished by k_nn1_open from 2M queries on and in place below (a compacted list of 100k queries is a few hundred
    // waves whose dependent loads nothing hides: 116 vs 99 us at 1M x 1M; 688 vs 792 us at 10M x 10M); 2 / 3 force the
    // list / the in-place finish (tests, measurements)
    int form = ix->opt.nn1_kernel;
    if (form != 0 && ix->n_orig > F2_MAX_REFS) form = 0;  // (the packed span record holds 26 bits of reference position)
    // (the folded transform needs the flat kernel and the identity order; the caller checks with grid_nn1_takes_transform)
    const float* pre_T = form != 0 && order == nullptr ? ix->pre_transform : nullptr;
    if (ix->pre_transform && !pre_T) { set_error("internal: pre_transform without the flat kernel / identity order"); return PCC_ERR_INVALID; }
    if (form != 0) {
        const bool listed = form == 2 || (form == 1 && nq >= 2000000);
        const unsigned int dm = (unsigned int)ix->opt.nn1_dense_min;
        const unsigned int f2_bs = 128u, f2_grid = (n + f2_bs - 1) / f2_bs;
        const unsigned int shard_cap = (f2_grid + PCC_OPEN_SHARDS - 1) / PCC_OPEN_SHARDS * f2_bs;
        const size_t list_cap = (size_t)shard_cap * PCC_OPEN_SHARDS;
        unsigned int* open_list = nullptr;
        unsigned long long* open_keys = nullptr;
        unsigned int* open_total = ix->small.as<unsigned int>() + PCC_OPEN_CTR0;
        if (listed) {
            PCC_TRY(ix->scratch_f.reserve(list_cap * 12 + 64));
            open_list = ix->scratch_f.as<unsigned int>();
            open_keys = reinterpret_cast<unsigned long long*>(ix->scratch_f.as<char>() + ((list_cap * 4 + 15) & ~(size_t)15));
        }
#define PCC_F2_ARGS ix->cell_refs.as<float4>(), ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, order, n_sorted, n, out, fb_list, \
                               fb_count, xcd_run, ball_walk, warm ? ix->refs.as<float4>() : nullptr, dm, open_list, open_keys, open_total, pre_T, \
                               const_cast<float4*>(q)
        ix->open_pending = listed;
        if (!listed) ix->stats[7] = 0;
        if (listed) {
            hipLaunchKernelGGL((k_grid_nn1_flat2<4, 4, 2, true>), dim3(f2_grid), dim3(f2_bs), 0, s, PCC_F2_ARGS);
            // (chunks per shard: enough blocks for every listed query to have a lane at once, at most 64)
            unsigned int chunks = (shard_cap / 4 + 255) / 256;
            chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
            if (ix->opt.nn1_open_flat)
                hipLaunchKernelGGL((k_nn1_open_flat<4, 4, 2>), dim3(chunks * 2 * PCC_OPEN_SHARDS), dim3(128), 0, s, ix->cell_refs.as<float4>(),
                                   ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, open_list, open_keys, open_total, out, fb_list,
                                   fb_count, ball_walk, warm ? ix->refs.as<float4>() : nullptr, shard_cap);
            else
                hipLaunchKernelGGL(k_nn1_open, dim3(chunks * PCC_OPEN_SHARDS), dim3(256), 0, s, ix->cell_refs.as<float4>(), ix->cell_start.as<unsigned int>(),
                                   ix->d_grid.as<GridDev>(), q, open_list, open_keys, open_total, out, fb_list, fb_count, ball_walk,
                                   warm ? ix->refs.as<float4>() : nullptr, shard_cap);
        } else {
            hipLaunchKernelGGL((k_grid_nn1_flat2<4, 4, 2, false>), dim3(f2_grid), dim3(f2_bs), 0, s, PCC_F2_ARGS);
        }
#undef PCC_F2_ARGS
    }
    if (form == 0)
        hipLaunchKernelGGL(k_grid_nn1<4>, dim3((n + BS - 1) / BS), dim3(BS), 0, s, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, order, n_sorted, n, out, fb_list,
                           fb_count, (xcd_run + 1) / 2, ball_walk, warm ? ix->refs.as<float4>() : nullptr);
    PCC_HIP(hipGetLastError());
    ev_mark(ix, EV_MAIN1);
    // queries the cell walk could not resolve.  When an earlier search on this index had such
    // queries (count mirrored to pinned memory by k_unpack; read here WITHOUT waiting, it only
    // steers the choice), take the seed + ball-walk route; otherwise go straight to the exhaustive
    // kernel, which costs one launch when the list is empty
    ev_mark(ix, EV_FB0);
    const unsigned int seen = static_cast<volatile unsigned int*>(ix->pinned)[40];
    if (seen > ix->last_fallback_seen) ix->last_fallback_seen = seen;
    const int far_mode = ix->opt.far_mode;  // -1 auto, 0 off, 1 on
    // (ICP passes always take it: their loop may be enqueued as a whole before the first count comes back)
    const bool far = far_mode == 1 || (far_mode == -1 && (ix->last_fallback_seen >= 64 || ix->keep_order));
    if (far) {
        unsigned int* fb2_list = fb_list + n + 64;
        unsigned int* fb2_count = ix->small.as<unsigned int>() + 33;
        // (fb2_count is the second of the two words zeroed at the top of this call -- or by the pack / transform kernel before it)
        const size_t n_seeds = (ix->n_orig + SEED_STRIDE - 1) / SEED_STRIDE;
        // (a warm-started query brings its bound along: no seed scan)
        if (!warm) PCC_TRY(launch_nn1_brute(s, ix->seeds.as<float4>(), n_seeds, q, n, out, fb_list, fb_count, n, true));
        unsigned int gfar = (n + 3) / 4;  // one wave per listed query, 4 waves per workgroup
        if (gfar > 2048) gfar = 2048;
        hipLaunchKernelGGL((k_grid_far<4>), dim3(gfar), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, fb_list, fb_count, out,
                           fb2_list, fb2_count);
        PCC_HIP(hipGetLastError());
        PCC_TRY(launch_nn1_brute(s, ix->refs.as<float4>(), ix->n_orig, q, n, out, fb2_list, fb2_count, n));
    } else {
        PCC_TRY(launch_nn1_brute(s, ix->refs.as<float4>(), ix->n_orig, q, n, out, fb_list, fb_count, n));
    }
    ev_mark(ix, EV_FB1);
    // pcc_index_stats reads the fallback count lazily (k_unpack mirrors it to pinned memory)
    ix->stats_pending = true;
    ix->last_nq = nq;
    return PCC_OK;
}

PCC_PAIRS_TAKE(grid)

}  // namespace pcc
