// grid_params_device.hpp -- the grid of an index derived ON THE DEVICE from the pack kernel's per-workgroup statistics
// (bounding boxes, non-finite counts): cell edge, dimensions, axis assignment, slack.  Shared by k_grid_params (grid.hip: a
// launch of its own) and k_pack (pack.hip: run by the last workgroup of the pack to finish -- one launch less per build).
#pragma once
#include "pcc_internal.hpp"

namespace pcc {

// One workgroup reduces the pack kernel's per-workgroup rows (invalid count, bbox) and derives
// the grid: cell edge so that the mean occupancy over the non-flat dimensions of the bounding
// box is `ppc` points per cell, grown until the cell count fits nc_cap.  Keeping this on the
// device removes the host round trip (D2H, wait, launch) from every index build: 46 us of a
// 385 us step at 1M points.
// maximum over the 64 lanes, in every lane: DPP row shifts + row broadcasts (no LDS crossbar: a __shfl_xor chain cost
// this kernel 11 us), the result read back from lane 63
__device__ __forceinline__ float wave_max_f32(float x) {
#define PCC_MAX_STEP(CTRL, ROWMASK) \
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp((int)0xff800000u, __float_as_int(x), CTRL, ROWMASK, 0xf, false)))
    PCC_MAX_STEP(0x111, 0xf);  // row_shr:1
    PCC_MAX_STEP(0x112, 0xf);  // row_shr:2
    PCC_MAX_STEP(0x114, 0xf);  // row_shr:4
    PCC_MAX_STEP(0x118, 0xf);  // row_shr:8
    PCC_MAX_STEP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    PCC_MAX_STEP(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef PCC_MAX_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
// k-th largest of each of six values over the lanes of a wave (lanes that do not take part pass -inf)
__device__ __forceinline__ void wave_kth_max6(float v[6], int k) {
    const unsigned int lane = threadIdx.x & 63;
    float m[6];
    for (int it = 0; it < k; ++it) {
#pragma unroll
        for (int a = 0; a < 6; ++a) m[a] = wave_max_f32(v[a]);
        if (it + 1 < k) {
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const unsigned long long hit = __ballot(v[a] == m[a]);
                if (hit && lane == (unsigned int)__ffsll((long long)hit) - 1) v[a] = -__builtin_inff();
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) v[a] = m[a];
}

// trim_k > 0: the grid is laid over a TRIMMED box.  Every row of `blk` is the bounding box of an interleaved
// sample of the cloud (one pack workgroup); a stray point far from the scene inflates one row, not the others.
// Per group of 64 rows the trim_k-th extreme is taken, then the widest group: a handful of outliers no longer
// stretches the cells over empty space (one point at 10 km made every cell 10x wider and sent the search to its
// exhaustive fallback).  Points beyond the box fall into the boundary cells, which every search already treats
// as open-ended (cell_coord clamps, outside_bound2 takes no bound from a face on the grid's edge, cell_range
// clamps): results stay exact, and the handle still reports the true bounding box.
// One workgroup (any multiple of 64 threads up to 1024).  AGENT: the rows were written by other workgroups of the SAME launch
// (the pack kernel's last workgroup to finish runs this, pack.hip) -- they are read past the caches.
template <bool AGENT>
__device__ __forceinline__ float blk_ld(const float* p) {
    if (AGENT) return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}
template <bool AGENT>
__device__ __forceinline__ void grid_params_block(const float* __restrict__ blk, int nblk, unsigned int n, float ppc, unsigned int nc_cap,
                                                  int trim_k, int axes, GridDev* __restrict__ out, GridDev* __restrict__ host_mirror) {
    __shared__ float red[16][8];
    __shared__ float rob[16][6];  // (1024 threads: one wave per group of 64 rows)
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        float tl[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
        float th[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        if (trim_k > 0) {
            for (int g0 = wave * 64; g0 < nblk; g0 += (int)blockDim.x) {  // wave-uniform
                if (nblk - g0 < 16) continue;                  // a short tail group says little
                const int b = g0 + lane;
                const float* r = blk + (size_t)(b < nblk ? b : 0) * 8;
                float e[6];  // -lo (so that the smallest lo is a maximum too), hi
                for (int a = 0; a < 3; ++a) {
                    e[a] = b < nblk ? -blk_ld<AGENT>(r + 1 + a) : -__builtin_inff();
                    e[3 + a] = b < nblk ? blk_ld<AGENT>(r + 4 + a) : -__builtin_inff();
                }
                wave_kth_max6(e, trim_k);
                for (int a = 0; a < 3; ++a) { tl[a] = fminf(tl[a], -e[a]); th[a] = fmaxf(th[a], e[3 + a]); }
            }
        }
        if (lane == 0)
            for (int a = 0; a < 3; ++a) { rob[wave][a] = tl[a]; rob[wave][3 + a] = th[a]; }
    }
    unsigned int bad = 0;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
        const float* r = blk + (size_t)b * 8;
        bad += __float_as_uint(blk_ld<AGENT>(r));
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], blk_ld<AGENT>(r + 1 + a)); hi[a] = fmaxf(hi[a], blk_ld<AGENT>(r + 4 + a)); }
    }
    for (int off = 32; off > 0; off >>= 1) {
        bad += __shfl_down(bad, off, 64);
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][0] = __uint_as_float(bad);
        for (int a = 0; a < 3; ++a) { red[threadIdx.x >> 6][1 + a] = lo[a]; red[threadIdx.x >> 6][4 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    bad = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        bad += __float_as_uint(red[w][0]);
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], red[w][1 + a]); hi[a] = fmaxf(hi[a], red[w][4 + a]); }
    }
    GridDev d;
    d.n_invalid = bad;
    d.n_valid = n - bad;
    d.voxel = 0;
    float ext[3], maxext = 0.f, maxabs = 0.f;
    for (int a = 0; a < 3; ++a) {
        if (d.n_valid == 0) { lo[a] = 0.f; hi[a] = 0.f; }
        d.lo[a] = lo[a];  // the true bounding box is what the handle reports
        d.hi[a] = hi[a];
        if (trim_k > 0 && d.n_valid != 0) {  // the grid lies over the trimmed one (never wider than the true box)
            float tl = __builtin_inff(), th = -__builtin_inff();
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { tl = fminf(tl, rob[w][a]); th = fmaxf(th, rob[w][3 + a]); }
            if (tl <= th) { lo[a] = fmaxf(lo[a], tl); hi[a] = fminf(hi[a], th); }
        }
        ext[a] = hi[a] - lo[a];
        if (!(ext[a] >= 0.f) || !(ext[a] < __builtin_inff())) ext[a] = 0.f;  // overflowed extents: one cell
        maxext = fmaxf(maxext, ext[a]);
        maxabs = fmaxf(maxabs, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
    }
    int nd = 0;
    double vol = 1.0;
    for (int a = 0; a < 3; ++a)
        if (ext[a] > 1e-6f * maxext && ext[a] > 0.f) { vol *= (double)ext[a]; ++nd; }
    double cells_wanted = fmax(1.0, (double)d.n_valid / (double)ppc);
    if (cells_wanted > (double)nc_cap) cells_wanted = (double)nc_cap;
    double hcell = nd ? pow(vol / cells_wanted, 1.0 / (double)nd) : 1.0;
    if (!(hcell > 0.0) || !(hcell < 1e300)) hcell = 1.0;
    // Which coordinate each grid axis follows (GridParams::ax).  A query's neighbourhood is its own row of cells, the rows above
    // and below it (dim0 cells away in memory) and the same rows of the layers before and behind (dim0 * dim1 cells away): the
    // SHORTEST extent goes on axis 1 and the longest on axis 2, so that a layer -- what the searches of neighbouring queries keep
    // re-reading from the L2 -- is as small as the cloud allows.  A room scan is long and wide and 2.7 m high: with z on axis 2
    // (rounds 1-5) the rows of the layer above were a whole floor plan away.  Ties keep x, y, z order.
    GridParams g;
    // (GRID_AXES_MIN_POINTS, pcc_internal.hpp: measured both ways, set to 0)
    // (axes = -2: by extent whatever the size -- tests)
    grid_axes_for(ext, axes == -1 && d.n_valid < GRID_AXES_MIN_POINTS ? 0 : axes, g.ax);
    for (int iter = 0; iter < 200; ++iter) {  // grow the cell until the grid fits nc_cap
        g.h = (float)hcell;
        g.inv_h = 1.0f / g.h;
        if (!(g.inv_h > 0.f) || !(g.inv_h < __builtin_inff()) || !(g.h > 0.f)) { g.h = 1.f; g.inv_h = 1.f; }
        double tot = 1.0;
        for (int a = 0; a < 3; ++a) {
            double dd = floor((double)pick3(ext, g.ax[a]) * (double)g.inv_h) + 1.0;
            if (dd > 1048576.0) dd = 1048576.0;
            g.dim[a] = (int)dd;
            tot *= dd;
        }
        if (tot <= (double)nc_cap) break;
        hcell *= 1.26;
    }
    if ((double)g.dim[0] * g.dim[1] * g.dim[2] > (double)nc_cap) { g.dim[0] = g.dim[1] = g.dim[2] = 1; }  // cannot happen; stay in bounds
    for (int a = 0; a < 3; ++a) {
        g.org[a] = pick3(lo, g.ax[a]);
        d.glo[a] = pick3(d.lo, g.ax[a]);
        d.ghi[a] = pick3(d.hi, g.ax[a]);
    }
    g.ncells = g.dim[0] * g.dim[1] * g.dim[2];
    d.g = g;
    float far = 0.f;
    for (int a = 0; a < 3; ++a) far = fmaxf(far, fmaxf(fabsf(g.org[a]), fabsf(g.org[a] + g.dim[a] * g.h)));
    d.slack = 4e-6f * far + 1e-6f * g.h;
    *out = d;
    *host_mirror = d;  // pinned host memory: a separate 5 us D2H copy on the stream is avoided
}


}  // namespace pcc
