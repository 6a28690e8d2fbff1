// grid_device.hpp -- device helpers shared by the GRID-engine kernels (grid.hip, knn.hip,
// cluster.hip).  Built with -ffp-contract=off: dist2 must round like FLANN's L2_Simple.
#pragma once
#include "pcc_internal.hpp"

namespace pcc {

constexpr int GRID_KMAX = 8;  // largest cube half-width before the exhaustive fallback

__device__ __forceinline__ int cell_coord(float v, float org, float inv_h, int dim) {
    // clamp in float first (no int overflow); the SAME expression runs at build and query time
    float t = fminf(fmaxf((v - org) * inv_h, 0.f), (float)(dim - 1));
    return (int)t;
}
// A point's coordinates along the GRID's axes: axis 0 runs along the cells of a row (fastest in the linear cell id), axis 1
// over the rows of a layer, axis 2 over the layers.  Which coordinate of the cloud each of them follows is chosen per index
// from the extents of the cloud (GridParams::ax, k_grid_params) -- the cell arithmetic of every kernel works in this frame
// (ux, uy, uz; cells cx, cy, cz), the DISTANCES stay in the cloud's own (x, y, z): their bits depend on the order of the sum.
__device__ __forceinline__ float axis_pick(int a, float x, float y, float z) { return a == 0 ? x : (a == 1 ? y : z); }
__device__ __forceinline__ void grid_frame(const GridParams& g, float x, float y, float z, float& ux, float& uy, float& uz) {
    ux = axis_pick(g.ax[0], x, y, z);
    uy = axis_pick(g.ax[1], x, y, z);
    uz = axis_pick(g.ax[2], x, y, z);
}
__device__ __forceinline__ unsigned int cell_id(const float4& v, const GridParams& g) {
    float ux, uy, uz;
    grid_frame(g, v.x, v.y, v.z, ux, uy, uz);
    int cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
    int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
    int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
    return ((unsigned int)cz * g.dim[1] + cy) * g.dim[0] + cx;
}

// PCL VoxelGrid voxel of a point (pcl/filters/impl/voxel_grid.hpp):
//   ijk_a = int(floor(p_a * inverse_leaf) - float(min_b_a)),  idx = ijk0 + ijk1*div0 + ijk2*div0*div1
// here org[a] = float(min_b_a), inv_h = inverse_leaf (float), dim = div_b
__device__ __forceinline__ unsigned int voxel_id(const float4& v, const GridParams& g) {
    const int i0 = min(max((int)(floorf(v.x * g.inv_h) - g.org[0]), 0), g.dim[0] - 1);
    const int i1 = min(max((int)(floorf(v.y * g.inv_h) - g.org[1]), 0), g.dim[1] - 1);
    const int i2 = min(max((int)(floorf(v.z * g.inv_h) - g.org[2]), 0), g.dim[2] - 1);
    return ((unsigned int)i2 * g.dim[1] + i1) * g.dim[0] + i0;
}

// ((dx*dx) + dy*dy) + dz*dz, every operation rounded to float.  x and y go through the packed fp32 pipe
// (v_pk_add_f32 / v_pk_mul_f32: two IEEE operations per instruction, same rounding as the scalar forms), 6 VALU
// instructions per candidate instead of 8 -- the pruned kernels are VALU-issue bound.
typedef float pcc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dist2_nc(float qx, float qy, float qz, const float4& r) {
    const pcc_f2 q2 = {qx, qy}, r2 = {r.x, r.y};
    const pcc_f2 d2 = q2 - r2;
    const pcc_f2 s2 = d2 * d2;
    const float dz = qz - r.z;
    float d = s2.x + s2.y;
    d = d + dz * dz;
    return d;
}

// ---- pair counter of the PROFILING build (make prof: -DPCC_COUNT_PAIRS, libpcc_nn_prof.so) -------------------------
// Every (query, reference) distance a pruned kernel evaluates is counted, so that pairs/s -- and with 9 VALU operations
// per pair the fraction of the non-FMA fp32 roof -- can be reported for kernels whose HBM traffic says nothing about
// them.  One sharded counter array per translation unit (static: no relocatable device code), summed and
// cleared by pcc_index_stats.  The compiler folds a wave's increments into one atomic.  Not compiled into libpcc_nn.so.
#ifdef PCC_COUNT_PAIRS
static __device__ unsigned long long g_pcc_pairs[64 * 16];
#define PCC_PAIR(pred) do { if (pred) atomicAdd(&g_pcc_pairs[(blockIdx.x & 63u) * 16u], 1ull); } while (0)
#define PCC_PAIRS_TAKE(NAME)                                                              \
    unsigned long long pairs_take_##NAME() {                                              \
        static unsigned long long h[64 * 16];                                             \
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pcc_pairs), sizeof(h)) != hipSuccess) return 0; \
        unsigned long long sum = 0;                                                       \
        for (int i = 0; i < 64 * 16; ++i) { sum += h[i]; h[i] = 0; }                      \
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pcc_pairs), h, sizeof(h));                   \
        return sum;                                                                       \
    }
#else
#define PCC_PAIR(pred) ((void)0)
#define PCC_PAIRS_TAKE(NAME)
#endif

// the counted form: every lane that gets here evaluates a candidate (sites that also run clamped tail lanes use
// dist2_nc and count under their own predicate)
__device__ __forceinline__ float dist2(float qx, float qy, float qz, const float4& r) {
    PCC_PAIR(true);
    return dist2_nc(qx, qy, qz, r);
}

// lower bound (squared, shrunk) of the distance from q to any point outside the cell cube
// [x0..x1] x [y0..y1] x [z0..z1]; +inf when the cube covers the whole grid.  (qx, qy, qz): the query in the GRID's frame
__device__ __forceinline__ float outside_bound2(float qx, float qy, float qz, int x0, int x1, int y0,
                                                int y1, int z0, int z1, const GridParams& g, float slack) {
    float lb = __builtin_inff();
    if (x0 > 0) lb = fminf(lb, qx - (g.org[0] + x0 * g.h));
    if (x1 < g.dim[0] - 1) lb = fminf(lb, (g.org[0] + (x1 + 1) * g.h) - qx);
    if (y0 > 0) lb = fminf(lb, qy - (g.org[1] + y0 * g.h));
    if (y1 < g.dim[1] - 1) lb = fminf(lb, (g.org[1] + (y1 + 1) * g.h) - qy);
    if (z0 > 0) lb = fminf(lb, qz - (g.org[2] + z0 * g.h));
    if (z1 < g.dim[2] - 1) lb = fminf(lb, (g.org[2] + (z1 + 1) * g.h) - qz);
    if (lb == __builtin_inff()) return lb;  // no face left: the cube is the whole grid
    lb = fmaxf(lb - slack, 0.f);      // absolute slack: cell-boundary rounding
    // relative slack: rounding of the fp32 distances.  Kept FINITE: with coordinates around 1e20 the square overflows,
    // and +inf means "nothing outside the cube" to the callers -- while the distances it stands for overflow as well
    // and tie with a lower index out there (found by tools/fuzz_gpu.py)
    return fminf(lb * lb * 0.9999f, 3.4028234664e38f);
}


// range of cells [c0, c1] along one axis that can hold points within r of coordinate v
__device__ __forceinline__ void cell_range(float v, float r, float org, float inv_h, int dim, int& c0, int& c1) {
    c0 = cell_coord(v - r, org, inv_h, dim);
    c1 = cell_coord(v + r, org, inv_h, dim);
}


// The cells a ball of squared radius bd around q can share with a REFERENCE: every reference lies inside the true
// bounding box [lo, hi] of the indexed cloud, so along one axis the ball only matters within
// sqrt(bd - (gap to the box along the other two axes)^2) of q.  For a query inside the box this is the plain ball; for
// one outside it (an ICP source that is still misaligned, 5 % of the C4 cloud for dozens of passes) the box of cells
// shrinks from (2 d / h)^3 to the few cells of the cap the ball cuts out of the cloud.  Gaps shrunk by the slack and
// 1e-4 relative, the radius stretched by 1e-5 and the slack: conservative like every other bound here.
// Everything here is in the GRID's frame (grid_frame): the query as (ux, uy, uz), the box as GridDev::glo / ghi.
struct BallBox {
    int x0, x1, y0, y1, z0, z1;
    float ex2, ey2, ez2;  // squared (shrunk) gaps from q to the cloud's bounding box, per axis
    bool finite;          // false: the ball is unbounded (no best yet, or an overflowed distance)
};
__device__ __forceinline__ BallBox ball_box(float qx, float qy, float qz, float bd, const GridDev* __restrict__ gd, const GridParams& g,
                                            float slack) {
    BallBox b;
    const float ex = fmaxf(fmaxf(gd->glo[0] - qx, qx - gd->ghi[0]) - slack, 0.f);
    const float ey = fmaxf(fmaxf(gd->glo[1] - qy, qy - gd->ghi[1]) - slack, 0.f);
    const float ez = fmaxf(fmaxf(gd->glo[2] - qz, qz - gd->ghi[2]) - slack, 0.f);
    b.ex2 = ex * ex * 0.9999f;
    b.ey2 = ey * ey * 0.9999f;
    b.ez2 = ez * ez * 0.9999f;
    const float rx = sqrtf(fmaxf(bd - (b.ey2 + b.ez2), 0.f)) * 1.00001f + slack;
    const float ry = sqrtf(fmaxf(bd - (b.ex2 + b.ez2), 0.f)) * 1.00001f + slack;
    const float rz = sqrtf(fmaxf(bd - (b.ex2 + b.ey2), 0.f)) * 1.00001f + slack;
    b.finite = bd < __builtin_inff();  // (false for NaN too)
    cell_range(qx, rx, g.org[0], g.inv_h, g.dim[0], b.x0, b.x1);
    cell_range(qy, ry, g.org[1], g.inv_h, g.dim[1], b.y0, b.y1);
    cell_range(qz, rz, g.org[2], g.inv_h, g.dim[2], b.z0, b.z1);
    return b;
}

}  // namespace pcc
