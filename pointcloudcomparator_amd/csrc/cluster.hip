// cluster.hip -- Euclidean clustering as connected components of the radius graph (gfx950).
//
// Replaces pcl::EuclideanClusterExtraction::extract (reference src/segmentation.cpp:125-131):
// PCL grows clusters with a sequential BFS over radiusSearch results; the clusters it returns
// are exactly the connected components of the graph {(i, j) : d2(i, j) < r2}, so
// the GPU path builds the components directly with a lock-free union-find over the GRID index
// and never materialises neighbour lists:
//   link    : every point scans the cells its r-ball touches and unions itself with each
//             neighbour of LOWER position (each edge once).  Roots only ever point to smaller
//             positions, so a component's root is its lowest member -- PCL's seed point.
//   flatten : parent[i] = root(i)
//   size    : histogram of roots
//   collect : roots whose size lies in [min, max]; the (few) survivors are ordered on the host
//             by size descending, lowest member first among equal sizes
//   label   : labels[original index] = cluster id or -1
//
// link, fast path (k_uf_link_cells): a second grid with cell edge 0.57 r.  The cell diagonal is then below r, so
// all points of a cell are mutually connected -- they are chained to the cell's first point without a single
// distance test -- and two cells can only be connected when their indices differ by at most 2 per axis.
// For every occupied cell the 62 "forward" neighbour cells are split among the cell's points; a neighbour
// cell whose first point already has the same root is skipped, otherwise its points are tested against
// the cell's until the first pair within r is found (one union per connected cell pair).  The search
// grid's cells (0.5 points per cell on average over the bounding box) hold tens of points each inside the
// objects, where the per-point ball scan does 500-1000 distance tests to find ~90 neighbours; the cell
// walk does a few tens.  Same components, hence the same clusters.
#include "pcc_internal.hpp"
#include "grid_device.hpp"
#include "uf_device.hpp"
#include "lane_ops.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace pcc {

__global__ void __launch_bounds__(256)
k_uf_init(unsigned int* __restrict__ parent, unsigned int n) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) parent[i] = i;
}

// thread t owns the t-th point in CELL order (neighbouring lanes touch neighbouring cells)
__global__ void __launch_bounds__(256)
k_uf_link(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
          const GridDev* __restrict__ gd, float r, float r2, unsigned int* __restrict__ parent) {
    const GridParams g = gd->g;
    const float slack = gd->slack;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= gd->n_valid) return;  // cell_refs holds the valid points only
    const float4 me = cell_refs[t];
    const unsigned int mypos = (unsigned int)__float_as_int(me.w);
    int x0, x1, y0, y1, z0, z1;
    const float rr = r + slack;
    float ux, uy, uz;  // the point in the grid's frame (grid_device.hpp)
    grid_frame(g, me.x, me.y, me.z, ux, uy, uz);
    cell_range(ux, rr, g.org[0], g.inv_h, g.dim[0], x0, x1);
    cell_range(uy, rr, g.org[1], g.inv_h, g.dim[1], y0, y1);
    cell_range(uz, rr, g.org[2], g.inv_h, g.dim[2], z0, z1);
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y) {
            const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
            const unsigned int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
            for (unsigned int p = s; p < e; ++p) {
                const float4 o = cell_refs[p];
                const unsigned int opos = (unsigned int)__float_as_int(o.w);
                if (opos < mypos && dist2(me.x, me.y, me.z, o) < r2) uf_union(parent, mypos, opos);
            }
        }
}

// thread t owns the t-th valid point in the order of the CLUSTERING grid (cr2 / cs2 / gd2)
__global__ void __launch_bounds__(256)
k_uf_link_cells(const float4* __restrict__ cr2, const unsigned int* __restrict__ cs2, const GridDev* __restrict__ gd2,
                float r2, unsigned int* __restrict__ parent) {
    const GridParams g = gd2->g;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= gd2->n_valid) return;
    const float4 me = cr2[t];
    const unsigned int mypos = (unsigned int)__float_as_int(me.w);
    float ux, uy, uz;  // the point in the grid's frame (grid_device.hpp)
    grid_frame(g, me.x, me.y, me.z, ux, uy, uz);
    const int cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
    const int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
    const int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
    const unsigned int c = ((unsigned int)cz * g.dim[1] + cy) * g.dim[0] + cx;
    const unsigned int a0 = cs2[c], a1 = cs2[c + 1];
    const unsigned int first = (unsigned int)__float_as_int(cr2[a0].w);
    if (t != a0) uf_union(parent, mypos, first);  // same cell: within r by construction
    // forward half of the 5 x 5 x 5 neighbourhood, 62 cells, numbered 0..61; this point takes every cnt-th
    const unsigned int cnt = a1 - a0, j = t - a0;
    for (unsigned int nb = j; nb < 62; nb += cnt) {
        const int lin = (int)nb + 63;  // 62 = own cell in the 5x5x5 numbering (dz,dy,dx from -2): forward = lin > 62
        const int dz = lin / 25 - 2, dy = (lin / 5) % 5 - 2, dx = lin % 5 - 2;
        const int x = cx + dx, y = cy + dy, z = cz + dz;
        if (x < 0 || x >= g.dim[0] || y < 0 || y >= g.dim[1] || z >= g.dim[2]) continue;
        const unsigned int cb = ((unsigned int)z * g.dim[1] + y) * g.dim[0] + x;
        const unsigned int b0 = cs2[cb], b1 = cs2[cb + 1];
        if (b0 == b1) continue;
        const unsigned int other = (unsigned int)__float_as_int(cr2[b0].w);
        if (uf_find(parent, first) == uf_find(parent, other)) continue;  // linked already (through anything)
        bool linked = false;
        for (unsigned int pb = b0; pb < b1 && !linked; ++pb) {
            const float4 o = cr2[pb];
            for (unsigned int pa = a0; pa < a1; ++pa) {
                const float4 m = cr2[pa];
                if (dist2(m.x, m.y, m.z, o) < r2) {
                    uf_union(parent, (unsigned int)__float_as_int(m.w), (unsigned int)__float_as_int(o.w));
                    linked = true;
                    break;
                }
            }
        }
    }
}

// The same links with the lanes over a cell's NEIGHBOURS.  In the kernel above a lane owns a point and walks its share of
// the 62 forward neighbour cells one after the other, each step a chain of dependent L2 round trips (cell bounds -> first
// point -> two root walks -> points): 10 of 64 lanes active, 86 % of the wave time waiting, 3.5 ms at 5M points.  Here a
// wave takes the cells that START among its 64 points one at a time and gives each of the 62 neighbour cells to a lane:
// the 62 chains run side by side, the neighbour offsets are per-lane constants (no division in the loop), and the
// bounds of 5 x-adjacent neighbour cells share a cache line: 3.48 -> 2.24 ms.  What is left is the "linked already?" walk
// of 84M cell pairs through agent-scope loads (the per-XCD L2s are not coherent, so each is a trip past them; the head
// loop with the bounds alone is 0.3 ms, with the neighbour's first point 0.6): ~170M uncached 4-byte loads in ~1.6 ms
// is what the fabric gives.  A pre-check through the caches (stale parents prove a link when equal) changed nothing.
// Same unions offered (a pair of cells is linked by the first
// pair of points within r it finds, or skipped when the roots already agree), same components.
__global__ void __launch_bounds__(256)
k_uf_link_cells_wave(const float4* __restrict__ cr2, const unsigned int* __restrict__ cs2, const GridDev* __restrict__ gd2,
                     float r2, unsigned int* __restrict__ parent) {
    const GridParams g = gd2->g;
    const unsigned int n_valid = gd2->n_valid;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int lane = threadIdx.x & 63;
    if ((t & ~63u) >= n_valid) return;  // wave-uniform
    const bool have = t < n_valid;
    int cx = 0, cy = 0, cz = 0;
    unsigned int a0 = 0, a1 = 0, first = 0;
    if (have) {
        const float4 me = cr2[t];
        float ux, uy, uz;  // the point in the grid's frame (grid_device.hpp)
        grid_frame(g, me.x, me.y, me.z, ux, uy, uz);
        cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
        cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
        cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
        const unsigned int c = ((unsigned int)cz * g.dim[1] + cy) * g.dim[0] + cx;
        a0 = cs2[c];
        a1 = cs2[c + 1];
        first = (unsigned int)__float_as_int(cr2[a0].w);
        if (t != a0) uf_union(parent, (unsigned int)__float_as_int(me.w), first);  // same cell: within r by construction
    }
    // this lane's forward neighbour in the 5 x 5 x 5 numbering (62 = the own cell; forward = beyond it)
    const int lin = (int)lane + 63;
    const int dz = lin / 25 - 2, dy = (lin / 5) % 5 - 2, dx = lin % 5 - 2;
    unsigned long long heads = __ballot(have && t == a0);
    while (heads) {
        const int h = __builtin_ctzll(heads);
        heads &= heads - 1;
        const int hx = __builtin_amdgcn_readlane(cx, h), hy = __builtin_amdgcn_readlane(cy, h), hz = __builtin_amdgcn_readlane(cz, h);
        const unsigned int ha0 = (unsigned int)__builtin_amdgcn_readlane((int)a0, h);
        const unsigned int ha1 = (unsigned int)__builtin_amdgcn_readlane((int)a1, h);
        const unsigned int hfirst = (unsigned int)__builtin_amdgcn_readlane((int)first, h);
        const int x = hx + dx, y = hy + dy, z = hz + dz;
        if (lane >= 62 || x < 0 || x >= g.dim[0] || y < 0 || y >= g.dim[1] || z >= g.dim[2]) continue;
        const unsigned int cb = ((unsigned int)z * g.dim[1] + y) * g.dim[0] + x;
        const unsigned int b0 = cs2[cb], b1 = cs2[cb + 1];
        if (b0 == b1) continue;
        const unsigned int other = (unsigned int)__float_as_int(cr2[b0].w);
        if (uf_linked(parent, hfirst, other)) continue;  // linked already (through anything)
        bool linked = false;
        for (unsigned int pb = b0; pb < b1 && !linked; ++pb) {
            const float4 o = cr2[pb];
            for (unsigned int pa = ha0; pa < ha1; ++pa) {
                const float4 m = cr2[pa];
                if (dist2(m.x, m.y, m.z, o) < r2) {
                    uf_union(parent, (unsigned int)__float_as_int(m.w), (unsigned int)__float_as_int(o.w));
                    linked = true;
                    break;
                }
            }
        }
    }
}

// ---- union-find over CELLS (round 5; PCC_OPT_EC_CELLS = 3, the default) ------------------------------------------
// The kernels above keep one parent word per POINT and ask "linked already?" for each of the 84M (cell, forward neighbour)
// pairs of a 5M-point scene through agent-scope loads -- the per-XCD L2s are not coherent, so every one of the ~170M reads
// is a trip past them: 1.6 of the link kernel's 2.24 ms.  But a cell of the clustering grid IS a clique (diagonal < r), so the
// graph whose components are wanted has the occupied cells as its nodes (1.3M instead of 5M, named by the sorted position of
// their first point, which the bound load hands out for free -- no gather of the neighbour's first point), and the question
// can be asked of a parent array that no union is touching:
//   phase 1 (k_ecc_link_faces)  every occupied cell is tested against its +x / +y / +z neighbour and hooked to it when a pair
//                               of points lies within r -- no "linked already?" at all, 3 pairs per cell.  Inside an object
//                               these face links alone connect nearly everything.
//   flatten (k_ecc_flatten)     a kernel boundary later nothing moves any more: parent[c] = root(c) with plain cached loads.
//   phase 2 (k_ecc_link_rest)   the other 59 forward neighbours, lanes over neighbours as above: two cells are linked
//                               already when their flat roots are EQUAL -- two plain loads through the caches, one of them
//                               wave-uniform.  A stale root can only say "not linked" too often, never "linked" wrongly
//                               (parents only move towards the root), so whatever phase 1 did not join is tested and hooked
//                               here with the agent-scope union: every edge of the cell graph is either proven redundant or
//                               offered, the components are those of the point graph.
//   flatten + sizes, labels     points inherit their cell's root.
// Roots are the lowest cell of a component in grid order; PCL's tie order among equal sizes (lowest member index first) comes
// from a per-component minimum of the original indices.
__global__ void __launch_bounds__(256)
k_ecc_init(unsigned int* __restrict__ cparent, unsigned int* __restrict__ size, unsigned int* __restrict__ minidx,
           int* __restrict__ id_of_root, unsigned int n, int32_t* __restrict__ labels, unsigned int n_labels) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < max(n, n_labels); i += gridDim.x * blockDim.x) {
        if (i < n) { cparent[i] = i; size[i] = 0u; minidx[i] = 0xffffffffu; id_of_root[i] = -1; }
        if (i < n_labels) labels[i] = -1;
    }
}

// first pair of points of the cells [a0, a1) x [b0, b1) within r (squared: r2)?
__device__ __forceinline__ bool ecc_cells_touch(const float4* __restrict__ cr2, unsigned int a0, unsigned int a1, unsigned int b0,
                                                unsigned int b1, float r2) {
    for (unsigned int pb = b0; pb < b1; ++pb) {
        const float4 o = cr2[pb];
        for (unsigned int pa = a0; pa < a1; ++pa) {
            const float4 m = cr2[pa];
            if (dist2(m.x, m.y, m.z, o) < r2) return true;
        }
    }
    return false;
}

// phase 1: the three forward FACE neighbours of every occupied cell, tested and hooked.  One lane per point finds the heads of
// cells (about one lane in four); the wave then deals its heads out THREE LANES EACH, one per face neighbour, 21 heads at a
// time -- with one lane per head walking its three neighbours in turn a wave ran at a quarter of its lanes for three times as
// long (0.59 ms at 5M points)
__global__ void __launch_bounds__(256)
k_ecc_link_faces(const float4* __restrict__ cr2, const unsigned int* __restrict__ cs2, const GridDev* __restrict__ gd2, float r2,
                 unsigned int* __restrict__ cparent) {
    const GridParams g = gd2->g;
    const unsigned int n_valid = gd2->n_valid;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int lane = threadIdx.x & 63;
    if ((t & ~63u) >= n_valid) return;  // wave-uniform
    __shared__ unsigned int head_lane[4][64];
    unsigned int* hl = head_lane[threadIdx.x >> 6];
    int cx = 0, cy = 0, cz = 0;
    unsigned int c = 0, a0 = 0xffffffffu, a1 = 0;
    if (t < n_valid) {
        const float4 me = cr2[t];
        float ux, uy, uz;  // the point in the grid's frame (grid_device.hpp)
        grid_frame(g, me.x, me.y, me.z, ux, uy, uz);
        cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
        cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
        cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
        c = ((unsigned int)cz * g.dim[1] + cy) * g.dim[0] + cx;
        a0 = cs2[c];
        a1 = cs2[c + 1];
    }
    const bool head = t == a0;
    const unsigned long long heads = __ballot(head);
    const unsigned int nheads = (unsigned int)__popcll(heads);
    if (head) hl[__popcll(heads & ((1ull << lane) - 1ull))] = lane;
    wave_lds_sync();
    const unsigned int sub = lane / 3u, nb = lane - 3u * sub;  // lane 63 idles
    for (unsigned int h0 = 0; h0 < nheads; h0 += 21u) {
        const unsigned int hi = h0 + sub;
        const bool mine = lane < 63u && hi < nheads;
        const int src = (int)hl[mine ? hi : 0u];
        const int hx = __shfl(cx, src, 64), hy = __shfl(cy, src, 64), hz = __shfl(cz, src, 64);
        const unsigned int hc = (unsigned int)__shfl((int)c, src, 64);
        const unsigned int ha0 = (unsigned int)__shfl((int)a0, src, 64), ha1 = (unsigned int)__shfl((int)a1, src, 64);
        if (!mine) continue;
        // (+x: the next cell of the row, its bounds share the line; +y / +z: one row / one layer on)
        const bool ok = nb == 0 ? hx + 1 < g.dim[0] : (nb == 1 ? hy + 1 < g.dim[1] : hz + 1 < g.dim[2]);
        if (!ok) continue;
        const unsigned int cb = hc + (nb == 0 ? 1u : (nb == 1 ? (unsigned int)g.dim[0] : (unsigned int)g.dim[0] * (unsigned int)g.dim[1]));
        const unsigned int b0 = cs2[cb], b1 = cs2[cb + 1];
        if (b0 != b1 && ecc_cells_touch(cr2, ha0, ha1, b0, b1, r2)) uf_union(cparent, ha0, b0);
    }
}

// ---- phase 1 by x-RUNS (round 6; PCC_OPT_EC_CELLS = 3, the default; 4 = k_ecc_link_faces above) ---------------------------------
// k_ecc_link_faces offers one agent-scope union per occupied cell and face -- 3.6M of them at 5M points, for components that need
// one link per pair of neighbouring rows.  Inside an object the occupied cells of a row form RUNS of consecutive cells whose
// points touch: a run is found by its wave with a segmented scan over the wave's cell heads, and its cells get the run's first
// cell as their parent with a plain (agent-scope) STORE -- no compare-and-swap, no root walk: in this kernel nothing else writes a
// cell that is not the first of its run, and compare-and-swaps only ever land on first cells (a run that continues into the next
// wave is hooked there by the one union its last head makes).  runid[head] = first cell of the head's run within its wave.
__global__ void __launch_bounds__(256)
k_ecc_link_xruns(const float4* __restrict__ cr2, const unsigned int* __restrict__ cs2, const GridDev* __restrict__ gd2, float r2,
                 unsigned int* __restrict__ cparent, unsigned int* __restrict__ runid) {
    const GridParams g = gd2->g;
    const unsigned int n_valid = gd2->n_valid;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int lane = threadIdx.x & 63;
    if ((t & ~63u) >= n_valid) return;  // wave-uniform
    __shared__ unsigned int head_lane[4][64];
    unsigned int* hl = head_lane[threadIdx.x >> 6];
    int cx = 0;
    unsigned int c = 0, a0 = 0xffffffffu, a1 = 0;
    if (t < n_valid) {
        const float4 me = cr2[t];
        float ux, uy, uz;  // the point in the grid's frame (grid_device.hpp)
        grid_frame(g, me.x, me.y, me.z, ux, uy, uz);
        cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
        const int cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
        const int cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
        c = ((unsigned int)cz * g.dim[1] + cy) * g.dim[0] + cx;
        a0 = cs2[c];
        a1 = cs2[c + 1];
    }
    const bool head = t == a0;
    const unsigned long long heads = __ballot(head);
    const unsigned int nheads = (unsigned int)__popcll(heads);
    if (nheads == 0) return;  // (a wave inside one big cell)
    if (head) hl[__popcll(heads & ((1ull << lane) - 1ull))] = lane;
    wave_lds_sync();
    // lane k < nheads stands for the k-th head of the wave (cells in ascending order)
    const bool mine = lane < nheads;
    const int src = (int)hl[mine ? lane : 0u];
    const int hx = __shfl(cx, src, 64);
    const unsigned int hc = (unsigned int)__shfl((int)c, src, 64);
    const unsigned int ha0 = (unsigned int)__shfl((int)a0, src, 64), ha1 = (unsigned int)__shfl((int)a1, src, 64);
    // the next cell of the row: occupied when its range [ha1, b1) is not empty (CSR: it starts where this one ends)
    bool touch = false;
    unsigned int b1 = 0;
    if (mine && hx + 1 < g.dim[0]) {
        b1 = cs2[hc + 2];
        touch = b1 != ha1 && ecc_cells_touch(cr2, ha0, ha1, ha1, b1, r2);
    }
    // a head starts a run unless the head before it (in the wave) is its left neighbour and touches it
    const unsigned int prev_c = (unsigned int)__shfl_up((int)hc, 1, 64);
    const bool prev_touch = __shfl_up((int)touch, 1, 64) != 0;
    const bool start = mine && (lane == 0 || !(prev_touch && prev_c + 1u == hc));
    // (head positions ascend with the lane: the latest start at or before a lane is a running maximum)
    const unsigned int first = wave_incl_scan_max(start ? ha0 + 1u : 0u) - 1u;
    if (mine) {
        runid[ha0] = first;
        if (!start) __hip_atomic_store(&cparent[ha0], first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the run goes on in a later wave: its next cell is the first head there -- a run start, hooked with the one union
        if (lane == nheads - 1u && touch) uf_union(cparent, first, ha1);
    }
}

// the +y / +z face neighbours of every occupied cell, two lanes per head; of the cells of one run that touch the SAME run of the
// neighbouring row only the first offers the union (both runs are connected in themselves)
__global__ void __launch_bounds__(256)
k_ecc_link_yz(const float4* __restrict__ cr2, const unsigned int* __restrict__ cs2, const GridDev* __restrict__ gd2, float r2,
              unsigned int* __restrict__ cparent, const unsigned int* __restrict__ runid) {
    const GridParams g = gd2->g;
    const unsigned int n_valid = gd2->n_valid;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int lane = threadIdx.x & 63;
    if ((t & ~63u) >= n_valid) return;  // wave-uniform
    __shared__ unsigned int head_lane[4][64];
    unsigned int* hl = head_lane[threadIdx.x >> 6];
    int cy = 0, cz = 0;
    unsigned int c = 0, a0 = 0xffffffffu, a1 = 0;
    if (t < n_valid) {
        const float4 me = cr2[t];
        float ux, uy, uz;  // the point in the grid's frame (grid_device.hpp)
        grid_frame(g, me.x, me.y, me.z, ux, uy, uz);
        const int cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
        cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
        cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
        c = ((unsigned int)cz * g.dim[1] + cy) * g.dim[0] + cx;
        a0 = cs2[c];
        a1 = cs2[c + 1];
    }
    const bool head = t == a0;
    const unsigned long long heads = __ballot(head);
    const unsigned int nheads = (unsigned int)__popcll(heads);
    if (head) hl[__popcll(heads & ((1ull << lane) - 1ull))] = lane;
    wave_lds_sync();
    const unsigned int sub = lane >> 1, dir = lane & 1u;  // dir 0: one row on (+y), 1: one layer on (+z)
    for (unsigned int h0 = 0; h0 < nheads; h0 += 32u) {  // wave-uniform
        const unsigned int hi = h0 + sub;
        const bool mine = hi < nheads;
        const int src = (int)hl[mine ? hi : 0u];
        const int hy = __shfl(cy, src, 64), hz = __shfl(cz, src, 64);
        const unsigned int hc = (unsigned int)__shfl((int)c, src, 64);
        const unsigned int ha0 = (unsigned int)__shfl((int)a0, src, 64), ha1 = (unsigned int)__shfl((int)a1, src, 64);
        unsigned int my_run = 0xffffffffu, nrun = 0xffffffffu, b0 = 0;
        bool touch = false;
        if (mine && (dir == 0 ? hy + 1 < g.dim[1] : hz + 1 < g.dim[2])) {
            const unsigned int cb = hc + (dir == 0 ? (unsigned int)g.dim[0] : (unsigned int)g.dim[0] * (unsigned int)g.dim[1]);
            b0 = cs2[cb];
            const unsigned int b1 = cs2[cb + 1];
            if (b0 != b1 && ecc_cells_touch(cr2, ha0, ha1, b0, b1, r2)) {
                touch = true;
                my_run = runid[ha0];
                nrun = runid[b0];
            }
        }
        // the head before this one (same direction: two lanes back) linked the same two runs already
        const unsigned int p_my = (unsigned int)__shfl_up((int)my_run, 2, 64), p_n = (unsigned int)__shfl_up((int)nrun, 2, 64);
        const bool dup = lane >= 2u && p_my == my_run && p_n == nrun;
        if (touch && !dup) uf_union(cparent, ha0, b0);
    }
}

// parent[c] = root(c) for the heads of cells, when no union runs (plain loads: whatever a cache holds is an ancestor)
__global__ void __launch_bounds__(256)
k_ecc_flatten(const float4* __restrict__ cr2, const unsigned int* __restrict__ cs2, const GridDev* __restrict__ gd2,
              unsigned int* __restrict__ cparent) {
    const GridParams g = gd2->g;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= gd2->n_valid) return;
    const float4 me = cr2[t];
    if (t != cs2[cell_id(me, g)]) return;
    const unsigned int r = uf_find_settled(cparent, t);
    if (r != t) cparent[t] = r;
}

// phase 2: a wave takes the cells that start among its 64 points one at a time, lanes over the forward neighbours that are not
// face neighbours; "linked already" = equal flat roots (plain loads)
__global__ void __launch_bounds__(256)
k_ecc_link_rest(const float4* __restrict__ cr2, const unsigned int* __restrict__ cs2, const GridDev* __restrict__ gd2, float r2,
                unsigned int* __restrict__ cparent) {
    const GridParams g = gd2->g;
    const unsigned int n_valid = gd2->n_valid;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int lane = threadIdx.x & 63;
    if ((t & ~63u) >= n_valid) return;  // wave-uniform
    const bool have = t < n_valid;
    int cx = 0, cy = 0, cz = 0;
    unsigned int a0 = 0, a1 = 0, ra = 0;
    if (have) {
        const float4 me = cr2[t];
        float ux, uy, uz;  // the point in the grid's frame (grid_device.hpp)
        grid_frame(g, me.x, me.y, me.z, ux, uy, uz);
        cx = cell_coord(ux, g.org[0], g.inv_h, g.dim[0]);
        cy = cell_coord(uy, g.org[1], g.inv_h, g.dim[1]);
        cz = cell_coord(uz, g.org[2], g.inv_h, g.dim[2]);
        const unsigned int c = ((unsigned int)cz * g.dim[1] + cy) * g.dim[0] + cx;
        a0 = cs2[c];
        a1 = cs2[c + 1];
        ra = cparent[a0];  // the cell's flat root as of phase 1 (plain load)
    }
    // this lane's forward neighbour in the 5 x 5 x 5 numbering (62 = the own cell; forward = beyond it); lanes 0, 4 and 24
    // are +x, +y, +z: done in phase 1
    const int lin = (int)lane + 63;
    const int dz = lin / 25 - 2, dy = (lin / 5) % 5 - 2, dx = lin % 5 - 2;
    const bool mine = lane < 62 && lane != 0 && lane != 4 && lane != 24;
    unsigned long long heads = __ballot(have && t == a0);
    while (heads) {
        const int h = __builtin_ctzll(heads);
        heads &= heads - 1;
        const int hx = __builtin_amdgcn_readlane(cx, h), hy = __builtin_amdgcn_readlane(cy, h), hz = __builtin_amdgcn_readlane(cz, h);
        const unsigned int ha0 = (unsigned int)__builtin_amdgcn_readlane((int)a0, h);
        const unsigned int ha1 = (unsigned int)__builtin_amdgcn_readlane((int)a1, h);
        const unsigned int hra = (unsigned int)__builtin_amdgcn_readlane((int)ra, h);
        const int x = hx + dx, y = hy + dy, z = hz + dz;
        if (!mine || x < 0 || x >= g.dim[0] || y < 0 || y >= g.dim[1] || z >= g.dim[2]) continue;
        const unsigned int cb = ((unsigned int)z * g.dim[1] + y) * g.dim[0] + x;
        const unsigned int b0 = cs2[cb], b1 = cs2[cb + 1];
        if (b0 == b1) continue;
        if (cparent[b0] == hra) continue;               // same component after phase 1: two cached words
        if (uf_linked(cparent, ha0, b0)) continue;      // joined meanwhile (agent scope; the rare pairs that get here)
        if (ecc_cells_touch(cr2, ha0, ha1, b0, b1, r2)) uf_union(cparent, ha0, b0);
    }
}

// final flatten, component sizes and lowest member: one lane per point in grid order.  A wave's points belong to a handful of
// components: per distinct root one atomicAdd of the count and one atomicMin of the lowest original index (a lane-wise
// atomic per point onto a few hundred hot words is what cost the point-level form 2.1 ms)
__global__ void __launch_bounds__(256)
k_ecc_sizes(const float4* __restrict__ cr2, const unsigned int* __restrict__ cs2, const GridDev* __restrict__ gd2,
            unsigned int* __restrict__ cparent, unsigned int* __restrict__ size, unsigned int* __restrict__ minidx) {
    const GridParams g = gd2->g;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int lane = threadIdx.x & 63;
    if ((t & ~63u) >= gd2->n_valid) return;  // wave-uniform
    const bool have = t < gd2->n_valid;
    unsigned int r = 0xffffffffu, orig = 0xffffffffu;
    if (have) {
        const float4 me = cr2[t];
        orig = (unsigned int)__float_as_int(me.w);
        const unsigned int a0 = cs2[cell_id(me, g)];
        r = uf_find_settled(cparent, a0);
        if (t == a0 && r != a0) cparent[a0] = r;  // (the head writes the cell's flat root: k_ecc_label reads one word per point)
    }
    unsigned long long todo = __ballot(have);
    while (todo) {
        const int l = __builtin_ctzll(todo);
        const unsigned int rr = (unsigned int)__builtin_amdgcn_readlane((int)r, l);
        const unsigned long long same = __ballot(have && r == rr);
        todo &= ~same;
        unsigned int m = (have && r == rr) ? orig : 0xffffffffu;
        for (int off = 32; off > 0; off >>= 1) m = min(m, (unsigned int)__shfl_xor((int)m, off, 64));
        if ((int)lane == l) {
            atomicAdd(&size[rr], (unsigned int)__popcll(same));
            atomicMin(&minidx[rr], m);
        }
    }
}

// roots whose size passes the filter: (root, size, lowest member)
__global__ void __launch_bounds__(256)
k_ecc_collect(const unsigned int* __restrict__ cparent, const unsigned int* __restrict__ size, const unsigned int* __restrict__ minidx,
              unsigned int n, unsigned int min_size, unsigned int max_size, uint4* __restrict__ list, unsigned int* __restrict__ count,
              unsigned int cap) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned int sz = size[i];  // > 0 only at the roots of components (heads of cells that are their own parent)
        if (sz == 0 || cparent[i] != i) continue;
        if (sz < min_size || sz > max_size) continue;
        const unsigned int slot = atomicAdd(count, 1u);
        if (slot < cap) list[slot] = make_uint4(i, sz, minidx[i], 0u);
    }
}
__global__ void __launch_bounds__(256)
k_ecc_set_ids(const uint4* __restrict__ sorted_list, unsigned int ncl, int* __restrict__ id_of_root) {
    for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < ncl; k += gridDim.x * blockDim.x)
        id_of_root[sorted_list[k].x] = (int)k;
}
__global__ void __launch_bounds__(256)
k_ecc_label(const float4* __restrict__ cr2, const unsigned int* __restrict__ cs2, const GridDev* __restrict__ gd2,
            const unsigned int* __restrict__ cparent, const int* __restrict__ id_of_root, int32_t* __restrict__ labels) {
    const GridParams g = gd2->g;
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= gd2->n_valid) return;
    const float4 me = cr2[t];
    const unsigned int a0 = cs2[cell_id(me, g)];
    const int id = id_of_root[cparent[a0]];  // (flat since k_ecc_sizes)
    if (id >= 0) labels[(unsigned int)__float_as_int(me.w)] = id;
}

__global__ void __launch_bounds__(256)
k_uf_flatten_count(const float4* __restrict__ refs, unsigned int* __restrict__ parent, unsigned int n,
                   unsigned int* __restrict__ size) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (__float_as_int(refs[i].w) < 0) continue;  // non-finite point: in no cluster
        unsigned int r = uf_find(parent, i);
        atomicMin(&parent[i], r);  // other threads still reach the same root through older values
        atomicAdd(&size[r], 1u);
    }
}

// the same in the order of the clustering grid: neighbouring lanes are neighbouring points, whole runs of a
// wave share their root, and one atomicAdd per RUN replaces one per point (5M adds onto 256 hot words took
// 2.1 ms; a run is up to 64 points)
__global__ void __launch_bounds__(256)
k_uf_flatten_count_runs(const float4* __restrict__ cr2, const GridDev* __restrict__ gd2, unsigned int* __restrict__ parent,
                        unsigned int* __restrict__ size) {
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int lane = threadIdx.x & 63;
    const bool have = t < gd2->n_valid;
    unsigned int r = 0xffffffffu;
    if (have) {
        // (the links are done: a plain walk and a plain store of the root -- nobody else writes anything but ancestors, and
        // the root is the smallest of them)
        const unsigned int i = (unsigned int)__float_as_int(cr2[t].w);
        r = uf_find_settled(parent, i);
        parent[i] = r;
    }
    const unsigned int prev = __shfl_up(r, 1, 64);
    const bool head = have && (lane == 0 || prev != r);
    const unsigned long long heads = __ballot(head) | (__ballot(!have) & ~((__ballot(!have) << 1)));  // a run also ends where the points end
    if (head) {
        const unsigned long long after = lane == 63 ? 0ull : (heads >> (lane + 1));
        const unsigned int len = after ? (unsigned int)__ffsll((long long)after) : 64u - lane;
        atomicAdd(&size[r], len);
    }
}

__global__ void __launch_bounds__(256)
k_uf_collect(const unsigned int* __restrict__ parent, const unsigned int* __restrict__ size, unsigned int n,
             unsigned int min_size, unsigned int max_size, uint2* __restrict__ list, unsigned int* __restrict__ count,
             unsigned int cap) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (parent[i] != i) continue;
        unsigned int sz = size[i];  // 0 for a non-finite point (never counted)
        if (sz == 0) continue;
        if (sz < min_size || sz > max_size) continue;
        unsigned int slot = atomicAdd(count, 1u);
        if (slot < cap) list[slot] = make_uint2(i, sz);
    }
}

__global__ void __launch_bounds__(256)
k_uf_set_ids(const uint2* __restrict__ sorted_list, unsigned int ncl, int* __restrict__ id_of_root) {
    for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < ncl; k += gridDim.x * blockDim.x)
        id_of_root[sorted_list[k].x] = (int)k;
}

__global__ void __launch_bounds__(256)
k_uf_label(const float4* __restrict__ refs, const unsigned int* __restrict__ parent, const int* __restrict__ id_of_root,
           unsigned int n, int32_t* __restrict__ labels) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (__float_as_int(refs[i].w) >= 0) labels[i] = id_of_root[parent[i]];  // position == original index
}

static inline int g1(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// the union-find over cells (kernels k_ecc_*): the clustering grid is built (vox_a: points in its cell order, vox_c: CSR
// starts, vox_b: its GridDev)
static int grid_clusters_cells(pcc_index* ix, float r2, uint32_t min_size, uint32_t max_size, int32_t* labels_dev,
                               int32_t* n_clusters, int32_t* sizes, int max_sizes) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)ix->n_orig;
    const unsigned int cap = min_size > 0 ? n / min_size + 1 : n;
    PCC_TRY(ix->scratch_c.reserve((size_t)n * 4));
    PCC_TRY(ix->scratch_d.reserve((size_t)n * 4));
    PCC_TRY(ix->scratch_e.reserve((size_t)n * 4));
    PCC_TRY(ix->scratch_f.reserve((size_t)n * 4));
    PCC_TRY(ix->scratch_b.reserve((size_t)cap * sizeof(uint4) + 16));
    unsigned int* cparent = ix->scratch_c.as<unsigned int>();
    unsigned int* size = ix->scratch_d.as<unsigned int>();
    int* id_of_root = ix->scratch_e.as<int>();
    unsigned int* minidx = ix->scratch_f.as<unsigned int>();
    uint4* list = ix->scratch_b.as<uint4>();
    unsigned int* d_count = ix->small.as<unsigned int>() + 40;
    const float4* cr2 = ix->vox_a.as<float4>();
    const unsigned int* cs2 = ix->vox_c.as<unsigned int>();
    const GridDev* gd2 = ix->vox_b.as<GridDev>();
    const dim3 gp((n + 255) / 256), blk(256);
    ev_mark(ix, EV_MAIN0);
    PCC_HIP(hipMemsetAsync(d_count, 0, 4, s));
    hipLaunchKernelGGL(k_ecc_init, dim3(g1(n)), blk, 0, s, cparent, size, minidx, id_of_root, n, labels_dev, n);
    if (ix->opt.ec_cells == 4) {  // round 5: one union per occupied cell and face
        hipLaunchKernelGGL(k_ecc_link_faces, gp, blk, 0, s, cr2, cs2, gd2, r2, cparent);
    } else {  // x-runs by store, then one union per pair of neighbouring runs
        PCC_TRY(ix->scratch_g.reserve((size_t)n * 4 + 256));  // (the query sort's order buffer: no search is in flight on this handle)
        unsigned int* runid = ix->scratch_g.as<unsigned int>();
        hipLaunchKernelGGL(k_ecc_link_xruns, gp, blk, 0, s, cr2, cs2, gd2, r2, cparent, runid);
        hipLaunchKernelGGL(k_ecc_link_yz, gp, blk, 0, s, cr2, cs2, gd2, r2, cparent, (const unsigned int*)runid);
    }
    hipLaunchKernelGGL(k_ecc_flatten, gp, blk, 0, s, cr2, cs2, gd2, cparent);
    hipLaunchKernelGGL(k_ecc_link_rest, gp, blk, 0, s, cr2, cs2, gd2, r2, cparent);
    hipLaunchKernelGGL(k_ecc_sizes, gp, blk, 0, s, cr2, cs2, gd2, cparent, size, minidx);
    hipLaunchKernelGGL(k_ecc_collect, dim3(g1(n)), blk, 0, s, cparent, size, minidx, n, min_size, max_size, list, d_count, cap);
    PCC_HIP(hipGetLastError());
    unsigned int* h = static_cast<unsigned int*>(ix->pinned);
    PCC_HIP(hipMemcpyAsync(h, d_count, 4, hipMemcpyDeviceToHost, s));
    PCC_HIP(hipStreamSynchronize(s));
    const unsigned int ncl = h[0];
    if (ncl > cap) { set_error("cluster list overflow (%u > %u)", ncl, cap); return PCC_ERR_OVERFLOW; }
    std::vector<uint4> host_list(ncl);
    if (ncl) {
        PCC_HIP(hipMemcpyAsync(host_list.data(), list, (size_t)ncl * sizeof(uint4), hipMemcpyDeviceToHost, s));
        PCC_HIP(hipStreamSynchronize(s));
        // PCL: std::sort(clusters.rbegin(), clusters.rend(), by size) -> largest first; equal sizes are left unspecified there,
        // here: lowest member index first (.z)
        std::sort(host_list.begin(), host_list.end(), [](const uint4& a, const uint4& b) {
            if (a.y != b.y) return a.y > b.y;
            return a.z < b.z;
        });
        PCC_HIP(hipMemcpyAsync(list, host_list.data(), (size_t)ncl * sizeof(uint4), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_ecc_set_ids, dim3(g1(ncl)), blk, 0, s, list, ncl, id_of_root);
        hipLaunchKernelGGL(k_ecc_label, gp, blk, 0, s, cr2, cs2, gd2, cparent, id_of_root, labels_dev);
        PCC_HIP(hipGetLastError());
    }
    ev_mark(ix, EV_MAIN1);
    PCC_HIP(hipStreamSynchronize(s));  // host_list must outlive the H2D copy
    if (n_clusters) *n_clusters = (int32_t)ncl;
    if (sizes)
        for (unsigned int k = 0; k < ncl && (int)k < max_sizes; ++k) sizes[k] = (int32_t)host_list[k].y;
    return PCC_OK;
}

int grid_clusters(pcc_index* ix, float r, float r2, uint32_t min_size, uint32_t max_size,
                  int32_t* labels_dev, int32_t* n_clusters, int32_t* sizes, int max_sizes) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)ix->n_orig;
    // (built first: the cell sort uses the scratch buffers the union-find arrays live in afterwards)
    // clustering grid: cell edge 0.57 r (diagonal 0.987 r < r), over the bounding box of the valid points
    bool cells_ok = false;
    const bool no_cells = ix->opt.ec_cells == 0;
    if (!no_cells && r > 0.f && n >= 4096) {
        PCC_TRY(sync_info(ix));
        GridDev hd;
        memset(&hd, 0, sizeof(hd));
        const float h = r * 0.57f;
        double cells = 1;
        // NOT the search grid's layout rule.  The workgroups in flight at any moment own a contiguous stretch of the cell order --
        // a slab of a few layers --, and every union / size update of theirs is an atomic on the ROOT word of a component: a slab
        // should cut through as many components as the scene has, i.e. be as THIN as possible -- the SHORTEST extent over the
        // layers (objects stand side by side on a floor).  With the search grid's rule (the longest extent over the layers) a slab
        // of the 5M-point object scene held the cells of ~10 of its 256 balls and the union-find kernels took 2.1-2.25 ms instead
        // of 1.05-1.08 (profiles/r06_exp_axis_order.txt).  Rows along the second shortest extent, as in the search grid.
        float ext3[3];
        for (int a = 0; a < 3; ++a) ext3[a] = ix->bbox_hi[a] - ix->bbox_lo[a];
        grid_axes_for(ext3, ix->opt.grid_axes, hd.g.ax);
        if (ix->opt.grid_axes < 0) {  // by extent: (second shortest, shortest, longest) -> (second shortest, longest, shortest)
            const int shortest = hd.g.ax[1], longest = hd.g.ax[2];
            hd.g.ax[1] = longest; hd.g.ax[2] = shortest;
        }
        for (int a = 0; a < 3; ++a) {
            const int c = hd.g.ax[a];
            const double ext = (double)ix->bbox_hi[c] - (double)ix->bbox_lo[c];
            hd.g.org[a] = ix->bbox_lo[c];
            hd.g.dim[a] = (int)std::min(ext / h + 2.0, 2.0e9);
            cells *= (double)hd.g.dim[a];
            hd.lo[c] = hd.glo[a] = ix->bbox_lo[c];
            hd.hi[c] = hd.ghi[a] = ix->bbox_hi[c];
        }
        if (ix->n_valid > 0 && cells <= (double)(1u << 26)) {
            hd.g.h = h;
            hd.g.inv_h = 1.0f / h;
            hd.g.ncells = hd.g.dim[0] * hd.g.dim[1] * hd.g.dim[2];
            hd.n_valid = (unsigned int)ix->n_valid;
            PCC_TRY(ix->vox_a.reserve((size_t)n * sizeof(float4) + 64));
            PCC_TRY(ix->vox_b.reserve(sizeof(GridDev)));
            PCC_TRY(ix->vox_c.reserve(((size_t)hd.g.ncells + 8) * sizeof(unsigned int)));
            PCC_HIP(hipMemcpyAsync(ix->vox_b.p, &hd, sizeof(hd), hipMemcpyHostToDevice, s));
            PCC_TRY(cell_sort(ix, ix->refs.as<float4>(), n, true, ix->vox_a.as<float4>(), nullptr, ix->vox_c.as<unsigned int>(),
                              nullptr, ix->vox_b.as<GridDev>(), (unsigned int)hd.g.ncells));
            PCC_HIP(hipStreamSynchronize(s));  // hd lives on this stack frame
            cells_ok = true;
        }
    }
    if (cells_ok && ix->opt.ec_cells >= 3)
        return grid_clusters_cells(ix, r2, min_size, max_size, labels_dev, n_clusters, sizes, max_sizes);
    // scratch: parent[n] | size[n] | id_of_root[n] | list[cap]
    const unsigned int cap = min_size > 0 ? n / min_size + 1 : n;
    PCC_TRY(ix->scratch_c.reserve((size_t)n * 4));
    PCC_TRY(ix->scratch_d.reserve((size_t)n * 4));
    PCC_TRY(ix->scratch_e.reserve((size_t)n * 4));
    PCC_TRY(ix->scratch_b.reserve((size_t)cap * sizeof(uint2) + 16));
    unsigned int* parent = ix->scratch_c.as<unsigned int>();
    unsigned int* size = ix->scratch_d.as<unsigned int>();
    int* id_of_root = ix->scratch_e.as<int>();
    uint2* list = ix->scratch_b.as<uint2>();
    unsigned int* d_count = ix->small.as<unsigned int>() + 40;
    ev_mark(ix, EV_MAIN0);
    hipLaunchKernelGGL(k_uf_init, dim3(g1(n)), dim3(256), 0, s, parent, n);
    PCC_HIP(hipMemsetAsync(size, 0, (size_t)n * 4, s));
    PCC_HIP(hipMemsetAsync(id_of_root, 0xff, (size_t)n * 4, s));
    PCC_HIP(hipMemsetAsync(d_count, 0, 4, s));
    PCC_HIP(hipMemsetAsync(labels_dev, 0xff, ix->n_orig * sizeof(int32_t), s));
    if (cells_ok) {
        if (ix->opt.ec_cells == 2)  // (the lane-per-point form, kept for comparison)
            hipLaunchKernelGGL(k_uf_link_cells, dim3((n + 255) / 256), dim3(256), 0, s, ix->vox_a.as<float4>(),
                               ix->vox_c.as<unsigned int>(), ix->vox_b.as<GridDev>(), r2, parent);
        else
            hipLaunchKernelGGL(k_uf_link_cells_wave, dim3((n + 255) / 256), dim3(256), 0, s, ix->vox_a.as<float4>(),
                               ix->vox_c.as<unsigned int>(), ix->vox_b.as<GridDev>(), r2, parent);
    } else {
        hipLaunchKernelGGL(k_uf_link, dim3((n + 255) / 256), dim3(256), 0, s, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), r, r2, parent);
    }
    if (cells_ok)
        hipLaunchKernelGGL(k_uf_flatten_count_runs, dim3((n + 255) / 256), dim3(256), 0, s, ix->vox_a.as<float4>(),
                           ix->vox_b.as<GridDev>(), parent, size);
    else
        hipLaunchKernelGGL(k_uf_flatten_count, dim3(g1(n)), dim3(256), 0, s, ix->refs.as<float4>(), parent, n, size);
    hipLaunchKernelGGL(k_uf_collect, dim3(g1(n)), dim3(256), 0, s, parent, size, n, min_size, max_size, list, d_count, cap);
    PCC_HIP(hipGetLastError());
    unsigned int* h = static_cast<unsigned int*>(ix->pinned);
    PCC_HIP(hipMemcpyAsync(h, d_count, 4, hipMemcpyDeviceToHost, s));
    PCC_HIP(hipStreamSynchronize(s));
    unsigned int ncl = h[0];
    if (ncl > cap) { set_error("cluster list overflow (%u > %u)", ncl, cap); return PCC_ERR_OVERFLOW; }
    std::vector<uint2> host_list(ncl);
    if (ncl) {
        PCC_HIP(hipMemcpyAsync(host_list.data(), list, (size_t)ncl * sizeof(uint2), hipMemcpyDeviceToHost, s));
        PCC_HIP(hipStreamSynchronize(s));
        // PCL: std::sort(clusters.rbegin(), clusters.rend(), by size) -> largest first; equal sizes are
        // left unspecified there, here: lowest member index first (root == lowest member)
        std::sort(host_list.begin(), host_list.end(), [](const uint2& a, const uint2& b) {
            if (a.y != b.y) return a.y > b.y;
            return a.x < b.x;
        });
        PCC_HIP(hipMemcpyAsync(list, host_list.data(), (size_t)ncl * sizeof(uint2), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_uf_set_ids, dim3(g1(ncl)), dim3(256), 0, s, list, ncl, id_of_root);
    }
    hipLaunchKernelGGL(k_uf_label, dim3(g1(n)), dim3(256), 0, s, ix->refs.as<float4>(), parent, id_of_root, n, labels_dev);
    PCC_HIP(hipGetLastError());
    ev_mark(ix, EV_MAIN1);
    PCC_HIP(hipStreamSynchronize(s));  // host_list must outlive the H2D copy
    if (n_clusters) *n_clusters = (int32_t)ncl;
    if (sizes)
        for (unsigned int k = 0; k < ncl && (int)k < max_sizes; ++k) sizes[k] = (int32_t)host_list[k].y;
    return PCC_OK;
}

PCC_PAIRS_TAKE(cluster)

}  // namespace pcc
