// pcc_internal.hpp -- shared declarations of libpcc_nn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include <string>
#include "pcc_nn.h"
#include "flann_tree.hpp"

// every PCC_SEED_STRIDE-th reference is a "seed": the exhaustive scan of the seeds bounds a far query's ball
#define PCC_SEED_SHIFT 6
#define PCC_SEED_STRIDE (1 << PCC_SEED_SHIFT)
// pcc_index::small (uint32 words): [32] fallback count, [33] far-list count, [52] ticket of the build's pack kernel (grid.hip), [53] ticket of k_icp_sums (fused solve), [PCC_OPEN_CTR0 + s * PCC_OPEN_CTR_STRIDE] open-lane
// count of shard s -- one 128-byte line each, PCC_OPEN_SHARDS of them (a single word takes ~88 atomics per microsecond)
#define PCC_OPEN_SHARDS 64
#define PCC_OPEN_CTR0 1024
#define PCC_OPEN_CTR_STRIDE 32
// [PCC_TIE_CTR0 + s * PCC_OPEN_CTR_STRIDE + {0, 1}] queries listed as tied / indices changed, per slice of the tie list (PCC_TIES_FLANN)
#define PCC_TIE_SHARDS 16
#define PCC_TIE_CTR0 (PCC_OPEN_CTR0 + PCC_OPEN_SHARDS * PCC_OPEN_CTR_STRIDE)
#define PCC_SMALL_BYTES ((PCC_TIE_CTR0 + PCC_TIE_SHARDS * PCC_OPEN_CTR_STRIDE) * 4)
#define FLANN_DEV_STACK_MAX 256  // deepest tree k_tie_walk takes (20 bytes of scratch per level and lane)

namespace pcc {

// ---- error plumbing --------------------------------------------------------
void set_error(const char* fmt, ...);
#define PCC_HIP(expr)                                                            \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) {                                                  \
            pcc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                  \
            return PCC_ERR_DEVICE;                                               \
        }                                                                        \
    } while (0)
#define PCC_TRY(expr)                 \
    do {                              \
        int _s = (expr);              \
        if (_s != PCC_OK) return _s;  \
    } while (0)

// ---- device buffer that only grows ------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);  // returns pcc_status
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// grow-only pinned host buffer (read-backs that the host then walks: SOR mean distances, ...)
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Uniform-grid parameters (device + host copy).  Cell of a point along GRID axis a (0: the cells of a row, fastest in the
// linear id (c2 * dim1 + c1) * dim0 + c0; 1: the rows of a layer; 2: the layers):
//   c_a = clamp(int((p[ax_a] - org_a) * inv_h), 0, dim_a - 1)
// The same expression (same rounding) is used at build and query time.  ax[] names the coordinate of the cloud (0 x, 1 y, 2 z)
// each grid axis follows: k_grid_params puts the cloud's SHORTEST extent on axis 1 and the longest on axis 2, so that the rows
// above / below and the layers before / behind a query's row lie as close to it in memory as the cloud allows.
struct GridParams {
    float org[3];  // grid axes
    float h;
    float inv_h;
    int dim[3];    // grid axes
    int ncells;
    int ax[3];     // grid axis -> coordinate of the cloud
};


// The rule behind GridParams::ax (one place: k_grid_params on the device, the clustering grid on the host): the coordinate
// with the second shortest extent runs along the rows, the shortest over the rows of a layer, the longest over the layers;
// ties keep x, y, z order.  forced 0..5 = xyz, xzy, yxz, yzx, zxy, zyx (PCC_OPT_GRID_AXES), anything else: by extent.
// GRID_AXES_MIN_POINTS: clouds below it would keep x, y, z.  Measured on ONE handle (same memory for every setting, tools/exp_ab.py,
// exp_knn_ab.py): for k = 1 the layout by extent costs 3-5 % of a step at 1M-1.4M points (C2 0.2263 -> 0.2327 ms, room scan 0.3601 ->
// 0.3787), gains 1.4-3 % at 4M and is time-neutral at 10M (1.3333 / 1.3334 ms), where it takes a sixth of the search's fabric traffic away;
// for k-NN at 1M it is worth 7 % at K = 51 and 17 % at K = 100 on the corridor scene (1.229 -> 1.143, 2.196 -> 1.814 ms: the ball boxes of the
// bound path span rows of several layers) and nothing on the room scan.  A cloud of the reference's sizes spends 5-20x longer in its k-NN
// consumers (SOR, normals, region growing) than in a k = 1 step: 0 -- by extent at every size.
constexpr unsigned int GRID_AXES_MIN_POINTS = 0u;
__host__ __device__ inline void grid_axes_for(const float ext[3], int forced, int ax[3]) {
    // (no array is indexed by a variable: on the device that would put it in scratch memory)
    const float e0 = ext[0], e1 = ext[1], e2 = ext[2];
    // rank of each coordinate by ascending extent, ties in x, y, z order (a stable sort of three)
    const int r0 = (e1 < e0 ? 1 : 0) + (e2 < e0 ? 1 : 0);
    const int r1 = (e0 <= e1 ? 1 : 0) + (e2 < e1 ? 1 : 0);
    const int r2 = (e0 <= e2 ? 1 : 0) + (e1 <= e2 ? 1 : 0);
    const int shortest = r0 == 0 ? 0 : (r1 == 0 ? 1 : 2), second = r0 == 1 ? 0 : (r1 == 1 ? 1 : 2), longest = r0 == 2 ? 0 : (r1 == 2 ? 1 : 2);
    (void)r2;
    ax[0] = second; ax[1] = shortest; ax[2] = longest;
    if (forced >= 0 && forced < 6) {
        ax[0] = forced >> 1;                          // 0 0 1 1 2 2
        const int o0 = ax[0] == 0 ? 1 : 0, o1 = ax[0] == 2 ? 1 : 2;  // the other two, ascending
        ax[1] = (forced & 1) ? o1 : o0;
        ax[2] = (forced & 1) ? o0 : o1;
    }
}
// element `i` of a three-vector without indexing memory by a variable
template <class T>
__host__ __device__ inline T pick3(const T v[3], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }

// Device-resident description of the grid, written by k_grid_params from the pack kernel's
// per-workgroup statistics.  The host never waits for it on the build path: launches are sized
// from upper bounds it knows (n, nc_cap) and kernels read the actual values from here.
struct GridDev {
    GridParams g;
    float slack;            // absolute slack of the outside-of-cube bound (cell-boundary rounding)
    unsigned int n_valid;   // finite points (PCL total_nr_points_)
    unsigned int n_invalid;
    float lo[3], hi[3];     // bounding box of the valid points (x, y, z: what the handle reports)
    float glo[3], ghi[3];   // the same box along the grid's axes (ball_box, k_grid_far)
    unsigned int voxel;     // 1: cells are PCL VoxelGrid voxels -- id from floor(v*inv_h) - org (org = float(min_b))
};

// Tuning knobs of one handle (pcc_index_set_option).  The PCC_* environment variables only supply the defaults a new
// handle starts with; nothing in the library reads the environment after that.
struct Options {
    double grid_ppc = 0.75;         // PCC_OPT_GRID_PPC: mean references per cell the grid aims for (0.5 was the optimum of the lane-per-query
                                    // kernel; with the flat kernels 0.7-1.0 is a plateau for volumetric scenes, surface scans of 10M points
                                    // prefer 0.25-0.5, those of the reference's sizes 1.0: DESIGN.md 5)
    int grid_trim = 3;              // PCC_OPT_GRID_TRIM: k of the trimmed bounding box (0: plain bounding box)
    int far_mode = -1;              // PCC_OPT_FAR_MODE: -1 auto, 0 exhaustive fallback only, 1 always seed scan + ball walk
    int icp_sorted = 1;             // PCC_OPT_ICP_SORTED: pcc_icp_align keeps its source cloud in the target grid's cell order (one gather, then
                                    // every pass reads queries and writes keys front to back); 0 = caller's order, gathered / scattered per pass
    int icp_warm = 1;               // PCC_OPT_ICP_WARM: ICP passes start from the previous pass's neighbours
    int icp_device_loop = 1;        // PCC_OPT_ICP_DEVICE_LOOP: 0 = the host-driven loop (same bits)
    int ec_cells = 3;               // PCC_OPT_EC_CELLS: clustering on the clique-cell grid: 3 union-find over CELLS (round 5), 1 / 2 over points
                                    // (lanes over neighbour cells / over points); 0: per-point ball scan on the search grid
    double sort_mp_min = 1.8e6;     // PCC_OPT_SORT_MP_MIN: references from which the three-level sort is used (round 5, build us two-level /
                                    // three-level: 1M 89 / 124, 1.5M 122 / 135, 2M 155 / 149)
    double sort_mp_min_q = 2.5e6;   // PCC_OPT_SORT_MP_MIN_Q: the same for query clouds (5M until round 5: with the cells from the pack kernel and the
                                    // larger level-3 stage the three-level form wins from ~2.5M on -- 3M 117 -> 88 us, 4M 131 -> 98 us, 2M 75 = 77)
    int nn1_kernel = 1;             // PCC_OPT_NN1_KERNEL: 0 one lane per query; 1 rows drained flat, lanes over candidates (2 / 3: open lanes listed / in place)
    int knn_cache_k = 0;            // PCC_OPT_KNN_CACHE_K: self k-NN rows searched with at least this K and kept (0: off)
    int knn_kernel = 1;             // PCC_OPT_KNN_KERNEL: 1 selection kernel for k <= 128, 0 merge network only
    int nn1_dense_min = 4;          // PCC_OPT_NN1_DENSE_MIN: references per own cell from which a wave starts with the own cell alone
    int sort_stage1 = 1;            // PCC_OPT_SORT_STAGE1: level 1 of the three-level sort writes bucket-sorted LDS tiles (1: reference points; 2: query pairs too; 0: one store per point)
    int nn1_open_flat = 1;          // PCC_OPT_NN1_OPEN_FLAT: the listed open lanes drained flat (k_nn1_open_flat); 0 = one lane per query
    int flann_split = 0;            // PCC_OPT_FLANN_SPLIT: 0 middleSplit_, 1 middleSplit (which rule FLANN's divideTree is replayed with)
    int grid_axes = -1;             // PCC_OPT_GRID_AXES: which coordinate the grid's axes (row, rows of a layer, layers) follow: -1 by extent (second
                                    // shortest, shortest, longest; clouds below GRID_AXES_MIN_POINTS -- 0 -- would keep xyz); -2 by extent whatever that says;
                                    // 0 xyz (the layout of rounds 1-5), 1 xzy, 2 yxz, 3 yzx, 4 zxy, 5 zyx
    int knn_run = 16;               // PCC_OPT_KNN_RUN: k-NN selection kernel: consecutive cell-sorted queries a wave takes in a row, every one after the
                                    // first starting from its predecessor's K-th distance + their separation (knn.hip); 1 = every query on its own.
                                    // One handle, 1M self query, runs of 4 / 8 / 16 / 32, ms: corridor K = 51 1.167 / 1.146 / 1.119 / 1.215, K = 100
                                    // 1.910 / 1.816 / 1.775 / 2.013; room scan K = 51 0.976 / 0.971 / 0.946 / 0.991, K = 100 1.210 / 1.179 / 1.148 / 1.212
    int scan_chained = 1;           // PCC_OPT_SCAN_CHAINED: exclusive scans of up to 512 x 2048 counters in ONE launch (workgroups pass their totals on as
                                    // tagged 64-bit atomics and wait for the workgroups in front of them: relies on in-order dispatch); 0 = the two-launch
                                    // form (block totals, then apply), which waits for nothing
    int host_pipe = 1;              // PCC_OPT_HOST_PIPE: clouds / results of 8 MB and more in pageable HOST memory cross PCIe through the library's own pinned
                                    // chunk buffers, staged by a few host threads (x, y, z only when the stride is 24 bytes or more); 0 = one
                                    // hipMemcpyAsync of the raw array (rounds 1-5)
    int fuse_params = 0;            // PCC_OPT_FUSE_PARAMS (bits): 1 = the grid parameters come out of the build's pack kernel (its last workgroup to finish)
                                    // instead of k_grid_params -- built in round 6 and a LOSS: the hand-over needs an agent-scope release fence in every
                                    // workgroup, which on this chip writes the XCD's L2 back, in a kernel that has just dirtied 160 MB of it: build
                                    // 0.437 -> 0.497 ms at C3, 0.089 -> 0.112 at C2 (EXPERIMENTS.md); 2 = a pass of pcc_icp_align is solved by the last
                                    // workgroup of k_icp_sums (a kernel that writes 65 KB) instead of k_icp_solve
    int xcd_run = 256;              // PCC_OPT_XCD_RUN: consecutive workgroups of the k = 1 search steered to the same XCD (its L2).  32 until round 5;
                                    // with the layers of the grid a few rows apart (grid_axes) a run should hold several LAYERS, so that the rows of
                                    // the layer behind are re-read from the same L2: C3 fabric traffic of the search 2.95 -> 2.42 GB per call.  Time,
                                    // on one handle: C3 step 1.3517 (by extent, 32) -> 1.3334 (by extent, 256); x / y / z layout 1.3333 / 1.3316
    int overlap_prep = 1;           // PCC_OPT_OVERLAP_PREP: a k = 1 search that follows setInputCloud directly packs and sorts its queries on a
                                    // second stream while the build's cell sort is still running (they share nothing but the grid parameters);
                                    // from 2M queries on, 2 = at every size
    void from_env();
};

}  // namespace pcc

namespace pcc { struct HostPipe; }  // host_pipe.hpp (api.hip): pipelined transfers between pageable host memory and the device
#define PCC_EV_SLOTS 64
#define PCC_EV_KINDS 10
// The opaque handle of the C-ABI.
struct pcc_index {
    std::mutex mu;                     // every entry point holds it: calls on ONE handle from several threads are serialised
    pcc::Options opt;                  // pcc_index_set_option
    int device = 0;
    hipStream_t stream = nullptr;      // stream in use
    hipStream_t own_stream = nullptr;  // library-owned stream
    hipEvent_t edge_ev = nullptr;      // pcc_index_wait_stream / pcc_stream_wait_index
    // Query staging beside the build (PCC_OPT_OVERLAP_PREP, api.hip: PrepOverlap).  params_ev is recorded behind k_grid_params of
    // every build; a search that is the NEXT call on the handle (build_fresh) sends its pack + query sort to side_stream behind
    // that event -- not behind the build's sort -- and the main stream picks the result up through side_ev.  The sort's scratch is
    // swapped for the `side` set meanwhile, so the two sorts share no buffer.
    hipStream_t side_stream = nullptr;
    hipEvent_t params_ev = nullptr, side_ev = nullptr;
    bool params_ev_set = false;        // params_ev was recorded by the build the handle currently holds, on the stream in use
    bool build_fresh = false;          // nothing has been enqueued on the handle since that build
    bool after_build = false;          // build_fresh as the entry point in progress found it
    bool edge_fresh = false;           // one pcc_index_wait_stream came between the build and now: side_stream waits for edge_ev too
    struct SideScratch {
        pcc::DevBuf a, b, c, e, mp_a, mp_b, mp_c, scan_flags;
        unsigned int scan_epoch = 0;
    } side;
    unsigned int* pre_order = nullptr;    // a query order prepared ahead of grid_nn1 (this call only): order, count word, queries
    unsigned int* pre_nsorted = nullptr;
    size_t pre_order_nq = 0;
    size_t n_orig = 0;                 // points handed to pcc_index_create / set_input
    size_t n_valid = 0;                // finite points (PCL total_nr_points_); valid after sync_info()
    unsigned int nc_cap = 0;           // upper bound of the grid's cell count the host sizes launches with
    pcc::DevBuf d_grid;                // GridDev on the device
    pcc::GridDev* h_grid = nullptr;    // pinned host mirror, filled asynchronously
    bool stats_pending = false;        // fallback counter of the last GRID search in flight to pinned[40]
    size_t last_nq = 0;
    bool info_pending = false;         // h_grid copy in flight: sync before reading n_valid / grid / bbox
    int engine = PCC_ENGINE_BRUTE;     // resolved engine
    int engine_requested = PCC_ENGINE_AUTO;
    float bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};  // of the valid points
    // references in ORIGINAL order: (x, y, z, bits(index)), n_orig entries; non-finite points stay
    // in place flagged w = -1 (every kernel skips them), so position == original index
    pcc::DevBuf refs;
    // GRID engine: references permuted into cell order + CSR cell starts
    bool has_grid = false;
    pcc::GridParams grid{};
    pcc::DevBuf cell_refs;   // float4[n_valid], cell-sorted, .w = orig index
    pcc::DevBuf cell_start;  // uint32[ncells + 1]
    pcc::DevBuf seeds;       // float4[ceil(n / PCC_SEED_STRIDE)]: every 64th reference (w = its position; strides 32 / 128 / 256 measured 38.3 / 40.0 / 44.2 ms vs 38.0 on the ICP config) -- upper bounds for far queries
    bool fb_zeroed = false;  // the query pack kernel of this call already zeroed the fallback counter
    // ICP moves the same source cloud rigidly from pass to pass: the lane order of its first pass keeps
    // neighbouring lanes on neighbouring points, so later passes skip the query sort
    bool keep_order = false, order_valid = false;
    const float* pre_transform = nullptr;  // ICP loop in cell order: the 3 x 4 matrix (device memory) the next k = 1 search applies to its
                                           // queries, and writes back, before it looks -- the pass's k_transform folded into the search
    bool warm_start = false;  // ICP passes after the first: out_packed holds the previous pass's keys (grid_nn1 starts from them)
    size_t order_nq = 0;
    unsigned int* order_ptr = nullptr;
    unsigned int* order_nsorted = nullptr;
    unsigned int last_fallback_seen = 0;  // fallback count of an earlier search (heuristic only, may be stale)
    // scratch (grow-only, reused across calls on the index's stream)
    pcc::DevBuf q_raw, q_packed, out_packed, out_idx, out_d2, scratch_a, scratch_b,
        scratch_c, scratch_d, scratch_e, scratch_f, scratch_g, small, blk_stats, icp_src, icp_state, vox_a, vox_b, vox_c,
        mp_a, mp_b, mp_c,  // cellsort_mp.hip: two intermediate point buffers, bucket counters
        q_cells,     // grid cell of every staged query (k_pack), read by level 1 of the three-level sort instead of the points
        scan_flags,  // k_scan_chained: one tagged total per workgroup (pack.hip); nothing else ever writes here
        rows_idx, rows_d2;  // pcc_radius_fill_max: the k-NN rows it cuts at the radius (no scratch the query sort touches)
    // PCC_TIES_FLANN (flann_tree.hpp): kd-tree of FLANN's shape, built on the host at the first search after every
    // set_input, walked on the device (flann_order.hip)
    int tie_mode = PCC_TIES_LOWEST_INDEX;
    pcc::FlannTree flann;
    bool flann_valid = false;
    size_t small_raw_n = 0, small_raw_stride = 0;  // the indexed cloud's raw records are in the pinned small-call buffer (slot 0): n, stride; 0 = not
    int self_rows_k = 0;      // self_rows holds the self k-NN rows of the indexed cloud with this many neighbours (0: nothing kept)
    pcc::DevBuf self_rows;
    bool occ_valid = false;   // occ (device word): number of non-empty cells of the current grid, counted at the first radius count
    pcc::DevBuf occ;
    pcc::DevBuf tie_buf, flann_nodes, flann_leaf;
    pcc::DevBuf knn_fb;  // a list of up to n indices + its count, one user at a time: queries the k-NN selection kernel hands back,
                         // rows a fused radius fill leaves to k_sort_rows, region growing's points with a cross edge
    size_t q_cells_n = 0;                         // q_cells describes q_packed[0 .. q_cells_n) as staged (0: stale -- consumed, or q_packed has moved since)
    unsigned int scan_epoch = 0;                  // tag of the last chained scan on this handle
    bool sor_exact_last = true;                   // the last pcc_sor took its sums on the device (no addition of PCL's order rounds)
    bool open_pending = false;                    // the open-lane counters of the last listed k = 1 search are still on the device
    bool ties_pending = false;                    // the tie counters of the last search are still on the device
    uint64_t ties_flagged = 0, ties_changed = 0;  // of the last search in FLANN mode
    void* pinned = nullptr;  // small pinned host block for scalar read-backs
    pcc::HostBuf host_a, host_b;  // large pinned read-back buffers
    pcc::HostBuf host_c;          // pinned staging of the FLANN tree a small call builds (flann_order.hip)
    pcc::HostPipe* pipe = nullptr;  // two pinned chunk buffers + events, made at the first large host transfer (api.hip)
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // HIP-event instrumentation (pcc_index_enable_timing): event pairs on the index's stream
    // ring of PCC_EV_SLOTS calls so a timed region of many steps is covered without syncing
    int timing = 0;  // 0 off, 1 main kernel only (2 events per call), 2 full breakdown
    hipEvent_t ev[PCC_EV_SLOTS][PCC_EV_KINDS] = {};
    bool ev_rec[PCC_EV_SLOTS][PCC_EV_KINDS] = {};
    unsigned int ev_slot = 0;  // slot of the call in flight
};

namespace pcc {

// A search key is (d2 bits << 32 | position); ~0 means "nothing found".  FLANN starts every result set with a worst
// distance of FLT_MAX and rejects dist >= worst (KNNSimpleResultSet), so a candidate whose squared distance
// overflowed -- coordinates around 1e19 -- is no neighbour either: every consumer of a key treats d2 >= FLT_MAX like ~0.
// (The searches themselves keep such candidates: they only matter when nothing finite exists.)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool key_none(unsigned long long key) { return (unsigned int)(key >> 32) >= 0x7f7fffffu; }

enum { EV_MAIN0 = 0, EV_MAIN1, EV_FB0, EV_FB1, EV_CALL0, EV_CALL1, EV_BUILD0, EV_BUILD1, EV_SORT0, EV_SORT1 };
// every entry point that may enqueue work passes here once it holds the handle's mutex
inline void entered(pcc_index* ix) {
    ix->after_build = ix->build_fresh;
    ix->build_fresh = false;
    ix->pre_order = nullptr;  // (an order prepared by a call that failed before using it)
}
inline void ev_mark(pcc_index* ix, int id) {
    if (!ix->timing || (ix->timing == 1 && id > EV_MAIN1)) return;
    unsigned int s = ix->ev_slot % PCC_EV_SLOTS;
    if (ix->ev[s][id]) { (void)hipEventRecord(ix->ev[s][id], ix->stream); ix->ev_rec[s][id] = true; }
}
// a new instrumented call begins: advance the ring and forget what the slot held
inline void ev_next(pcc_index* ix) {
    if (!ix->timing) return;
    ++ix->ev_slot;
    unsigned int s = ix->ev_slot % PCC_EV_SLOTS;
    for (int k = 0; k < PCC_EV_KINDS; ++k) ix->ev_rec[s][k] = false;
}

// ---- kernels / launchers (pack.hip) -------------------------------------------
// AoS (stride bytes, 3 floats at offset 0) -> float4(x,y,z,bits(i)); non-finite points are
// written with w = -1.  With blk_stats != nullptr every workgroup b also writes 8 floats:
// [0] bits(invalid count), [1..3] min xyz, [4..6] max xyz of its valid points; *n_blocks rows.
constexpr int PACK_MAX_BLOCKS = 1024;
// grid (index builds, with blk_stats): the pack kernel's last workgroup also derives the index's grid (what k_grid_params does in
// a launch of its own): ticket = a zeroed device word, the rest are k_grid_params' arguments
struct PackGrid {
    unsigned int* ticket;
    float ppc;
    unsigned int nc_cap;
    int trim_k, axes;
    GridDev* out;
    GridDev* host_mirror;
};
int launch_pack(hipStream_t s, const void* aos, size_t n, size_t stride, float4* out,
                float* blk_stats = nullptr, int* n_blocks = nullptr, unsigned int* zero_word = nullptr, float4* seeds = nullptr,
                unsigned long long* invalid_keys = nullptr, unsigned int* cells = nullptr, const GridDev* gd = nullptr,
                const PackGrid* grid = nullptr);
// exclusive scan of uint32 data[n] in place; data[n] receives the total when
// write_total.  tmp is grown as needed.
int launch_exclusive_scan(pcc_index* ix, hipStream_t s, unsigned int* data, size_t n, DevBuf& tmp);
// packed u64 keys (d2 bits << 32 | packed position) -> original idx, d2; n keys; q (nullable,
// one per key) flags invalid queries (w < 0) which get -1/+inf, as do empty keys (~0)
// mirror (nullable): device word copied to *mirror_host (pinned) by the kernel -- fallback counter for stats
int launch_unpack(hipStream_t s, const unsigned long long* packed, const float4* q, size_t n,
                  int32_t* idx, float* d2, const unsigned int* mirror_dev = nullptr, unsigned int* mirror_host = nullptr);
int launch_transform(hipStream_t s, const float* T16_dev_or_null, const float T[16],
                     const void* src, size_t n, size_t sstride, void* dst, size_t dstride, unsigned int* zero_word = nullptr);
// packed points whose w flags them invalid get NaN coordinates again (in place): what a raw cloud looked like
int launch_nanify(hipStream_t s, float4* pts, size_t n);
// dst[i].w = src[i].w (validity flags of packed points)
int launch_gather_sorted(hipStream_t s, const float4* q, const unsigned int* order, const unsigned int* n_sorted, size_t n, float4* dst,
                         unsigned int* ns_word);
int launch_copy_w(hipStream_t s, const float4* src, float4* dst, size_t n);
// SOR: mean_dist[orig(i)] = float(sum_{j=1..K-1} sqrt(double(d2_j)) / (K-1)) from the K-NN keys of
// the self query; rows with fewer than K neighbours keep 0
int launch_sor_mean(hipStream_t s, const unsigned long long* keys, const float4* refs, size_t n, int K,
                    float* mean_dist, const float* d2_rows = nullptr);
// SOR: sum / sq_sum / threshold / inlier mask of the mean distances on the device (pack.hip); stats_dev receives
// {double sum, sq, thr; uint64 kept; uint32 exact, pad}: exact == 0 -> the tree sums may differ from PCL's in-order sums
int launch_sor_stats(hipStream_t s, const float* m, size_t n, const GridDev* gd, int K, double stddev_mult, double* scratch,
                     void* stats_dev, uint8_t* inlier_dev);
int launch_sor_partial(hipStream_t s, const float* m, size_t n, double* scratch, double* out4_dev);
int launch_sor_threshold_mask(hipStream_t s, const float* m, size_t n, const GridDev* gd, int K, double stddev_mult,
                              const double* in4_dev, void* stats_dev, uint8_t* inlier_dev);
void sor_threshold_host(const double in4[4], double n_valid, int K, double stddev_mult, double* thr, int* exact);
int launch_clamp_counts(hipStream_t s, int32_t* counts, size_t n, int32_t cap);
int launch_knn_rows_to_csr(hipStream_t s, const unsigned long long* keys, const int32_t* ridx, const float* rd2, int K, float r2,
                           const int64_t* offsets, size_t nq, int32_t* idx_out, float* d2_out);
int launch_copy_row_prefix(hipStream_t s, const unsigned long long* src, int k_src, unsigned long long* dst, int k_dst, size_t n);

// ---- exhaustive engine (nn1_brute.hip) -------------------------------------------
// For every query q[i] (float4, w<0 = invalid) min over refs[0..m) of the unfused
// squared distance, lowest original index on ties, merged into out[i] with a 64-bit
// atomicMin (out must be pre-set to ~0).  If qlist != nullptr only the queries
// qlist[0..*qcount) are processed (GRID fallback list; count read on device).
// idx_from_w (list mode only): report refs[p].w as the index instead of p (seed subset, see grid.hip)
int launch_nn1_brute(hipStream_t s, const float4* refs, size_t m, const float4* q,
                     size_t n, unsigned long long* out, const unsigned int* qlist,
                     const unsigned int* qcount_dev, size_t qcount_max, bool idx_from_w = false);

// ---- small-call form (small.hip): the query side of a k = 1 call in one launch -- raw host-pinned queries in, (idx, d2) out into
// pinned host memory, q_packed / out_packed written as the separate launches would
int launch_small_nn1(hipStream_t s, const void* raw_q, size_t nq, size_t stride, const float4* refs, size_t n, float4* q_packed,
                     unsigned long long* out_packed, int32_t* idx, float* d2, unsigned int* tie_blocks, unsigned char* tie_q);  // PCC_TIES_FLANN: tied queries per 64 (pinned) and a flag per query (device)
constexpr size_t SMALL_FUSED_REFS = 4096;    // indexed clouds up to here take it (PCC_ENGINE_AUTO's exhaustive range)
constexpr size_t SMALL_FUSED_POINTS = 8192;  // indexed clouds up to here: the grid parameters come from the pack kernel's last workgroup

// ---- grid.hip --------------------------------------------------------------------------
int grid_params(pcc_index* ix, const float* blk_stats_dev, int n_blocks);  // async: d_grid + pinned mirror
int grid_params_fused(pcc_index* ix, PackGrid* pg);                        // the same from inside the pack kernel: fills *pg for launch_pack
int grid_build(pcc_index* ix);                                             // async: cell sort of the references
int sync_info(pcc_index* ix);                                              // wait for the pinned mirror, refresh host fields
unsigned int grid_nc_cap(size_t n, double ppc);
int grid_nn1(pcc_index* ix, const float4* q, size_t nq, unsigned long long* out);
// sort queries by reference-grid cell: order[0..*n_sorted) (device) lists the valid queries
bool grid_nn1_takes_transform(const pcc_index* ix);
int grid_sort_queries(pcc_index* ix, const float4* q, size_t nq, unsigned int** order_dev,
                      unsigned int** n_sorted_dev);
float grid_slack(const GridParams& g);
// ---- cellsort.hip: LDS-based two-level counting sort by cell ---------------------------------------
int cell_sort(pcc_index* ix, const float4* pts, size_t n, bool refs, float4* out_pts, unsigned int* out_order,
              unsigned int* cell_start, unsigned int** n_sorted_dev, const GridDev* gd_override = nullptr,
              unsigned int nc_cap_override = 0);
// ---- cellsort_mp.hip: the same contract in three coalesced levels (large clouds); out_pts may also be given for queries
int cell_sort_mp(pcc_index* ix, const float4* pts, size_t n, bool refs, float4* out_pts, unsigned int* out_order,
                 unsigned int* cell_start, unsigned int** n_sorted_dev, const GridDev* gd_override = nullptr,
                 unsigned int nc_cap_override = 0);
// ---- voxel.hip: pcl::VoxelGrid (leaf-lattice centroids) ---------------------------------------------
int voxel_grid(pcc_index* ctx, const void* pts, size_t n, size_t stride, int mem, float leaf, int has_rgb,
               void* out, size_t out_stride, size_t* out_n);
// ---- knn.hip: k-NN, radius search (GRID engine) ----------------------------------------
// keys[nq][K] (pre-set to ~0) receive the K smallest (d2, position) keys ascending
int grid_knn(pcc_index* ix, const float4* q, size_t nq, int K, unsigned long long* keys, int32_t* idx_out = nullptr,
             float* d2_out = nullptr);
bool grid_knn_delivers(int K);
// counts[i] = #refs with d2 < r2; with fill != 0 also writes keys at offsets[i]..
// idx_out / d2_out / delivered (fill only): when given, a fill that takes the wave-per-query route writes the caller's
// arrays itself (sorted in registers) and sets *delivered
int grid_radius(pcc_index* ix, const float4* q, size_t nq, float r, float r2, int32_t* counts,
                const int64_t* offsets, unsigned long long* keys, int sorted, size_t total = 0, int32_t* idx_out = nullptr,
                float* d2_out = nullptr, bool* delivered = nullptr);
constexpr int PCC_ERR_RETRY_HOST = -1000;  // sac_plane without a host array met a degenerate sample: repeat with one (internal)
int sac_plane(pcc_index* ix, const float4* pts_dev, size_t n, const char* host_base, size_t host_stride,
              int max_iterations, double threshold, double probability, int optimize, int32_t* inliers_dev,
              size_t* n_inliers, float coeff[4], int* iterations_out);
int launch_normals(hipStream_t s, const unsigned long long* keys, const float4* refs, const float4* cell_refs,
                   const GridDev* gd, size_t n, int K, const float vp[3], float4* out);
int normals_radius(pcc_index* ix, double radius, const float vp[3], float4* out);
int grid_region_growing(pcc_index* ix, const unsigned long long* keys, const float4* normals, int K, float smoothness,
                        float curvature_threshold, uint32_t min_size, uint32_t max_size, int32_t* labels_dev,
                        int32_t* n_clusters);
int grid_first_within(pcc_index* ix, const float4* q, size_t nq, double radius, int32_t* idx_dev);
// ---- cluster.hip ------------------------------------------------------------------------
int grid_clusters(pcc_index* ix, float r, float r2, uint32_t min_size, uint32_t max_size,
                  int32_t* labels_dev /* n_orig, device */, int32_t* n_clusters, int32_t* sizes, int max_sizes);
// ---- flann_order.hip: flags[i] = 1 when another reference shares query i's minimum distance; the tied queries walked
// through FLANN's tree on the device
int resolve_ties_flann(pcc_index* ix, const float4* q, unsigned long long* keys, size_t nq, bool may_wait = false);
// The small-call form of the replay (small.hip has flagged the tied queries, tie_q[i] != 0, and the indexed cloud's raw records are
// still in the pinned buffer): tree built from those records, uploaded without a wait, ONE launch walks the flagged queries and
// writes the changed indices into idx_host (pinned) and their number per 64 queries into changed_blocks (pinned).  *done = false:
// the tree is deeper than the device walk takes, nothing was enqueued (the caller takes resolve_ties_flann).
int small_tie_replay(pcc_index* ix, const void* raw_refs, size_t nq, const unsigned char* tie_q, int32_t* idx_host,
                     unsigned int* changed_blocks, bool* done);
// ---- icp.hip -----------------------------------------------------------------------------
// per-workgroup partial sums (17 doubles each) of the matched pairs; returns #blocks written
struct IcpState;
// fuse (device-resident loop on one GPU): the last workgroup of k_icp_sums to finish also solves the pass (k_icp_solve's work);
// ticket = a zeroed device word; center_dev must be given
struct IcpFuse {
    unsigned int* ticket;
    IcpState* st;
    int max_iter, fixed;
    unsigned int* zero_word;
};
int launch_icp_sums(hipStream_t s, const float4* src, size_t n, const unsigned long long* keys,
                    const float4* refs, double* partials, int* n_blocks, const unsigned int* mirror_dev = nullptr,
                    unsigned int* mirror_host = nullptr, const double* center_dev = nullptr,  // sums of p - center, q - center (device pointer)
                    const IcpFuse* fuse = nullptr);
constexpr int ICP_MAX_BLOCKS = 480;  // (480 rows of 17 doubles fit the 64 KB of LDS k_icp_solve stages them in)
// state of the device-resident ICP loop (pcc_icp_align): no host round trip per pass
struct IcpState {
    float Ti[16];     // transform of the pass just solved (identity once the loop has stopped)
    float T[16];      // running product T_i * ... * T_1
    double prev_mse;  // mean squared distance of the previous pass (convergence test)
    int it;           // iterations completed
    int stopped;      // the loop has ended (criteria met, iteration cap, or too few correspondences): later passes are no-ops
    int converged;    // pcl::DefaultConvergenceCriteria's verdict at the stop
};
int launch_icp_solve(hipStream_t s, const double* partials, int n_blocks, IcpState* state, int max_iter, int fixed,
                     const double* center_dev, unsigned int* zero_word = nullptr);
int launch_icp_center(hipStream_t s, const float4* src, size_t n, double* center_dev);  // first valid point of src
// the per-workgroup rows added up in workgroup order (what k_icp_solve does before it solves): sums17[k] on the device
int launch_icp_rows_to_sums(hipStream_t s, const double* partials, int n_blocks, double* sums17);
// collectives of the sharded ICP loop, supplied by comm.hip (RCCL on the handle's stream); api.hip knows no RCCL type
struct IcpHooks {
    void* ctx;
    int (*allreduce_sum_f64)(void* ctx, double* dev, int count, hipStream_t s);
    int (*bcast_f64)(void* ctx, double* dev, int count, int root, hipStream_t s);
    int (*agree)(void* ctx, int local_status);  // the worst status over the ranks (one all-reduce): same return everywhere
};
int icp_align_impl(pcc_index* ix, const IcpHooks* hooks, const void* src, size_t n, size_t stride, int mem, int max_iter, int fixed,
                   float T[16], double* fitness, int* iterations, int* converged);
// ---- api.hip internals comm.hip builds on -------------------------------------------------------------------------
int check_points(const void* pts, size_t n, size_t stride, int mem);
int stage_queries(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem);
int nn1_packed(pcc_index* ix, size_t nq);
int set_input(pcc_index* ix, const void* pts, size_t n, size_t stride, int mem);
int need_grid(pcc_index* ix);                              // (the GRID engine's index, built on demand)
int make_handle(int device, int engine, pcc_index** out);  // an empty handle on `device`

}  // namespace pcc
