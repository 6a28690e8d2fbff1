// api.hip -- the C-ABI of libpcc_nn (include/pcc_nn.h) on top of the gfx950 kernels.
// Host-side glue only: argument checks, H2D/D2H staging, launch order.  There is
// no CPU compute fallback: without a HIP device every entry point fails with
// PCC_ERR_DEVICE.
#include "pcc_internal.hpp"
#include <atomic>
#include "rigid_solve.hpp"
#include "host_pipe.hpp"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <thread>

namespace pcc {

#ifdef PCC_COUNT_PAIRS
unsigned long long pairs_take_grid();
unsigned long long pairs_take_knn();
unsigned long long pairs_take_cluster();
unsigned long long pairs_take_flann();
#endif

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

// the range every option value must lie in, whoever sets it (pcc_index_set_option or a PCC_* environment default)
static bool option_in_range(int option, double value) {
    if (!std::isfinite(value)) return false;
    switch (option) {
        case PCC_OPT_GRID_PPC: return value > 0 && value <= 1024;
        case PCC_OPT_GRID_TRIM: return value >= 0 && value <= 8;
        case PCC_OPT_FAR_MODE: return value >= -1 && value <= 1;
        case PCC_OPT_SORT_MP_MIN: case PCC_OPT_SORT_MP_MIN_Q: return value >= 0;
        case PCC_OPT_NN1_KERNEL: return value >= 0 && value <= 3;
        case PCC_OPT_EC_CELLS: return value >= 0 && value <= 4;
        case PCC_OPT_KNN_CACHE_K: return value >= 0 && value <= 512;
        case PCC_OPT_NN1_DENSE_MIN: return value >= 1 && value <= 1000000;
        case PCC_OPT_FLANN_SPLIT: return value >= 0 && value <= 2;
        case PCC_OPT_SORT_STAGE1: return value >= 0 && value <= 2;
        case PCC_OPT_OVERLAP_PREP: return value >= 0 && value <= 2;
        case PCC_OPT_GRID_AXES: return value >= -2 && value <= 5;
        case PCC_OPT_XCD_RUN: return value >= 1 && value <= 4096;
        case PCC_OPT_FUSE_PARAMS: return value >= 0 && value <= 3;
        case PCC_OPT_KNN_RUN: return value >= 1 && value <= 64;
        case PCC_OPT_HOST_PIPE: case PCC_OPT_SCAN_CHAINED: return value == 0 || value == 1;
        default: return value == 0 || value == 1;
    }
}
static double* option_slot(Options& o, int option, int** as_int) {
    *as_int = nullptr;
    switch (option) {
        case PCC_OPT_GRID_PPC: return &o.grid_ppc;
        case PCC_OPT_SORT_MP_MIN: return &o.sort_mp_min;
        case PCC_OPT_SORT_MP_MIN_Q: return &o.sort_mp_min_q;
        case PCC_OPT_GRID_TRIM: *as_int = &o.grid_trim; return nullptr;
        case PCC_OPT_FAR_MODE: *as_int = &o.far_mode; return nullptr;
        case PCC_OPT_ICP_WARM: *as_int = &o.icp_warm; return nullptr;
        case PCC_OPT_ICP_DEVICE_LOOP: *as_int = &o.icp_device_loop; return nullptr;
        case PCC_OPT_EC_CELLS: *as_int = &o.ec_cells; return nullptr;
        case PCC_OPT_NN1_KERNEL: *as_int = &o.nn1_kernel; return nullptr;
        case PCC_OPT_FLANN_SPLIT: *as_int = &o.flann_split; return nullptr;
        case PCC_OPT_NN1_DENSE_MIN: *as_int = &o.nn1_dense_min; return nullptr;
        case PCC_OPT_KNN_KERNEL: *as_int = &o.knn_kernel; return nullptr;
        case PCC_OPT_KNN_CACHE_K: *as_int = &o.knn_cache_k; return nullptr;
        case PCC_OPT_NN1_OPEN_FLAT: *as_int = &o.nn1_open_flat; return nullptr;
        case PCC_OPT_SORT_STAGE1: *as_int = &o.sort_stage1; return nullptr;
        case PCC_OPT_ICP_SORTED: *as_int = &o.icp_sorted; return nullptr;
        case PCC_OPT_OVERLAP_PREP: *as_int = &o.overlap_prep; return nullptr;
        case PCC_OPT_GRID_AXES: *as_int = &o.grid_axes; return nullptr;
        case PCC_OPT_XCD_RUN: *as_int = &o.xcd_run; return nullptr;
        case PCC_OPT_FUSE_PARAMS: *as_int = &o.fuse_params; return nullptr;
        case PCC_OPT_HOST_PIPE: *as_int = &o.host_pipe; return nullptr;
        case PCC_OPT_SCAN_CHAINED: *as_int = &o.scan_chained; return nullptr;
        case PCC_OPT_KNN_RUN: *as_int = &o.knn_run; return nullptr;
        default: return nullptr;
    }
}

// PCC_* environment variables give a new handle its defaults; a value outside the option's range is ignored (the
// built-in default stays), exactly what pcc_index_set_option would have refused
void Options::from_env() {
    static const struct { const char* name; int option; } vars[] = {
        {"PCC_GRID_PPC", PCC_OPT_GRID_PPC}, {"PCC_GRID_TRIM", PCC_OPT_GRID_TRIM}, {"PCC_GRID_FAR", PCC_OPT_FAR_MODE},
        {"PCC_ICP_WARM", PCC_OPT_ICP_WARM}, {"PCC_ICP_DEVICE_LOOP", PCC_OPT_ICP_DEVICE_LOOP}, {"PCC_EC_CELLS", PCC_OPT_EC_CELLS},
        {"PCC_SORT_MP_MIN", PCC_OPT_SORT_MP_MIN}, {"PCC_SORT_MP_MIN_Q", PCC_OPT_SORT_MP_MIN_Q}, {"PCC_NN1_KERNEL", PCC_OPT_NN1_KERNEL},
        {"PCC_FLANN_SPLIT", PCC_OPT_FLANN_SPLIT}, {"PCC_NN1_DENSE_MIN", PCC_OPT_NN1_DENSE_MIN}, {"PCC_KNN_KERNEL", PCC_OPT_KNN_KERNEL},
        {"PCC_KNN_CACHE_K", PCC_OPT_KNN_CACHE_K}, {"PCC_NN1_OPEN_FLAT", PCC_OPT_NN1_OPEN_FLAT}, {"PCC_SORT_STAGE1", PCC_OPT_SORT_STAGE1},
        {"PCC_ICP_SORTED", PCC_OPT_ICP_SORTED}, {"PCC_OVERLAP_PREP", PCC_OPT_OVERLAP_PREP},
        {"PCC_GRID_AXES", PCC_OPT_GRID_AXES}, {"PCC_XCD_RUN", PCC_OPT_XCD_RUN},
        {"PCC_FUSE_PARAMS", PCC_OPT_FUSE_PARAMS}, {"PCC_HOST_PIPE", PCC_OPT_HOST_PIPE},
        {"PCC_SCAN_CHAINED", PCC_OPT_SCAN_CHAINED}, {"PCC_KNN_RUN", PCC_OPT_KNN_RUN}};
    for (const auto& v : vars) {
        const char* txt = getenv(v.name);
        if (!txt || !*txt) continue;
        char* end = nullptr;
        const double value = strtod(txt, &end);
        int* pi = nullptr;
        double* pd = option_slot(*this, v.option, &pi);
        if (end == txt || !option_in_range(v.option, pi ? std::trunc(value) : value)) continue;
        if (pd) *pd = value; else *pi = (int)value;
    }
}

// test hook (pcc_debug_fail_alloc): the n-th device allocation from now on fails as an exhausted hipMalloc would
static std::atomic<int> g_fail_alloc{0};

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap && p) return PCC_OK;
    // (one compare-exchange per armed growth: two threads can never both take the count through zero)
    int armed = g_fail_alloc.load(std::memory_order_relaxed);
    while (armed > 0 && !g_fail_alloc.compare_exchange_weak(armed, armed - 1)) {}
    if (armed == 1) {
        set_error("hipMalloc(%zu) failed: injected by pcc_debug_fail_alloc", bytes);
        return PCC_ERR_NOMEM;  // (the buffer keeps what it had: exactly what a refused growth leaves behind)
    }
    if (bytes == 0) bytes = 256;
    size_t want = bytes + bytes / 8;  // slack so slowly growing batches do not realloc each call
    want = (want + 255) & ~(size_t)255;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return PCC_ERR_NOMEM;
    }
    cap = want;
    return PCC_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

int HostBuf::reserve(size_t bytes) {
    if (bytes <= cap && p) return PCC_OK;
    if (bytes == 0) bytes = 256;
    size_t want = bytes + bytes / 8;
    want = (want + 4095) & ~(size_t)4095;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return PCC_ERR_NOMEM;
    }
    cap = want;
    return PCC_OK;
}
void HostBuf::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        int cur;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};
#define PCC_ENTER(ix)                                                         \
    if (!(ix)) { pcc::set_error("null index"); return PCC_ERR_INVALID; }      \
    std::lock_guard<std::mutex> _lock((ix)->mu);                              \
    pcc::DeviceGuard _guard((ix)->device);                                    \
    if (!_guard.ok) { pcc::set_error("hipSetDevice(%d) failed", (ix)->device); return PCC_ERR_DEVICE; } \
    pcc::entered(ix);
// entry points that put nothing on the stream leave "the build was the last thing enqueued" as they found it
#define PCC_NOTHING_ENQUEUED(ix) (ix)->build_fresh = (ix)->after_build

// Stage a caller cloud (host or device AoS) as packed float4 on the device.
// host: the raw array (or, for large pageable clouds, its x / y / z alone: host_pipe.hpp) to the device, then the pack kernel.
static int stage_points(pcc_index* ix, const void* pts, size_t n, size_t stride, int mem,
                        DevBuf& raw, float4* packed, float* blk_stats = nullptr, int* n_blocks = nullptr,
                        unsigned int* zero_word = nullptr, float4* seeds = nullptr, unsigned long long* invalid_keys = nullptr,
                        unsigned int* cells = nullptr, const GridDev* gd = nullptr, const PackGrid* grid = nullptr) {
    const void* src = pts;
    if (mem == PCC_MEM_HOST && n > 0) {
        const size_t bytes = (n - 1) * stride + 12;
        if (ix->opt.host_pipe && bytes >= PIPE_MIN_BYTES && !host_pointer_is_pinned(pts)) {
            // (everything enqueued so far may still be reading `raw`: the chunks are enqueued on the same stream, in order behind it)
            const size_t dst_stride = stride >= 24 ? 12 : stride;
            PCC_TRY(raw.reserve(n * dst_stride + 16));
            if (!ix->pipe) ix->pipe = new HostPipe();
            ix->small_raw_n = 0;  // (the chunk buffers are the small-call buffers: an indexed cloud's raw records do not survive this)
            PCC_TRY(ix->pipe->upload(ix->stream, static_cast<const char*>(pts), n, stride, raw.as<char>(), dst_stride));
            stride = dst_stride;
        } else if (ix->opt.host_pipe && bytes <= SMALL_DIRECT_BYTES) {
            // SMALL clouds -- the reference's descriptor clouds, 4 ... 18 381 records of 128 bytes, up to 300 calls per comparison
            // (src/comparator.cpp:560-588): a hipMemcpyAsync from pageable memory makes the host wait for a staged copy (~15 us a
            // piece).  Up to SMALL_DIRECT_BYTES the cloud is copied into the handle's pinned buffer instead (slot 0: indexed clouds,
            // 1: query clouds) and the pack kernel reads it from there across the link -- no copy command at all.  (Between 1 and
            // 8 MB the runtime's own staging is as good: 18 381 descriptors 244 us a call against 277 through the pinned buffer.)
            if (!ix->pipe) ix->pipe = new HostPipe();
            PCC_TRY(ix->pipe->init());
            const int slot = blk_stats ? 0 : 1;
            PCC_HIP(hipEventSynchronize(ix->pipe->ev[slot]));  // (whoever read this buffer last has finished: nearly always true already)
            memcpy(ix->pipe->buf[slot].p, pts, bytes);
            src = ix->pipe->buf[slot].p;
            if (slot == 0) { ix->small_raw_n = n; ix->small_raw_stride = stride; }  // (the indexed cloud's records stay there: small_tie_replay)
            PCC_TRY(launch_pack(ix->stream, src, n, stride, packed, blk_stats, n_blocks, zero_word, seeds, invalid_keys, cells, gd, grid));
            PCC_HIP(hipEventRecord(ix->pipe->ev[slot], ix->stream));
            return PCC_OK;
        } else {
            PCC_TRY(raw.reserve(n * stride));
            PCC_HIP(hipMemcpyAsync(raw.p, pts, bytes, hipMemcpyHostToDevice, ix->stream));
        }
        src = raw.p;
    }
    return launch_pack(ix->stream, src, n, stride, packed, blk_stats, n_blocks, zero_word, seeds, invalid_keys, cells, gd, grid);
}

int check_points(const void* pts, size_t n, size_t stride, int mem) {
    if (mem != PCC_MEM_HOST && mem != PCC_MEM_DEVICE) { set_error("bad mem space %d", mem); return PCC_ERR_INVALID; }
    if (n && !pts) { set_error("null point pointer"); return PCC_ERR_INVALID; }
    if (stride < 12 || stride % 4) { set_error("stride %zu must be a multiple of 4 and >= 12", stride); return PCC_ERR_INVALID; }
    if (n >= (1ull << 31)) { set_error("more than 2^31 points"); return PCC_ERR_UNSUPPORTED; }
    return PCC_OK;
}

// queries -> ix->q_packed (float4, w < 0 marks a non-finite query)
int stage_queries(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem) {
    PCC_TRY(ix->q_packed.reserve(nq * sizeof(float4)));
    PCC_TRY(ix->out_packed.reserve(nq * sizeof(unsigned long long)));
    // the pack kernel also zeroes the GRID engine's fallback counter (small + 32) and presets the result key of
    // every non-finite query to "nothing found"
    ix->fb_zeroed = true;
    // clouds that will take the three-level sort: the pack kernel also writes every query's grid cell (4 B), which level 1 of
    // the sort then reads instead of the points (pcc_index::q_cells; valid until the sort has used it)
    unsigned int* cells = nullptr;
    ix->q_cells_n = 0;
    if (ix->engine == PCC_ENGINE_GRID && ix->has_grid && nq >= (size_t)ix->opt.sort_mp_min_q) {
        PCC_TRY(ix->q_cells.reserve(nq * sizeof(unsigned int) + 64));
        cells = ix->q_cells.as<unsigned int>();
        ix->q_cells_n = nq;
    }
    return stage_points(ix, q, nq, stride, mem, ix->q_raw, ix->q_packed.as<float4>(), nullptr, nullptr,
                        ix->small.as<unsigned int>() + 32, nullptr, ix->out_packed.as<unsigned long long>(), cells,
                        cells ? ix->d_grid.as<GridDev>() : nullptr);
}

// deliver device results to the caller's memory space
template <class T>
static int deliver(pcc_index* ix, const T* dev, T* user, size_t count, int mem) {
    if (!user || count == 0) return PCC_OK;
    if (mem == PCC_MEM_HOST) {
        if (ix->opt.host_pipe && count * sizeof(T) >= PIPE_MIN_BYTES && !host_pointer_is_pinned(user)) {
            if (!ix->pipe) ix->pipe = new HostPipe();
            ix->small_raw_n = 0;  // (as in stage_points)
            PCC_TRY(ix->pipe->download(ix->stream, reinterpret_cast<const char*>(dev), reinterpret_cast<char*>(user), count * sizeof(T)));
        } else
        PCC_HIP(hipMemcpyAsync(user, dev, count * sizeof(T), hipMemcpyDeviceToHost, ix->stream));
    } else if (user != dev) {
        PCC_HIP(hipMemcpyAsync(user, dev, count * sizeof(T), hipMemcpyDeviceToDevice, ix->stream));
    }
    return PCC_OK;
}


// k = 1 search of ix->q_packed[0..nq) into ix->out_packed (u64 per query)
// The small-call form (small.hip): host queries against a small exhaustively searched cloud -- the reference's descriptor
// matching.  PCC_OPT_HOST_PIPE = 0 keeps the separate launches (pack, preset, search, unpack).
static bool small_call(const pcc_index* ix, size_t nq, size_t stride, int mem) {
    return mem == PCC_MEM_HOST && ix->opt.host_pipe && ix->engine == PCC_ENGINE_BRUTE && ix->n_orig <= SMALL_FUSED_REFS && nq > 0 &&
           (nq - 1) * stride + 12 <= SMALL_DIRECT_BYTES && nq * sizeof(float) <= SMALL_RESULT_BYTES;
}
// queries to the pinned buffer, one launch, one wait; the results are in host_a (indices) and host_b (squared distances).
// PCC_TIES_FLANN: the kernel also counts the queries whose minimum is shared by a second reference; only if there are any does the
// tie replay (a kd-tree of FLANN's shape over the indexed cloud, flann_order.hip) run, on the q_packed / out_packed the kernel left
static int small_nn1(pcc_index* ix, const void* q, size_t nq, size_t stride, bool want_idx, bool want_d2) {
    const bool flann = ix->tie_mode == PCC_TIES_FLANN;
    const size_t nblk = (nq + 63) / 64;
    PCC_TRY(ix->q_packed.reserve(nq * sizeof(float4)));
    PCC_TRY(ix->out_packed.reserve(nq * sizeof(unsigned long long)));
    PCC_TRY(ix->host_a.reserve((nq + 2 * nblk) * sizeof(int32_t)));  // (+ per workgroup: tied queries, indices the replay changed)
    PCC_TRY(ix->host_b.reserve(nq * sizeof(float)));
    if (flann) PCC_TRY(ix->tie_buf.reserve(nq + 64));  // (a tie flag per query)
    if (!ix->pipe) ix->pipe = new HostPipe();
    PCC_TRY(ix->pipe->init());
    PCC_HIP(hipEventSynchronize(ix->pipe->ev[1]));  // (whoever read the query buffer last has finished)
    memcpy(ix->pipe->buf[1].p, q, (nq - 1) * stride + 12);
    ix->fb_zeroed = false;
    ix->q_cells_n = 0;
    ix->stats[0] = 0;
    ix->stats[1] = nq;
    ix->stats_pending = false;
    int32_t* hidx = ix->host_a.as<int32_t>();
    unsigned int* tie_blocks = flann ? reinterpret_cast<unsigned int*>(hidx + nq) : nullptr;
    unsigned char* tie_q = flann ? ix->tie_buf.as<unsigned char>() : nullptr;
    ev_mark(ix, EV_MAIN0);
    PCC_TRY(launch_small_nn1(ix->stream, ix->pipe->buf[1].p, nq, stride, ix->refs.as<float4>(), ix->n_orig, ix->q_packed.as<float4>(),
                             ix->out_packed.as<unsigned long long>(), want_idx || flann ? hidx : nullptr,
                             want_d2 ? ix->host_b.as<float>() : nullptr, tie_blocks, tie_q));
    ev_mark(ix, EV_MAIN1);
    ev_mark(ix, EV_CALL1);
    // (the wait below is also what lets the next call overwrite the pinned query buffer: no event is recorded for it)
    PCC_HIP(hipStreamSynchronize(ix->stream));
    if (!flann) return PCC_OK;
    unsigned int tied = 0;
    for (size_t b = 0; b < nblk; ++b) tied += tie_blocks[b];
    ix->ties_pending = false;
    ix->ties_flagged = tied;
    ix->ties_changed = 0;
    if (tied == 0) return PCC_OK;
    if (ix->small_raw_n == ix->n_orig) {
        unsigned int* changed_blocks = tie_blocks + nblk;
        bool done = false;
        PCC_TRY(small_tie_replay(ix, ix->pipe->buf[0].p, nq, tie_q, hidx, changed_blocks, &done));
        if (done) {
            PCC_HIP(hipStreamSynchronize(ix->stream));
            for (size_t b = 0; b < nblk; ++b) ix->ties_changed += changed_blocks[b];
            return PCC_OK;
        }
    }
    // (the separate launches' replay: its tree build downloads the packed cloud into host_a, which may move -- the indices are
    // unpacked again, all of them, into wherever it is afterwards; the distances in host_b stand)
    PCC_TRY(resolve_ties_flann(ix, ix->q_packed.as<float4>(), ix->out_packed.as<unsigned long long>(), nq));
    PCC_TRY(ix->host_a.reserve((nq + 2 * nblk) * sizeof(int32_t)));
    PCC_TRY(launch_unpack(ix->stream, ix->out_packed.as<unsigned long long>(), nullptr, nq, ix->host_a.as<int32_t>(), nullptr));
    PCC_HIP(hipStreamSynchronize(ix->stream));
    return PCC_OK;
}

int nn1_packed(pcc_index* ix, size_t nq) {
    PCC_TRY(ix->out_packed.reserve(nq * sizeof(unsigned long long)));
    auto* out = ix->out_packed.as<unsigned long long>();
    // GRID: every valid query's key is written by the search kernel itself (no 8 MB memset)
    if (ix->engine == PCC_ENGINE_GRID) return grid_nn1(ix, ix->q_packed.as<float4>(), nq, out);
    PCC_HIP(hipMemsetAsync(out, 0xff, nq * sizeof(unsigned long long), ix->stream));
    ix->stats[0] = 0;
    ix->stats[1] = nq;
    ix->stats_pending = false;
    ev_mark(ix, EV_MAIN0);
    int st = launch_nn1_brute(ix->stream, ix->refs.as<float4>(), ix->n_orig, ix->q_packed.as<float4>(), nq,
                              out, nullptr, nullptr, 0);
    ev_mark(ix, EV_MAIN1);
    return st;
}

static int resolve_engine(int requested, size_t n) {
    // the grid build costs a few passes over the cloud; below ~4k points one exhaustive
    // sweep is cheaper than building it
    if (requested == PCC_ENGINE_AUTO) return n >= 4096 ? PCC_ENGINE_GRID : PCC_ENGINE_BRUTE;
    return requested;
}

// (re)build the index over a new cloud, fully asynchronous on the index's stream:
// pack (+ per-workgroup bbox / non-finite counts) -> grid sizing ON THE DEVICE -> cell sort.
// Non-finite points stay in place flagged w = -1 (no compaction: position == original index).
// Host-visible facts (n_valid, bbox, grid) arrive through a pinned mirror; sync_info() waits.
static int set_input_impl(pcc_index* ix, const void* pts, size_t n, size_t stride, int mem) {
    ev_next(ix);
    ev_mark(ix, EV_BUILD0);
    PCC_TRY(ix->refs.reserve(n * sizeof(float4)));
    int nblk = 0;
    PCC_TRY(ix->seeds.reserve(((n + PCC_SEED_STRIDE - 1) / PCC_SEED_STRIDE) * sizeof(float4)));  // the pack kernel also emits the seed subset
    PackGrid pg{};
    // (small clouds: a launch is what a call costs there, and the fence the fused form pays is nothing over a handful of rows)
    const bool fused = (ix->opt.fuse_params & 1) != 0 || (ix->opt.host_pipe && n <= SMALL_FUSED_POINTS);
    if (fused) PCC_TRY(grid_params_fused(ix, &pg));
    PCC_TRY(stage_points(ix, pts, n, stride, mem, ix->q_raw, ix->refs.as<float4>(), ix->blk_stats.as<float>(), &nblk,
                         nullptr, ix->seeds.as<float4>(), nullptr, nullptr, nullptr, fused ? &pg : nullptr));
    if (!fused) PCC_TRY(grid_params(ix, ix->blk_stats.as<float>(), nblk));
    ix->engine = resolve_engine(ix->engine_requested, n);
    // everything a query needs to be packed and sorted exists from here on (PrepOverlap below)
    if (ix->engine == PCC_ENGINE_GRID && ix->opt.overlap_prep) {
        if (!ix->params_ev) PCC_HIP(hipEventCreateWithFlags(&ix->params_ev, hipEventDisableTiming));
        PCC_HIP(hipEventRecord(ix->params_ev, ix->stream));
        ix->params_ev_set = true;
    }
    if (ix->engine == PCC_ENGINE_GRID) PCC_TRY(grid_build(ix));
    ev_mark(ix, EV_BUILD1);
    return PCC_OK;
}
// on any failure the index is left EMPTY (n_orig = 0: every search then answers PCC_ERR_EMPTY instead of
// launching over buffers a failed reserve() has freed)
int set_input(pcc_index* ix, const void* pts, size_t n, size_t stride, int mem) {
    ix->n_valid = 0;
    ix->has_grid = false;
    ix->order_valid = false;
    ix->flann_valid = false;
    ix->small_raw_n = 0;
    ix->occ_valid = false;
    ix->q_cells_n = 0;  // (cells staged against the grid that is about to be replaced)
    ix->self_rows_k = 0;
    ix->n_orig = n;  // the build steps size their launches from it
    ix->params_ev_set = false;
    ix->edge_fresh = false;
    const int st = set_input_impl(ix, pts, n, stride, mem);
    if (st != PCC_OK) { ix->n_orig = 0; ix->has_grid = false; }
    ix->build_fresh = st == PCC_OK && ix->params_ev_set;  // (until the next entry point: pcc::entered)
    return st;
}

// ---- query staging beside the build (PCC_OPT_OVERLAP_PREP) -----------------------------------------------------------------
// The reference builds its tree and asks at once (src/comparator.cpp:564-577).  Here the build is pack -> grid parameters ->
// three sort levels, and a query needs only the grid parameters to be packed (with its cell) and sorted: two thirds of the
// build and the whole query staging are independent streaming passes.  While an object of this type lives, the handle's
// launches go to side_stream -- which waits for the build's k_grid_params, i.e. for everything enqueued before it as well, but
// not for the build's sort -- and the sort's scratch buffers are swapped for a set of their own; its end joins the side
// stream back into the main one, error or not.  Buffers the staging writes besides (q_packed, q_cells, out_packed, the
// order in scratch_g, the counters in `small`) are touched by no build kernel, and their last readers were enqueued before
// the build.
struct PrepOverlap {
    pcc_index* ix;
    hipStream_t main_stream = nullptr;
    bool on = false;
    // (below ~2M queries the staging is a few launches of 10-20 us each and the second stream's events cost what they hide:
    // 1M x 1M 0.221 / 0.222 / 0.226 ms without, 0.225 with; 10M x 10M 1.331 / 1.334 / 1.319 -> 1.308 / 1.303 / 1.301)
    static constexpr size_t min_queries = 2000000;
    // Only on the library's OWN stream: a caller that has handed its stream over (pcc_index_set_stream) may have enqueued the
    // kernel that produces the queries on it between the build and this search -- "enqueued on the index's stream" is the
    // ordering pcc_nn.h promises -- and the side stream, which waits for the build's k_grid_params only, would read them early.
    static bool wanted(const pcc_index* ix, size_t nq) {
        return ix->opt.overlap_prep && ix->stream == ix->own_stream && ix->after_build && ix->params_ev_set && ix->engine == PCC_ENGINE_GRID && ix->has_grid &&
               !ix->keep_order && (nq >= min_queries || ix->opt.overlap_prep == 2);  // (2: whatever the size -- tests, fuzz)
    }
    explicit PrepOverlap(pcc_index* i) : ix(i) {}
    void swap_scratch() {
        std::swap(ix->scratch_a, ix->side.a);
        std::swap(ix->scratch_b, ix->side.b);
        std::swap(ix->scratch_c, ix->side.c);
        std::swap(ix->scratch_e, ix->side.e);
        std::swap(ix->mp_a, ix->side.mp_a);
        std::swap(ix->mp_b, ix->side.mp_b);
        std::swap(ix->mp_c, ix->side.mp_c);
        std::swap(ix->scan_flags, ix->side.scan_flags);
        std::swap(ix->scan_epoch, ix->side.scan_epoch);
    }
    int begin() {
        if (!ix->side_stream) PCC_HIP(hipStreamCreateWithFlags(&ix->side_stream, hipStreamNonBlocking));
        if (!ix->side_ev) PCC_HIP(hipEventCreateWithFlags(&ix->side_ev, hipEventDisableTiming));
        PCC_HIP(hipStreamWaitEvent(ix->side_stream, ix->params_ev, 0));
        if (ix->edge_fresh) PCC_HIP(hipStreamWaitEvent(ix->side_stream, ix->edge_ev, 0));  // (the caller's producer stream)
        main_stream = ix->stream;
        ix->stream = ix->side_stream;
        swap_scratch();
        on = true;
        return PCC_OK;
    }
    int end() {
        if (!on) return PCC_OK;
        on = false;
        swap_scratch();
        ix->stream = main_stream;
        PCC_HIP(hipEventRecord(ix->side_ev, ix->side_stream));
        PCC_HIP(hipStreamWaitEvent(main_stream, ix->side_ev, 0));
        return PCC_OK;
    }
    ~PrepOverlap() { (void)end(); }
};

}  // namespace pcc

using namespace pcc;

extern "C" {

int pcc_version(void) { return PCC_VERSION; }
const char* pcc_last_error(void) { return g_err.c_str(); }

int pcc_device_count(int* count) {
    if (!count) { set_error("null count"); return PCC_ERR_INVALID; }
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return PCC_ERR_DEVICE; }
    *count = c;
    return PCC_OK;
}

int pcc_index_destroy(pcc_index* ix) {
    if (!ix) return PCC_OK;
    DeviceGuard g(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    DevBuf* bufs[] = {&ix->refs, &ix->cell_refs, &ix->cell_start, &ix->q_raw, &ix->q_packed, &ix->out_packed,
                      &ix->out_idx, &ix->out_d2, &ix->scratch_a, &ix->scratch_b, &ix->scratch_c, &ix->scratch_d, &ix->scratch_e, &ix->scratch_f, &ix->scratch_g,
                      &ix->small, &ix->blk_stats, &ix->icp_src, &ix->icp_state, &ix->d_grid, &ix->seeds, &ix->vox_a, &ix->vox_b, &ix->vox_c, &ix->tie_buf, &ix->knn_fb, &ix->occ, &ix->self_rows, &ix->flann_nodes, &ix->flann_leaf, &ix->mp_a, &ix->mp_b, &ix->mp_c, &ix->rows_idx, &ix->rows_d2, &ix->scan_flags, &ix->q_cells,
                      &ix->side.a, &ix->side.b, &ix->side.c, &ix->side.e, &ix->side.mp_a, &ix->side.mp_b, &ix->side.mp_c, &ix->side.scan_flags};
    for (DevBuf* b : bufs) b->release();
    for (int sl = 0; sl < PCC_EV_SLOTS; ++sl)
        for (int k = 0; k < PCC_EV_KINDS; ++k)
            if (ix->ev[sl][k]) (void)hipEventDestroy(ix->ev[sl][k]);
    ix->host_a.release();
    ix->host_b.release();
    ix->host_c.release();
    if (ix->pipe) { ix->pipe->release(); delete ix->pipe; ix->pipe = nullptr; }
    if (ix->pinned) (void)hipHostFree(ix->pinned);
    if (ix->h_grid) (void)hipHostFree(ix->h_grid);
    if (ix->edge_ev) (void)hipEventDestroy(ix->edge_ev);
    if (ix->params_ev) (void)hipEventDestroy(ix->params_ev);
    if (ix->side_ev) (void)hipEventDestroy(ix->side_ev);
    if (ix->side_stream) { (void)hipStreamSynchronize(ix->side_stream); (void)hipStreamDestroy(ix->side_stream); }
    if (ix->own_stream) (void)hipStreamDestroy(ix->own_stream);
    delete ix;
    return PCC_OK;
}

// shared by pcc_index_create and pcc_index_clone_to_device: an empty handle on `device`
static int new_handle(int device, int engine, pcc_index** out) {
    pcc_index* ix = new pcc_index();
    ix->device = device;
    ix->opt.from_env();
    DeviceGuard g(device);
    int st = PCC_OK;
    auto fail = [&](int s) { pcc_index_destroy(ix); return s; };
    if (!g.ok) { set_error("hipSetDevice(%d) failed", device); return fail(PCC_ERR_DEVICE); }
    if (hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(PCC_ERR_DEVICE); }
    ix->stream = ix->own_stream;
    if (hipHostMalloc(&ix->pinned, PACK_MAX_BLOCKS * 8 * sizeof(float) + 4096, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc failed"); return fail(PCC_ERR_DEVICE); }
    if ((st = ix->small.reserve(PCC_SMALL_BYTES)) != PCC_OK) return fail(st);
    if (hipMemset(ix->small.p, 0, PCC_SMALL_BYTES) != hipSuccess) { set_error("hipMemset failed"); return fail(PCC_ERR_DEVICE); }  // (the pack kernel's ticket word starts at 0)
    if ((st = ix->blk_stats.reserve(PACK_MAX_BLOCKS * 8 * sizeof(float))) != PCC_OK) return fail(st);
    ix->engine_requested = engine;
    ix->engine = engine;
    memset(ix->pinned, 0, PACK_MAX_BLOCKS * 8 * sizeof(float) + 4096);
    if (hipHostMalloc((void**)&ix->h_grid, sizeof(GridDev), hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc failed"); return fail(PCC_ERR_DEVICE); }
    memset(ix->h_grid, 0, sizeof(GridDev));
    *out = ix;
    return PCC_OK;
}

int pcc_index_create(const void* pts, size_t n, size_t stride, int dim, int mem, int device, int engine,
                     pcc_index** out) {
    if (!out) { set_error("null out"); return PCC_ERR_INVALID; }
    *out = nullptr;
    if (dim != 3) { set_error("dim %d unsupported: every hot call site of the reference searches 3 floats", dim); return PCC_ERR_UNSUPPORTED; }
    if (engine < PCC_ENGINE_AUTO || engine > PCC_ENGINE_GRID) { set_error("bad engine %d", engine); return PCC_ERR_INVALID; }
    PCC_TRY(check_points(pts, n, stride, mem));
    if (n == 0) { set_error("Cannot create a KDTree with an empty input cloud"); return PCC_ERR_EMPTY; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (libpcc_nn has no CPU path)"); return PCC_ERR_DEVICE; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d present)", device, ndev); return PCC_ERR_INVALID; }
    pcc_index* ix = nullptr;
    PCC_TRY(new_handle(device, engine, &ix));
    DeviceGuard g(device);
    int st = PCC_OK;
    auto fail = [&](int s) { pcc_index_destroy(ix); return s; };
    if ((st = set_input(ix, pts, n, stride, mem)) != PCC_OK) return fail(st);
    if ((st = sync_info(ix)) != PCC_OK) return fail(st);
    if (hipStreamSynchronize(ix->stream) != hipSuccess) { set_error("index build failed: %s", hipGetErrorString(hipGetLastError())); return fail(PCC_ERR_DEVICE); }
    if (ix->n_valid == 0) { set_error("Cannot create a KDTree with an empty input cloud (all %zu points non-finite)", n); return fail(PCC_ERR_EMPTY); }
    *out = ix;
    return PCC_OK;
}

int pcc_index_clone_to_device(pcc_index* src, int device, pcc_index** out) {
    return pcc_index_clone_to_devices(src, &device, 1, out);
}

int pcc_index_clone_to_devices(pcc_index* src, const int* devices, int count, pcc_index** out) {
    if (!out || !devices || count < 0) { set_error("null argument"); return PCC_ERR_INVALID; }
    for (int k = 0; k < count; ++k) out[k] = nullptr;
    if (!src) { set_error("null index"); return PCC_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (libpcc_nn has no CPU path)"); return PCC_ERR_DEVICE; }
    for (int k = 0; k < count; ++k)
        if (devices[k] < 0 || devices[k] >= ndev) { set_error("device %d out of range (%d present)", devices[k], ndev); return PCC_ERR_INVALID; }
    if (count == 0) return PCC_OK;
    // `src` stays locked until every peer copy has landed: a concurrent pcc_index_set_input on it may free or rewrite
    // the packed cloud the copies read
    std::lock_guard<std::mutex> lock(src->mu);
    {
        DeviceGuard g(src->device);
        if (!g.ok) { set_error("hipSetDevice(%d) failed", src->device); return PCC_ERR_DEVICE; }
        if (src->n_orig == 0) { set_error("index is empty"); return PCC_ERR_EMPTY; }
        PCC_HIP(hipStreamSynchronize(src->stream));  // the packed cloud is complete
    }
    const size_t n = src->n_orig;
    // One host thread per clone: handle creation (stream, pinned blocks, device allocations: ~3 ms of driver calls
    // each -- more than the copy itself), the peer copy on the clone's own stream over its own xGMI link, the build
    // behind it, the join.  Nothing of one clone waits for another: the copies to the other GPUs of a node and their
    // builds all overlap (one after the other: 7 x (3 ms + 160 MB + build) at 10M points).
    std::vector<int> status((size_t)count, PCC_OK);
    std::vector<std::string> errors((size_t)count);
    auto one = [&](int k) {
        auto run = [&]() -> int {
            pcc_index* ix = nullptr;
            PCC_TRY(new_handle(devices[k], src->engine_requested, &ix));
            out[k] = ix;
            ix->opt = src->opt;
            ix->tie_mode = src->tie_mode;
            DeviceGuard g(devices[k]);
            PCC_TRY(ix->icp_src.reserve(n * sizeof(float4)));
            if (hipMemcpyPeerAsync(ix->icp_src.p, devices[k], src->refs.p, src->device, n * sizeof(float4), ix->stream) != hipSuccess) {
                set_error("hipMemcpyPeerAsync %d -> %d failed: %s", src->device, devices[k], hipGetErrorString(hipGetLastError()));
                return PCC_ERR_DEVICE;
            }
            // (non-finite points get their NaN back so that the build sees what the original upload saw)
            PCC_TRY(launch_nanify(ix->stream, ix->icp_src.as<float4>(), n));
            PCC_TRY(set_input(ix, ix->icp_src.p, n, sizeof(float4), PCC_MEM_DEVICE));
            PCC_TRY(sync_info(ix));
            if (hipStreamSynchronize(ix->stream) != hipSuccess) { set_error("index build failed: %s", hipGetErrorString(hipGetLastError())); return PCC_ERR_DEVICE; }
            return PCC_OK;
        };
        status[(size_t)k] = run();
        if (status[(size_t)k] != PCC_OK) errors[(size_t)k] = g_err;  // (the message is thread-local)
    };
    if (count == 1) {
        one(0);
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < count; ++k) th.emplace_back(one, k);
        for (std::thread& t : th) t.join();
    }
    for (int k = 0; k < count; ++k)
        if (status[(size_t)k] != PCC_OK) {
            const int bad = status[(size_t)k];
            const std::string msg = errors[(size_t)k];
            for (int j = 0; j < count; ++j) { if (out[j]) pcc_index_destroy(out[j]); out[j] = nullptr; }
            set_error("%s", msg.c_str());
            return bad;
        }
    return PCC_OK;
}

int pcc_index_set_input(pcc_index* ix, const void* pts, size_t n, size_t stride, int dim, int mem) {
    PCC_ENTER(ix);
    if (dim != 3) { set_error("dim %d unsupported", dim); return PCC_ERR_UNSUPPORTED; }
    PCC_TRY(check_points(pts, n, stride, mem));
    if (n == 0) { ix->n_valid = 0; ix->n_orig = 0; set_error("Cannot create a KDTree with an empty input cloud"); return PCC_ERR_EMPTY; }
    // asynchronous: a cloud without any finite point is reported by pcc_index_size (0) and by
    // searches returning idx = -1, not by this call
    return set_input(ix, pts, n, stride, mem);
}

int pcc_index_enable_timing(pcc_index* ix, int on) {
    PCC_ENTER(ix);
    PCC_NOTHING_ENQUEUED(ix);
    PCC_HIP(hipStreamSynchronize(ix->stream));
    if (on) {
        for (int sl = 0; sl < PCC_EV_SLOTS; ++sl)
            for (int k = 0; k < PCC_EV_KINDS; ++k)
                if (!ix->ev[sl][k]) PCC_HIP(hipEventCreate(&ix->ev[sl][k]));
    }
    for (int sl = 0; sl < PCC_EV_SLOTS; ++sl)
        for (int k = 0; k < PCC_EV_KINDS; ++k) ix->ev_rec[sl][k] = false;
    ix->ev_slot = 0;
    ix->timing = on < 0 ? 0 : (on > 2 ? 2 : on);
    return PCC_OK;
}

int pcc_index_timing(pcc_index* ix, float ms[8]) {
    PCC_ENTER(ix);
    PCC_NOTHING_ENQUEUED(ix);
    if (!ms) { set_error("null ms"); return PCC_ERR_INVALID; }
    PCC_HIP(hipStreamSynchronize(ix->stream));
    const int pairs[5][2] = {{EV_MAIN0, EV_MAIN1}, {EV_FB0, EV_FB1}, {EV_CALL0, EV_CALL1}, {EV_BUILD0, EV_BUILD1}, {EV_SORT0, EV_SORT1}};
    for (int i = 0; i < 8; ++i) ms[i] = 0.f;
    for (int i = 0; i < 5; ++i) {
        double sum = 0;
        int cnt = 0;
        for (int sl = 0; sl < PCC_EV_SLOTS; ++sl) {
            int a = pairs[i][0], b = pairs[i][1];
            if (ix->ev[sl][a] && ix->ev[sl][b] && ix->ev_rec[sl][a] && ix->ev_rec[sl][b]) {
                float t = 0.f;
                if (hipEventElapsedTime(&t, ix->ev[sl][a], ix->ev[sl][b]) == hipSuccess) { sum += t; ++cnt; }
            }
        }
        if (cnt) ms[i] = (float)(sum / cnt);
        if (i == 0) ms[7] = (float)cnt;
    }
    return PCC_OK;
}

int pcc_index_size(const pcc_index* cix, size_t* n_valid) {
    if (!cix || !n_valid) { set_error("null argument"); return PCC_ERR_INVALID; }
    pcc_index* ix = const_cast<pcc_index*>(cix);  // may have to wait for the asynchronous build
    PCC_ENTER(ix);
    PCC_NOTHING_ENQUEUED(ix);
    PCC_TRY(sync_info(ix));
    *n_valid = ix->n_valid;
    return PCC_OK;
}
int pcc_index_set_stream(pcc_index* ix, void* s) {
    PCC_ENTER(ix);
    PCC_HIP(hipStreamSynchronize(ix->stream));
    ix->stream = s ? static_cast<hipStream_t>(s) : ix->own_stream;
    return PCC_OK;
}
// order the index's stream against another stream of the same device without blocking the host
static int stream_edge(pcc_index* ix, hipStream_t from, hipStream_t to) {
    if (from == to) return PCC_OK;
    if (!ix->edge_ev) PCC_HIP(hipEventCreateWithFlags(&ix->edge_ev, hipEventDisableTiming));
    PCC_HIP(hipEventRecord(ix->edge_ev, from));
    PCC_HIP(hipStreamWaitEvent(to, ix->edge_ev, 0));
    return PCC_OK;
}
int pcc_index_wait_stream(pcc_index* ix, void* producer) {
    PCC_ENTER(ix);
    PCC_TRY(stream_edge(ix, static_cast<hipStream_t>(producer), ix->stream));
    // (a search staged beside the build -- PrepOverlap -- must honour this edge on its side stream as well; edge_ev is the
    // producer's mark until the next edge, and one such edge is remembered)
    if (ix->after_build && !ix->edge_fresh && static_cast<hipStream_t>(producer) != ix->stream) {
        ix->build_fresh = true;
        ix->edge_fresh = true;
    }
    return PCC_OK;
}
int pcc_stream_wait_index(pcc_index* ix, void* consumer) {
    PCC_ENTER(ix);
    PCC_NOTHING_ENQUEUED(ix);
    return stream_edge(ix, ix->stream, static_cast<hipStream_t>(consumer));
}
int pcc_index_sync(pcc_index* ix) {
    PCC_ENTER(ix);
    PCC_NOTHING_ENQUEUED(ix);
    PCC_HIP(hipStreamSynchronize(ix->stream));
    return PCC_OK;
}
int pcc_index_engine(const pcc_index* ix, int* engine) {
    if (!ix || !engine) { set_error("null argument"); return PCC_ERR_INVALID; }
    *engine = ix->engine;
    return PCC_OK;
}
int pcc_index_set_engine(pcc_index* ix, int engine) {
    PCC_ENTER(ix);
    if (engine < PCC_ENGINE_AUTO || engine > PCC_ENGINE_GRID) { set_error("bad engine %d", engine); return PCC_ERR_INVALID; }
    ix->engine_requested = engine;
    engine = resolve_engine(engine, ix->n_orig);
    if (engine == PCC_ENGINE_GRID && !ix->has_grid && ix->n_orig) PCC_TRY(grid_build(ix));
    ix->engine = engine;
    return PCC_OK;
}
int pcc_index_set_tie_order(pcc_index* ix, int ties) {
    PCC_ENTER(ix);
    PCC_NOTHING_ENQUEUED(ix);
    if (ties != PCC_TIES_LOWEST_INDEX && ties != PCC_TIES_FLANN) { set_error("bad tie order %d", ties); return PCC_ERR_INVALID; }
    ix->tie_mode = ties;
    return PCC_OK;
}
int pcc_index_set_option(pcc_index* ix, int option, double value) {
    PCC_ENTER(ix);
    PCC_NOTHING_ENQUEUED(ix);
    int* pi = nullptr;
    double* pd = option_slot(ix->opt, option, &pi);
    if (!pd && !pi) { set_error("unknown option %d", option); return PCC_ERR_INVALID; }
    if (!std::isfinite(value)) { set_error("option %d: non-finite value", option); return PCC_ERR_INVALID; }
    if (!option_in_range(option, value)) { set_error("option %d: value %g out of range", option, value); return PCC_ERR_INVALID; }
    if (pd) *pd = value; else *pi = (int)value;
    if (option == PCC_OPT_FLANN_SPLIT) ix->flann_valid = false;  // the replayed tree has to be rebuilt with the other rule
    return PCC_OK;
}
int pcc_index_get_option(pcc_index* ix, int option, double* value) {
    PCC_ENTER(ix);
    PCC_NOTHING_ENQUEUED(ix);
    if (!value) { set_error("null value"); return PCC_ERR_INVALID; }
    int* pi = nullptr;
    double* pd = option_slot(ix->opt, option, &pi);
    if (!pd && !pi) { set_error("unknown option %d", option); return PCC_ERR_INVALID; }
    *value = pd ? *pd : (double)*pi;
    return PCC_OK;
}
int pcc_debug_fail_alloc(int nth) {
    g_fail_alloc.store(nth > 0 ? nth : 0);
    return PCC_OK;
}
int pcc_counts_pairs(void) {
#ifdef PCC_COUNT_PAIRS
    return 1;
#else
    return 0;
#endif
}
int pcc_index_stats(const pcc_index* cix, uint64_t stats[8]) {
    if (!cix || !stats) { set_error("null argument"); return PCC_ERR_INVALID; }
    pcc_index* ix = const_cast<pcc_index*>(cix);
    PCC_ENTER(ix);
    PCC_TRY(sync_info(ix));
#ifdef PCC_COUNT_PAIRS
    // profiling build: distances evaluated by the pruned kernels (process-wide, every handle) since the previous call
    PCC_HIP(hipStreamSynchronize(ix->stream));
    ix->stats[4] = pairs_take_grid() + pairs_take_knn() + pairs_take_cluster() + pairs_take_flann();
#endif
    if (ix->stats_pending) {
        PCC_HIP(hipStreamSynchronize(ix->stream));
        ix->stats_pending = false;
        ix->stats[1] = static_cast<unsigned int*>(ix->pinned)[40];
        ix->stats[0] = ix->last_nq - ix->stats[1];
    }
    if (ix->ties_pending) {  // the sharded counters of the last search in FLANN mode
        unsigned int h[PCC_TIE_SHARDS * PCC_OPEN_CTR_STRIDE];
        PCC_HIP(hipMemcpyAsync(h, ix->small.as<unsigned int>() + PCC_TIE_CTR0, sizeof(h), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipStreamSynchronize(ix->stream));
        ix->ties_flagged = ix->ties_changed = 0;
        for (int sh = 0; sh < PCC_TIE_SHARDS; ++sh) {
            ix->ties_flagged += h[sh * PCC_OPEN_CTR_STRIDE];
            ix->ties_changed += h[sh * PCC_OPEN_CTR_STRIDE + 1];
        }
        ix->ties_pending = false;
    }
    if (ix->open_pending) {  // lanes the 3x3x3 cube left open in the last listed k = 1 search (sharded counters)
        unsigned int h[PCC_OPEN_SHARDS * PCC_OPEN_CTR_STRIDE];
        PCC_HIP(hipMemcpyAsync(h, ix->small.as<unsigned int>() + PCC_OPEN_CTR0, sizeof(h), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipStreamSynchronize(ix->stream));
        ix->stats[7] = 0;
        for (int sh = 0; sh < PCC_OPEN_SHARDS; ++sh) ix->stats[7] += h[sh * PCC_OPEN_CTR_STRIDE];
        ix->open_pending = false;
    }
    ix->stats[5] = ix->ties_flagged;
    ix->stats[6] = ix->ties_changed;
    memcpy(stats, ix->stats, sizeof(ix->stats));
    return PCC_OK;
}

int pcc_nn1(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem, int32_t* idx, float* d2) {
    PCC_ENTER(ix);
    PCC_TRY(check_points(q, nq, stride, mem));
    if (nq == 0) return PCC_OK;
    if (ix->n_orig == 0) { set_error("index is empty"); return PCC_ERR_EMPTY; }
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    if (small_call(ix, nq, stride, mem)) {
        PCC_TRY(small_nn1(ix, q, nq, stride, idx != nullptr, d2 != nullptr));
        if (idx) memcpy(idx, ix->host_a.p, nq * sizeof(int32_t));
        if (d2) memcpy(d2, ix->host_b.p, nq * sizeof(float));
        return PCC_OK;
    }
    {
        PrepOverlap beside(ix);
        if (PrepOverlap::wanted(ix, nq)) PCC_TRY(beside.begin());
        PCC_TRY(stage_queries(ix, q, nq, stride, mem));
        if (beside.on) {
            PCC_TRY(grid_sort_queries(ix, ix->q_packed.as<float4>(), nq, &ix->pre_order, &ix->pre_nsorted));
            ix->pre_order_nq = nq;
            PCC_TRY(beside.end());
        }
    }
    PCC_TRY(nn1_packed(ix, nq));
    if (ix->tie_mode == PCC_TIES_FLANN)
        PCC_TRY(resolve_ties_flann(ix, ix->q_packed.as<float4>(), ix->out_packed.as<unsigned long long>(), nq, mem == PCC_MEM_HOST));
    int32_t* didx = idx;
    float* dd2 = d2;
    // small host results: the unpack kernel writes them into pinned host memory itself (no copy command, one wait)
    const bool direct = mem == PCC_MEM_HOST && ix->opt.host_pipe && nq * sizeof(float) <= SMALL_RESULT_BYTES;
    if (direct) {
        PCC_TRY(ix->host_a.reserve(nq * sizeof(int32_t)));
        PCC_TRY(ix->host_b.reserve(nq * sizeof(float)));
        didx = idx ? ix->host_a.as<int32_t>() : nullptr;
        dd2 = d2 ? ix->host_b.as<float>() : nullptr;
    } else if (mem == PCC_MEM_HOST) {
        PCC_TRY(ix->out_idx.reserve(nq * sizeof(int32_t)));
        PCC_TRY(ix->out_d2.reserve(nq * sizeof(float)));
        didx = idx ? ix->out_idx.as<int32_t>() : nullptr;
        dd2 = d2 ? ix->out_d2.as<float>() : nullptr;
    }
    PCC_TRY(launch_unpack(ix->stream, ix->out_packed.as<unsigned long long>(), nullptr, nq, didx, dd2,
                          ix->small.as<unsigned int>() + 32, static_cast<unsigned int*>(ix->pinned) + 40));
    ev_mark(ix, EV_CALL1);
    if (direct) {
        PCC_HIP(hipStreamSynchronize(ix->stream));
        if (idx) memcpy(idx, didx, nq * sizeof(int32_t));
        if (d2) memcpy(d2, dd2, nq * sizeof(float));
    } else if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, didx, idx, nq, mem));
        PCC_TRY(deliver(ix, dd2, d2, nq, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

// the searches below need the cell grid whatever engine k=1 uses
static int ensure_grid(pcc_index* ix) {
    if (ix->n_orig == 0) { set_error("index is empty"); return PCC_ERR_EMPTY; }
    if (!ix->has_grid) PCC_TRY(grid_build(ix));
    return PCC_OK;
}

int pcc_knn(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem, int k, int32_t* idx, float* d2) {
    PCC_ENTER(ix);
    PCC_TRY(check_points(q, nq, stride, mem));
    if (k < 1 || k > PCC_KNN_MAX_K) { set_error("k=%d outside [1, %d]", k, PCC_KNN_MAX_K); return PCC_ERR_UNSUPPORTED; }
    if (nq == 0) return PCC_OK;
    PCC_TRY(ensure_grid(ix));
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    PCC_TRY(stage_queries(ix, q, nq, stride, mem));
    int32_t* didx = idx;
    float* dd2 = d2;
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(ix->out_idx.reserve(nq * (size_t)k * sizeof(int32_t)));
        PCC_TRY(ix->out_d2.reserve(nq * (size_t)k * sizeof(float)));
        didx = idx ? ix->out_idx.as<int32_t>() : nullptr;
        dd2 = d2 ? ix->out_d2.as<float>() : nullptr;
    }
    if (grid_knn_delivers(k)) {
        // the search writes indices and distances itself (no key array, no unpack pass)
        PCC_TRY(grid_knn(ix, ix->q_packed.as<float4>(), nq, k, nullptr, didx, dd2));
    } else {
        PCC_TRY(ix->out_packed.reserve(nq * (size_t)k * sizeof(unsigned long long)));
        auto* keys = ix->out_packed.as<unsigned long long>();
        PCC_TRY(grid_knn(ix, ix->q_packed.as<float4>(), nq, k, keys));
        // rows of invalid queries were never touched (all ~0) and unpack to -1 / +inf
        PCC_TRY(launch_unpack(ix->stream, keys, nullptr, nq * (size_t)k, didx, dd2));
    }
    ev_mark(ix, EV_CALL1);
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, didx, idx, nq * (size_t)k, mem));
        PCC_TRY(deliver(ix, dd2, d2, nq * (size_t)k, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

// radiusSearch(pt, double radius): r2 = float(radius*radius) evaluated in double
static inline float radius2(double radius) { return (float)(radius * radius); }

int pcc_radius_count(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem, double radius, int32_t* counts) {
    return pcc_radius_count_max(ix, q, nq, stride, mem, radius, 0u, counts);
}
// radiusSearch's max_nn, decided under the handle's lock for count and fill alike: "all" when it is 0 or reaches the number of
// FINITE indexed points (PCL's total_nr_points_); beyond PCC_KNN_MAX_K both calls refuse (the rows come from the k-NN kernels)
static int radius_max_mode(pcc_index* ix, unsigned int max_nn, bool* all) {
    PCC_TRY(sync_info(ix));
    *all = max_nn == 0 || (size_t)max_nn >= ix->n_valid;
    if (!*all && max_nn > (unsigned int)PCC_KNN_MAX_K) { set_error("max_nn=%u beyond %d", max_nn, PCC_KNN_MAX_K); return PCC_ERR_UNSUPPORTED; }
    return PCC_OK;
}
static int radius_fill_impl(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem, double radius, int sorted,
                            const int64_t* offsets, int32_t* idx, float* d2);

int pcc_radius_count_max(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem, double radius, unsigned int max_nn,
                         int32_t* counts) {
    PCC_ENTER(ix);
    PCC_TRY(check_points(q, nq, stride, mem));
    if (!counts) { set_error("null counts"); return PCC_ERR_INVALID; }
    if (!(radius >= 0)) { set_error("bad radius"); return PCC_ERR_INVALID; }
    if (nq == 0) return PCC_OK;
    PCC_TRY(ensure_grid(ix));
    bool max_nn_is_all = true;
    PCC_TRY(radius_max_mode(ix, max_nn, &max_nn_is_all));  // (the same decision, and the same refusal, as the fill's)
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    PCC_TRY(stage_queries(ix, q, nq, stride, mem));
    int32_t* dcnt = counts;
    if (mem == PCC_MEM_HOST) { PCC_TRY(ix->out_idx.reserve(nq * sizeof(int32_t))); dcnt = ix->out_idx.as<int32_t>(); }
    PCC_HIP(hipMemsetAsync(dcnt, 0, nq * sizeof(int32_t), ix->stream));
    PCC_TRY(grid_radius(ix, ix->q_packed.as<float4>(), nq, (float)radius, radius2(radius), dcnt, nullptr, nullptr, 0));
    // KdTreeFLANN::radiusSearch(..., max_nn): 0 or anything from the cloud's size on means "all" -- the size PCL compares
    // with is total_nr_points_, the FINITE points --; else FLANN keeps the max_nn nearest within the radius
    if (!max_nn_is_all) PCC_TRY(launch_clamp_counts(ix->stream, dcnt, nq, (int32_t)max_nn));
    ev_mark(ix, EV_CALL1);
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, dcnt, counts, nq, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

int pcc_radius_fill_max(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem, double radius, int sorted, unsigned int max_nn,
                        const int64_t* offsets, int32_t* idx, float* d2) {
    // the max_nn NEAREST within the radius, ascending (FLANN's KNNRadiusResultSet, whatever `sorted` says): the k-NN rows
    // with k = max_nn, cut at the radius.  Unused by the reference's own call sites (src/segmentation.cpp:125-131 passes 0):
    // correctness first, no kernel of its own
    PCC_ENTER(ix);
    PCC_TRY(check_points(q, nq, stride, mem));
    if (!offsets) { set_error("null offsets"); return PCC_ERR_INVALID; }
    if (!(radius >= 0)) { set_error("bad radius"); return PCC_ERR_INVALID; }
    if (nq == 0) return PCC_OK;
    PCC_TRY(ensure_grid(ix));
    bool max_nn_is_all = true;
    PCC_TRY(radius_max_mode(ix, max_nn, &max_nn_is_all));
    if (max_nn_is_all) return radius_fill_impl(ix, q, nq, stride, mem, radius, sorted, offsets, idx, d2);
    PCC_TRY(stage_queries(ix, q, nq, stride, mem));
    const int K = (int)max_nn;
    int64_t total = 0;
    const int64_t* doff = offsets;
    if (mem == PCC_MEM_HOST) {
        total = offsets[nq];
        PCC_TRY(ix->scratch_d.reserve((nq + 1) * sizeof(int64_t)));
        PCC_HIP(hipMemcpyAsync(ix->scratch_d.p, offsets, (nq + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ix->stream));
        doff = ix->scratch_d.as<int64_t>();
    } else {
        PCC_HIP(hipMemcpyAsync(&total, offsets + nq, sizeof(int64_t), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    if (total < 0) { set_error("negative total"); return PCC_ERR_INVALID; }
    if (total == 0) return PCC_OK;
    const bool rows = grid_knn_delivers(K);
    unsigned long long* keys = nullptr;
    int32_t* ridx = nullptr;
    float* rd2 = nullptr;
    if (rows) {
        // (buffers of their own: the query sort inside grid_knn re-reserves scratch_e as its pair buffer, and a reserve
        // that grows FREES the old block -- with the rows in scratch_e a fresh handle, nq = 1 or max_nn = 1 left the k-NN
        // kernels writing into freed memory)
        PCC_TRY(ix->rows_idx.reserve(nq * (size_t)K * sizeof(int32_t)));
        PCC_TRY(ix->rows_d2.reserve(nq * (size_t)K * sizeof(float)));
        ridx = ix->rows_idx.as<int32_t>();
        rd2 = ix->rows_d2.as<float>();
        PCC_TRY(grid_knn(ix, ix->q_packed.as<float4>(), nq, K, nullptr, ridx, rd2));
    } else {
        PCC_TRY(ix->out_packed.reserve(nq * (size_t)K * sizeof(unsigned long long)));
        keys = ix->out_packed.as<unsigned long long>();
        PCC_TRY(grid_knn(ix, ix->q_packed.as<float4>(), nq, K, keys));
    }
    int32_t* didx = idx;
    float* dd2 = d2;
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(ix->out_idx.reserve((size_t)total * sizeof(int32_t)));
        PCC_TRY(ix->out_d2.reserve((size_t)total * sizeof(float)));
        didx = idx ? ix->out_idx.as<int32_t>() : nullptr;
        dd2 = d2 ? ix->out_d2.as<float>() : nullptr;
    }
    PCC_TRY(launch_knn_rows_to_csr(ix->stream, keys, ridx, rd2, K, radius2(radius), doff, nq, didx, dd2));
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, didx, idx, (size_t)total, mem));
        PCC_TRY(deliver(ix, dd2, d2, (size_t)total, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

int pcc_radius_fill(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem, double radius, int sorted,
                    const int64_t* offsets, int32_t* idx, float* d2) {
    PCC_ENTER(ix);
    return radius_fill_impl(ix, q, nq, stride, mem, radius, sorted, offsets, idx, d2);
}
// (the caller holds the handle's lock)
static int radius_fill_impl(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem, double radius, int sorted,
                            const int64_t* offsets, int32_t* idx, float* d2) {
    PCC_TRY(check_points(q, nq, stride, mem));
    if (!offsets) { set_error("null offsets"); return PCC_ERR_INVALID; }
    if (nq == 0) return PCC_OK;
    PCC_TRY(ensure_grid(ix));
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    PCC_TRY(stage_queries(ix, q, nq, stride, mem));
    // total = offsets[nq]; offsets live in the caller's memory space
    int64_t total = 0;
    const int64_t* doff = offsets;
    if (mem == PCC_MEM_HOST) {
        total = offsets[nq];
        PCC_TRY(ix->scratch_d.reserve((nq + 1) * sizeof(int64_t)));
        PCC_HIP(hipMemcpyAsync(ix->scratch_d.p, offsets, (nq + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ix->stream));
        doff = ix->scratch_d.as<int64_t>();
    } else {
        PCC_HIP(hipMemcpyAsync(&total, offsets + nq, sizeof(int64_t), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    if (total < 0) { set_error("negative total"); return PCC_ERR_INVALID; }
    if (total == 0) return PCC_OK;
    PCC_TRY(ix->out_packed.reserve((size_t)total * sizeof(unsigned long long)));
    auto* keys = ix->out_packed.as<unsigned long long>();
    int32_t* didx = idx;
    float* dd2 = d2;
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(ix->out_idx.reserve((size_t)total * sizeof(int32_t)));
        PCC_TRY(ix->out_d2.reserve((size_t)total * sizeof(float)));
        didx = idx ? ix->out_idx.as<int32_t>() : nullptr;
        dd2 = d2 ? ix->out_d2.as<float>() : nullptr;
    }
    // rows of 24 neighbours and more on average: the fill delivers index and distance itself (sorted in registers, no
    // key array in between); otherwise keys -> sort -> unpack.  (Slots a fill does not reach read "nothing found".)
    const bool wave_fill = (size_t)total >= 24 * nq && (didx || dd2);
    if (!wave_fill) PCC_HIP(hipMemsetAsync(keys, 0xff, (size_t)total * sizeof(unsigned long long), ix->stream));
    bool delivered = false;
    PCC_TRY(grid_radius(ix, ix->q_packed.as<float4>(), nq, (float)radius, radius2(radius), nullptr, doff, keys, sorted, (size_t)total,
                        didx, dd2, &delivered));
    if (!delivered) PCC_TRY(launch_unpack(ix->stream, keys, nullptr, (size_t)total, didx, dd2));
    ev_mark(ix, EV_CALL1);
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, didx, idx, (size_t)total, mem));
        PCC_TRY(deliver(ix, dd2, d2, (size_t)total, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

int pcc_voxel_grid(pcc_index* ix, const void* pts, size_t n, size_t stride, int mem, float leaf, int has_rgb,
                   void* out, size_t out_stride, size_t* out_n) {
    PCC_ENTER(ix);
    PCC_TRY(check_points(pts, n, stride, mem));
    if (!out || !out_n) { set_error("null output"); return PCC_ERR_INVALID; }
    if (out_stride < 12 || out_stride % 4) { set_error("bad output stride"); return PCC_ERR_INVALID; }
    if (has_rgb && (stride < 20 || out_stride < 20)) { set_error("rgb needs a stride of at least 20 bytes"); return PCC_ERR_INVALID; }
    if (!(leaf > 0.f)) { set_error("leaf size must be positive"); return PCC_ERR_INVALID; }
    *out_n = 0;
    if (n == 0) return PCC_OK;
    return voxel_grid(ix, pts, n, stride, mem, leaf, has_rgb, out, out_stride, out_n);
}

int pcc_sac_plane(pcc_index* ix, const void* pts, size_t n, size_t stride, int mem, int max_iterations,
                  double threshold, double probability, int optimize, int32_t* inliers, size_t* n_inliers,
                  float coeff[4], int* iterations) {
    PCC_ENTER(ix);
    PCC_TRY(check_points(pts, n, stride, mem));
    if (!n_inliers || !coeff || (n && !inliers)) { set_error("null output"); return PCC_ERR_INVALID; }
    if (max_iterations < 0 || !(threshold >= 0) || !(probability > 0 && probability < 1)) {
        set_error("bad RANSAC parameters");
        return PCC_ERR_INVALID;
    }
    *n_inliers = 0;
    coeff[0] = coeff[1] = coeff[2] = coeff[3] = 0.f;
    if (iterations) *iterations = 0;
    if (n == 0) return PCC_OK;
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    PCC_TRY(stage_queries(ix, pts, n, stride, mem));
    const float4* dp = ix->q_packed.as<float4>();
    // the sampling and the refit read single points on the host: from the caller's array when it is a host array; for a
    // cloud in device memory the few points needed are gathered there (sac.hip) -- round 3 copied the whole cloud back
    // (16 B x n per call; the -e plane-removal loop calls this a handful of times per cloud)
    int32_t* di = inliers;
    if (mem == PCC_MEM_HOST) { PCC_TRY(ix->out_idx.reserve(n * sizeof(int32_t))); di = ix->out_idx.as<int32_t>(); }
    size_t m = 0;
    int st = sac_plane(ix, dp, n, mem == PCC_MEM_HOST ? static_cast<const char*>(pts) : nullptr, stride, max_iterations, threshold,
                       probability, optimize, di, &m, coeff, iterations);
    if (st == PCC_ERR_RETRY_HOST) {
        // a degenerate sample (PCL redraws it at once, which the gathered form cannot replay): with a host copy of the cloud
        std::vector<float4> hp(n);
        PCC_HIP(hipMemcpyAsync(hp.data(), dp, n * sizeof(float4), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipStreamSynchronize(ix->stream));
        const float qnan = std::nanf("");
        for (size_t i = 0; i < n; ++i)  // the staged copy zeroes non-finite points; PCL would see them as they are
            if (__builtin_bit_cast(int, hp[i].w) < 0) hp[i].x = hp[i].y = hp[i].z = qnan;
        st = sac_plane(ix, dp, n, reinterpret_cast<const char*>(hp.data()), sizeof(float4), max_iterations, threshold, probability,
                       optimize, di, &m, coeff, iterations);
    }
    PCC_TRY(st);
    ev_mark(ix, EV_CALL1);
    if (mem == PCC_MEM_HOST && m) {
        PCC_TRY(deliver(ix, di, inliers, m, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    *n_inliers = m;
    return PCC_OK;
}

// The self k-NN rows (keys) of the indexed cloud with k neighbours, for pcc_normals / pcc_region_growing.  With
// PCC_OPT_KNN_CACHE_K the rows are searched with at least that many neighbours and kept until the next set_input; a request
// the kept rows cover is their prefix (k-NN rows are ascending), copied row by row -- 0.15 ms at 1M x 50 against a 1.1 ms search.
static int self_knn_keys(pcc_index* ix, int k, const unsigned long long** keys) {
    const size_t n = ix->n_orig;
    const int want_kept = ix->opt.knn_cache_k > 0 ? std::max(k, ix->opt.knn_cache_k) : 0;
    if (ix->self_rows_k < k && want_kept > 0 && want_kept <= PCC_KNN_MAX_K) {
        ix->self_rows_k = 0;
        PCC_TRY(ix->self_rows.reserve(n * (size_t)want_kept * sizeof(unsigned long long)));
        PCC_TRY(grid_knn(ix, ix->refs.as<float4>(), n, want_kept, ix->self_rows.as<unsigned long long>()));
        ix->self_rows_k = want_kept;
    }
    if (ix->self_rows_k >= k) {
        if (ix->self_rows_k == k) { *keys = ix->self_rows.as<unsigned long long>(); return PCC_OK; }
        PCC_TRY(ix->out_packed.reserve(n * (size_t)k * sizeof(unsigned long long)));
        PCC_TRY(launch_copy_row_prefix(ix->stream, ix->self_rows.as<unsigned long long>(), ix->self_rows_k,
                                       ix->out_packed.as<unsigned long long>(), k, n));
        *keys = ix->out_packed.as<unsigned long long>();
        return PCC_OK;
    }
    PCC_TRY(ix->out_packed.reserve(n * (size_t)k * sizeof(unsigned long long)));
    PCC_TRY(grid_knn(ix, ix->refs.as<float4>(), n, k, ix->out_packed.as<unsigned long long>()));
    *keys = ix->out_packed.as<unsigned long long>();
    return PCC_OK;
}

int pcc_normals(pcc_index* ix, int k, const float viewpoint[3], int mem, float* out) {
    PCC_ENTER(ix);
    if (mem != PCC_MEM_HOST && mem != PCC_MEM_DEVICE) { set_error("bad mem space"); return PCC_ERR_INVALID; }
    if (!out) { set_error("null output"); return PCC_ERR_INVALID; }
    if (k < 1 || k > PCC_KNN_MAX_K) { set_error("k=%d outside [1, %d]", k, PCC_KNN_MAX_K); return PCC_ERR_UNSUPPORTED; }
    PCC_TRY(ensure_grid(ix));
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    const size_t n = ix->n_orig;
    const float origin[3] = {0.f, 0.f, 0.f};
    // self query on the packed references, as pcc_sor does
    const unsigned long long* keys = nullptr;
    PCC_TRY(self_knn_keys(ix, k, &keys));
    float4* dout = reinterpret_cast<float4*>(out);
    if (mem == PCC_MEM_HOST) { PCC_TRY(ix->out_d2.reserve(n * sizeof(float4))); dout = ix->out_d2.as<float4>(); }
    PCC_TRY(launch_normals(ix->stream, keys, ix->refs.as<float4>(), ix->cell_refs.as<float4>(), ix->d_grid.as<GridDev>(), n, k,
                           viewpoint ? viewpoint : origin, dout));
    ev_mark(ix, EV_CALL1);
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, reinterpret_cast<const float*>(dout), out, n * 4, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

int pcc_normals_radius(pcc_index* ix, double radius, const float viewpoint[3], int mem, float* out) {
    PCC_ENTER(ix);
    if (mem != PCC_MEM_HOST && mem != PCC_MEM_DEVICE) { set_error("bad mem space"); return PCC_ERR_INVALID; }
    if (!out) { set_error("null output"); return PCC_ERR_INVALID; }
    if (!(radius >= 0)) { set_error("bad radius"); return PCC_ERR_INVALID; }
    PCC_TRY(ensure_grid(ix));
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    const size_t n = ix->n_orig;
    const float origin[3] = {0.f, 0.f, 0.f};
    float4* dout = reinterpret_cast<float4*>(out);
    if (mem == PCC_MEM_HOST) { PCC_TRY(ix->out_d2.reserve(n * sizeof(float4))); dout = ix->out_d2.as<float4>(); }
    PCC_TRY(normals_radius(ix, radius, viewpoint ? viewpoint : origin, dout));
    ev_mark(ix, EV_CALL1);
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, reinterpret_cast<const float*>(dout), out, n * 4, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

int pcc_region_growing(pcc_index* ix, const float* normals, int mem, int k, float smoothness,
                       float curvature_threshold, uint32_t min_size, uint32_t max_size, int32_t* labels,
                       int32_t* n_clusters) {
    PCC_ENTER(ix);
    if (mem != PCC_MEM_HOST && mem != PCC_MEM_DEVICE) { set_error("bad mem space"); return PCC_ERR_INVALID; }
    if (!normals || !labels || !n_clusters) { set_error("null argument"); return PCC_ERR_INVALID; }
    if (k < 1 || k > PCC_KNN_MAX_K) { set_error("k=%d outside [1, %d]", k, PCC_KNN_MAX_K); return PCC_ERR_UNSUPPORTED; }
    PCC_TRY(ensure_grid(ix));
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    const size_t n = ix->n_orig;
    // findPointNeighbours: one batched self k-NN over the packed references
    const unsigned long long* keys = nullptr;
    PCC_TRY(self_knn_keys(ix, k, &keys));
    const float4* dn = reinterpret_cast<const float4*>(normals);
    int32_t* dl = labels;
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(ix->out_d2.reserve(n * sizeof(float4)));
        PCC_HIP(hipMemcpyAsync(ix->out_d2.p, normals, n * sizeof(float4), hipMemcpyHostToDevice, ix->stream));
        dn = ix->out_d2.as<float4>();
        PCC_TRY(ix->q_raw.reserve(n * sizeof(int32_t)));
        dl = ix->q_raw.as<int32_t>();
    }
    PCC_TRY(grid_region_growing(ix, keys, dn, k, smoothness, curvature_threshold, min_size, max_size, dl, n_clusters));
    ev_mark(ix, EV_CALL1);
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, dl, labels, n, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

int pcc_first_within(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem, double radius, int32_t* idx) {
    PCC_ENTER(ix);
    PCC_TRY(check_points(q, nq, stride, mem));
    if (!idx) { set_error("null idx"); return PCC_ERR_INVALID; }
    if (!(radius >= 0)) { set_error("bad radius"); return PCC_ERR_INVALID; }
    if (nq == 0) return PCC_OK;
    PCC_TRY(ensure_grid(ix));
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    PCC_TRY(stage_queries(ix, q, nq, stride, mem));
    int32_t* didx = idx;
    if (mem == PCC_MEM_HOST) { PCC_TRY(ix->out_idx.reserve(nq * sizeof(int32_t))); didx = ix->out_idx.as<int32_t>(); }
    PCC_TRY(grid_first_within(ix, ix->q_packed.as<float4>(), nq, radius, didx));
    ev_mark(ix, EV_CALL1);
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, didx, idx, nq, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

int pcc_euclidean_clusters(pcc_index* ix, double tolerance, uint32_t min_size, uint32_t max_size, int mem,
                           int32_t* labels, int32_t* n_clusters, int32_t* sizes, int max_sizes) {
    PCC_ENTER(ix);
    if (mem != PCC_MEM_HOST && mem != PCC_MEM_DEVICE) { set_error("bad mem space"); return PCC_ERR_INVALID; }
    if (!labels) { set_error("null labels"); return PCC_ERR_INVALID; }
    if (!(tolerance >= 0)) { set_error("bad tolerance"); return PCC_ERR_INVALID; }
    PCC_TRY(ensure_grid(ix));
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    // EuclideanClusterExtraction stores the tolerance as double, extractEuclideanClusters takes
    // it as float, radiusSearch squares it in double: r2 = float(double(float(tol))^2)
    const float tol_f = (float)tolerance;
    const float r2 = radius2((double)tol_f);
    int32_t* dl = labels;
    if (mem == PCC_MEM_HOST) { PCC_TRY(ix->out_idx.reserve(ix->n_orig * sizeof(int32_t))); dl = ix->out_idx.as<int32_t>(); }
    PCC_TRY(grid_clusters(ix, tol_f, r2, min_size, max_size, dl, n_clusters, sizes, max_sizes));
    ev_mark(ix, EV_CALL1);
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, dl, labels, ix->n_orig, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

int pcc_sor(pcc_index* ix, int mean_k, double stddev_mult, int mem, float* mean_dist, uint8_t* inlier,
            double* threshold, size_t* kept) {
    PCC_ENTER(ix);
    if (mem != PCC_MEM_HOST && mem != PCC_MEM_DEVICE) { set_error("bad mem space"); return PCC_ERR_INVALID; }
    if (mean_k < 1 || mean_k + 1 > PCC_KNN_MAX_K) { set_error("mean_k=%d outside [1, %d]", mean_k, PCC_KNN_MAX_K - 1); return PCC_ERR_UNSUPPORTED; }
    PCC_TRY(ensure_grid(ix));
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    const size_t n = ix->n_orig, no = ix->n_orig;
    const int K = mean_k + 1;
    // self query: the packed references ARE the queries (non-finite points are flagged, are
    // skipped by the search and keep distance 0, as in PCL's applyFilterIndices)
    // the mean needs the distances only: the search delivers rows of d2 (4 bytes an entry) where its wave kernels serve K
    // and the staged mean kernel's tile fits LDS (K <= 126), rows of keys otherwise
    const bool d2_only = grid_knn_delivers(K) && (size_t)2 * 64 * (K + 1) * sizeof(unsigned int) <= 64 * 1024;
    PCC_TRY(ix->out_packed.reserve(n * (size_t)K * (d2_only ? sizeof(float) : sizeof(unsigned long long))));
    auto* keys = d2_only ? nullptr : ix->out_packed.as<unsigned long long>();
    float* d2_rows = d2_only ? ix->out_packed.as<float>() : nullptr;
    PCC_TRY(grid_knn(ix, ix->refs.as<float4>(), n, K, keys, nullptr, d2_rows));
    PCC_TRY(ix->out_d2.reserve(no * sizeof(float)));
    float* dmean = ix->out_d2.as<float>();
    PCC_HIP(hipMemsetAsync(dmean, 0, no * sizeof(float), ix->stream));
    PCC_TRY(launch_sor_mean(ix->stream, keys, ix->refs.as<float4>(), n, K, dmean, d2_rows));
    // statistics, threshold and mask on the device (pack.hip: exact whenever no addition of PCL's in-order sums rounds);
    // the host sees 48 bytes.  Round 3 copied the means back, added them up on one host thread and sent a mask: 0.94 ms
    // beside a 1.2 ms search at 1M points
    struct { double sum, sq, thr; unsigned long long kept; unsigned int exact, pad; } hs{};
    PCC_TRY(ix->scratch_a.reserve((size_t)(3 * 1024 + 4) * sizeof(double) + 64));
    PCC_TRY(ix->scratch_b.reserve(no + 64));
    void* st_dev = ix->small.as<char>() + 256;  // (words 64..75 of the small block: free of the search counters)
    uint8_t* dmask = mem == PCC_MEM_DEVICE && inlier ? inlier : ix->scratch_b.as<uint8_t>();
    PCC_TRY(launch_sor_stats(ix->stream, dmean, no, ix->d_grid.as<GridDev>(), K, stddev_mult, ix->scratch_a.as<double>(), st_dev, dmask));
    ev_mark(ix, EV_CALL1);
    PCC_HIP(hipMemcpyAsync(&hs, st_dev, sizeof(hs), hipMemcpyDeviceToHost, ix->stream));
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(ix->host_a.reserve(no * sizeof(float)));
        PCC_TRY(ix->host_b.reserve(no));
        if (mean_dist) PCC_HIP(hipMemcpyAsync(ix->host_a.p, dmean, no * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
        if (inlier) PCC_HIP(hipMemcpyAsync(ix->host_b.p, dmask, no, hipMemcpyDeviceToHost, ix->stream));
    } else if (mean_dist) {
        PCC_HIP(hipMemcpyAsync(mean_dist, dmean, no * sizeof(float), hipMemcpyDeviceToDevice, ix->stream));
    }
    PCC_HIP(hipStreamSynchronize(ix->stream));
    double thr = hs.thr;
    size_t k_in = (size_t)hs.kept;
    if (!hs.exact) {
        // some addition of the in-order sums rounds (terms spread over more than 28 bits below the total): PCL's order
        // decides the last bits, so the sums are taken in that order -- on the host, as round 3 always did
        PCC_TRY(ix->host_a.reserve(no * sizeof(float)));
        PCC_TRY(ix->host_b.reserve(no));
        float* hm = ix->host_a.as<float>();
        uint8_t* hin = ix->host_b.as<uint8_t>();
        PCC_HIP(hipMemcpyAsync(hm, dmean, no * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipStreamSynchronize(ix->stream));
        PCC_TRY(sync_info(ix));
        const size_t valid = ix->n_valid >= (size_t)K ? ix->n_valid : 0;
        double sum = 0, sq = 0;
        for (size_t i = 0; i < no; ++i) { const float f = hm[i]; sum += f; sq += (double)(f * f); }  // PCL squares in float, then widens
        const double mean = sum / (double)valid;
        const double var = (sq - sum * sum / (double)valid) / ((double)valid - 1);
        thr = mean + stddev_mult * std::sqrt(var);
        k_in = 0;
        for (size_t i = 0; i < no; ++i) { hin[i] = !(hm[i] > thr); k_in += hin[i]; }
        if (mem == PCC_MEM_DEVICE && inlier) {
            PCC_HIP(hipMemcpyAsync(inlier, hin, no, hipMemcpyHostToDevice, ix->stream));
            PCC_HIP(hipStreamSynchronize(ix->stream));
        }
    }
    if (threshold) *threshold = thr;
    if (kept) *kept = k_in;
    if (mem == PCC_MEM_HOST) {
        if (mean_dist) memcpy(mean_dist, ix->host_a.p, no * sizeof(float));
        if (inlier) memcpy(inlier, ix->host_b.p, no);
    }
    ix->sor_exact_last = hs.exact != 0;
    return PCC_OK;
}

int pcc_index_sor_on_device(const pcc_index* ix, int* on_device) {
    if (!ix || !on_device) { set_error("null argument"); return PCC_ERR_INVALID; }
    *on_device = ix->sor_exact_last ? 1 : 0;
    return PCC_OK;
}

// ---- ICP ------------------------------------------------------------------------------------
// reduce the per-workgroup partial rows in a fixed order
static int icp_reduce(pcc_index* ix, size_t n, double sums[17], const double* center = nullptr) {
    PCC_TRY(ix->scratch_a.reserve((size_t)ICP_MAX_BLOCKS * 17 * sizeof(double)));
    int nb = 0;
    PCC_TRY(launch_icp_sums(ix->stream, ix->q_packed.as<float4>(), n, ix->out_packed.as<unsigned long long>(),
                            ix->refs.as<float4>(), ix->scratch_a.as<double>(), &nb,
                            ix->engine == PCC_ENGINE_GRID ? ix->small.as<unsigned int>() + 32 : nullptr,
                            static_cast<unsigned int*>(ix->pinned) + 40, center));
    std::vector<double> h((size_t)nb * 17);
    PCC_HIP(hipMemcpyAsync(h.data(), ix->scratch_a.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, ix->stream));
    PCC_HIP(hipStreamSynchronize(ix->stream));
    for (int k = 0; k < 17; ++k) sums[k] = 0;
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < 17; ++k) sums[k] += h[(size_t)b * 17 + k];
    return PCC_OK;
}

int pcc_rigid_from_sums(const double sums[17], float T[16]) { return pcc_rigid_from_sums_about(sums, nullptr, T); }

int pcc_rigid_from_sums_about(const double sums[17], const double center[3], float T[16]) {
    if (!sums || !T) { set_error("null argument"); return PCC_ERR_INVALID; }
    if (rigid_from_sums(sums, T, center) != 0) { set_error("fewer than 3 correspondences"); return PCC_ERR_INVALID; }
    return PCC_OK;
}

int pcc_icp_step(pcc_index* ix, const void* src, size_t n, size_t stride, int mem, int32_t* idx, float* d2,
                 double sums[17]) {
    return pcc_icp_step_about(ix, src, n, stride, mem, nullptr, idx, d2, sums);
}

int pcc_icp_step_about(pcc_index* ix, const void* src, size_t n, size_t stride, int mem, const double center[3],
                       int32_t* idx, float* d2, double sums[17]) {
    PCC_ENTER(ix);
    PCC_TRY(check_points(src, n, stride, mem));
    if (center && !(std::isfinite(center[0]) && std::isfinite(center[1]) && std::isfinite(center[2]))) {
        set_error("non-finite center");
        return PCC_ERR_INVALID;
    }
    if (!sums) { set_error("null sums"); return PCC_ERR_INVALID; }
    for (int k = 0; k < 17; ++k) sums[k] = 0;
    if (n == 0) return PCC_OK;
    if (ix->n_orig == 0) { set_error("index is empty"); return PCC_ERR_EMPTY; }
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    PCC_TRY(stage_queries(ix, src, n, stride, mem));
    PCC_TRY(nn1_packed(ix, n));
    const double* center_dev = nullptr;
    if (center) {  // the sums are taken about it (device copy behind the ICP loop state)
        PCC_TRY(ix->icp_state.reserve(sizeof(IcpState) + (3 + 17) * sizeof(double)));
        double* cd = reinterpret_cast<double*>(ix->icp_state.as<char>() + sizeof(IcpState));
        PCC_HIP(hipMemcpyAsync(cd, center, 3 * sizeof(double), hipMemcpyHostToDevice, ix->stream));
        PCC_HIP(hipStreamSynchronize(ix->stream));  // (center is the caller's memory)
        center_dev = cd;
    }
    PCC_TRY(icp_reduce(ix, n, sums, center_dev));
    ev_mark(ix, EV_CALL1);
    if (idx || d2) {
        int32_t* didx = idx;
        float* dd2 = d2;
        if (mem == PCC_MEM_HOST) {
            PCC_TRY(ix->out_idx.reserve(n * sizeof(int32_t)));
            PCC_TRY(ix->out_d2.reserve(n * sizeof(float)));
            didx = idx ? ix->out_idx.as<int32_t>() : nullptr;
            dd2 = d2 ? ix->out_d2.as<float>() : nullptr;
        }
        PCC_TRY(launch_unpack(ix->stream, ix->out_packed.as<unsigned long long>(), nullptr, n, didx, dd2));
        if (mem == PCC_MEM_HOST) {
            PCC_TRY(deliver(ix, didx, idx, n, mem));
            PCC_TRY(deliver(ix, dd2, d2, n, mem));
            PCC_HIP(hipStreamSynchronize(ix->stream));
        }
    }
    return PCC_OK;
}

int pcc_transform(pcc_index* ix, const float T[16], const void* src, size_t n, size_t sstride, void* dst,
                  size_t dstride, int mem) {
    PCC_ENTER(ix);
    PCC_TRY(check_points(src, n, sstride, mem));
    PCC_TRY(check_points(dst, n, dstride, mem));
    if (!T) { set_error("null T"); return PCC_ERR_INVALID; }
    if (n == 0) return PCC_OK;
    if (mem == PCC_MEM_DEVICE) return launch_transform(ix->stream, nullptr, T, src, n, sstride, dst, dstride);
    // host: stage src (and dst, so that its other fields survive) on the device
    PCC_TRY(ix->q_raw.reserve(n * sstride));
    PCC_HIP(hipMemcpyAsync(ix->q_raw.p, src, (n - 1) * sstride + 12, hipMemcpyHostToDevice, ix->stream));
    void* ddst = ix->q_raw.p;
    size_t dbytes = (n - 1) * dstride + 12;
    if (dst != src || dstride != sstride) {
        PCC_TRY(ix->scratch_d.reserve(n * dstride));
        PCC_HIP(hipMemcpyAsync(ix->scratch_d.p, dst, dbytes, hipMemcpyHostToDevice, ix->stream));
        ddst = ix->scratch_d.p;
    }
    PCC_TRY(launch_transform(ix->stream, nullptr, T, ix->q_raw.p, n, sstride, ddst, dstride));
    PCC_HIP(hipMemcpyAsync(dst, ddst, dbytes, hipMemcpyDeviceToHost, ix->stream));
    PCC_HIP(hipStreamSynchronize(ix->stream));
    return PCC_OK;
}

int pcc_icp_align(pcc_index* ix, const void* src, size_t n, size_t stride, int mem, int max_iter, int fixed,
                  float T[16], double* fitness, int* iterations, int* converged) {
    return pcc::icp_align_impl(ix, nullptr, src, n, stride, mem, max_iter, fixed, T, fitness, iterations, converged);
}
}  // extern "C"

// pcc_icp_align, and -- with `hooks` -- its sharded form: this handle holds one SHARD of the source cloud, the 17 sums of
// every pass are added up over the ranks (hooks->allreduce_sum_f64: RCCL on the handle's stream, comm.hip) before the
// solver sees them, so every rank solves the same transform and moves its shard (reference
// src/comparator.cpp:1089-1110).  With one rank the all-reduce is the identity and the result is pcc_icp_align's, bit for bit.
int pcc::icp_align_impl(pcc_index* ix, const pcc::IcpHooks* hooks, const void* src, size_t n, size_t stride, int mem, int max_iter,
                        int fixed, float T[16], double* fitness, int* iterations, int* converged) {
    // (PCC_ENTER spelled out: a device that cannot be selected is a failure of this rank alone and has to reach the status
    // exchange below like every other one -- an early return here would leave the peers waiting in it)
    if (!ix) { set_error("null index"); return PCC_ERR_INVALID; }  // (the sharded entry point has checked this before its peers can wait)
    std::lock_guard<std::mutex> _lock(ix->mu);
    pcc::DeviceGuard _guard(ix->device);
    pcc::entered(ix);
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int it = 0;
    bool conv = false;
    double prev_mse = 1.79769313486231570e308;
    double* center_dev = nullptr;
    bool sorted = false;
    // Everything that can fail on ONE rank alone -- argument checks, staging, allocations -- comes before the first
    // collective and ends in a status the ranks agree on (hooks->agree: all-reduce MIN of one word), so a rank that
    // cannot go on takes the others out with it instead of leaving them in the broadcast below (comm.hip).
    auto prepare = [&]() -> int {
        if (!_guard.ok) { set_error("hipSetDevice(%d) failed", ix->device); return PCC_ERR_DEVICE; }
        PCC_TRY(check_points(src, n, stride, mem));
        if (!T) { set_error("null T"); return PCC_ERR_INVALID; }
        if (ix->n_orig == 0) { set_error("index is empty"); return PCC_ERR_EMPTY; }
        memcpy(T, I, sizeof(I));
        if (iterations) *iterations = 0;
        if (converged) *converged = 0;
        if (n == 0 && hooks) { set_error("sharded ICP: every rank needs a non-empty shard"); return PCC_ERR_INVALID; }
        if (n == 0) return PCC_OK;
        // the source stays resident: q_packed is the moving cloud, icp_src keeps the input
        PCC_TRY(stage_queries(ix, src, n, stride, mem));
        PCC_TRY(ix->icp_src.reserve(n * sizeof(float4)));
        // Round 5: the loop's working set in the target grid's CELL order.  Nothing of the loop leaves per point -- T, fitness,
        // counts -- so the permutation that a search pays per call (queries gathered through the sort order, keys scattered
        // back: ~60 us of a 215-us pass at 2M points, tools/ubench/ubench_scatter.hip) is paid ONCE: the source is sorted by the
        // cell it starts in, gathered into that order, and every pass reads it front to back with the identity as its order
        // (a rigid motion keeps neighbouring points neighbours; any order is correct, as before).  The sums are added up in
        // this order by every form of the loop -- device-resident, host-driven, sharded -- so they agree with each other to
        // the bit as before; against the caller's order they differ in the last bits of a double sum.
        sorted = ix->opt.icp_sorted != 0 && ix->engine == PCC_ENGINE_GRID && ix->has_grid && n >= 4096;
        if (sorted) {
            unsigned int *order = nullptr, *n_sorted = nullptr;
            PCC_TRY(grid_sort_queries(ix, ix->q_packed.as<float4>(), n, &order, &n_sorted));
            PCC_TRY(launch_gather_sorted(ix->stream, ix->q_packed.as<float4>(), order, n_sorted, n, ix->icp_src.as<float4>(),
                                         ix->small.as<unsigned int>() + 48));
            PCC_HIP(hipMemcpyAsync(ix->q_packed.p, ix->icp_src.p, n * sizeof(float4), hipMemcpyDeviceToDevice, ix->stream));
        } else
        PCC_HIP(hipMemcpyAsync(ix->icp_src.p, ix->q_packed.p, n * sizeof(float4), hipMemcpyDeviceToDevice, ix->stream));
        // the sums of every pass are taken about a point of the source cloud (k_icp_center: no cancellation in the
        // covariance for clouds far from the origin); it sits behind the loop state in device memory
        PCC_TRY(ix->icp_state.reserve(sizeof(IcpState) + (3 + 17) * sizeof(double)));
        PCC_TRY(ix->scratch_a.reserve((size_t)ICP_MAX_BLOCKS * 17 * sizeof(double)));
        center_dev = reinterpret_cast<double*>(ix->icp_state.as<char>() + sizeof(IcpState));
        PCC_TRY(launch_icp_center(ix->stream, ix->q_packed.as<float4>(), n, center_dev));
        return PCC_OK;
    };
    int st_prep = prepare();
    if (hooks) st_prep = hooks->agree(hooks->ctx, st_prep);
    if (st_prep != PCC_OK) return st_prep;
    if (n == 0) return PCC_OK;
    if (hooks) PCC_TRY(hooks->bcast_f64(hooks->ctx, center_dev, 3, 0, ix->stream));  // every rank about rank 0's point
    const int warm_env = ix->opt.icp_warm;         // 0: every pass from scratch (measurements)
    const int loop_env = hooks ? 1 : ix->opt.icp_device_loop;  // 0: the host-driven loop (kept for comparison: same bits)
    double* sums_dev = center_dev + 3;  // (sharded: the 17 sums of a pass, all-reduced in place)
    struct KeepOrder {  // the passes below share the first pass's lane order (see pcc_index::keep_order)
        pcc_index* ix;
        explicit KeepOrder(pcc_index* i) : ix(i) { ix->keep_order = true; ix->order_valid = false; ix->warm_start = false; }
        ~KeepOrder() { ix->keep_order = false; ix->order_valid = false; ix->warm_start = false; ix->pre_transform = nullptr; }
    } keep_order_guard(ix);
    const bool fold = sorted && loop_env && grid_nn1_takes_transform(ix);  // the pass's transform applied by the next pass's search
    if (sorted) {  // (the passes take the identity as their order; the count of valid points sits in a word of its own)
        ix->order_valid = true;
        ix->order_nq = n;
        ix->order_ptr = nullptr;
        ix->order_nsorted = ix->small.as<unsigned int>() + 48;
    }
    if (loop_env) {
        // The loop lives on the device: every pass is NN -> sums -> k_icp_solve (one workgroup: the transform, the running
        // product and the convergence criteria) -> transform with the matrix the solver left in device memory.  Passes
        // are enqueued in chunks without a host round trip (the host loop below pays a stream synchronisation, a
        // read-back and a launch gap per pass, ~65 us of 0.44 ms); after each chunk the host looks whether the loop
        // has stopped.  Passes enqueued past the stop are no-ops on the state (identity transform), so a chunk costs at
        // most its own length in wasted searches -- none with a fixed count, where the whole loop is one chunk.
        IcpState h0{};
        memcpy(h0.Ti, I, sizeof(I));
        memcpy(h0.T, I, sizeof(I));
        h0.prev_mse = 1.79769313486231570e308;
        PCC_HIP(hipMemcpyAsync(ix->icp_state.p, &h0, sizeof(h0), hipMemcpyHostToDevice, ix->stream));
        PCC_HIP(hipStreamSynchronize(ix->stream));  // (h0 lives on this stack frame)
        IcpState* st = ix->icp_state.as<IcpState>();
        IcpState h1 = h0;
        // passes per host look: 5 with criteria active; with a fixed count the loop would need none, but a source that
        // leaves fewer than 3 correspondences stops it on the device and every pass enqueued beyond that is a wasted
        // search -- so at most 32 at a time (one look costs ~20 us)
        const int chunk = fixed ? (max_iter < 32 ? max_iter : 32) : 5;
        for (int pass = 0; pass < max_iter && !h1.stopped;) {
            for (int c = 0; c < chunk && pass < max_iter; ++c, ++pass) {
                ev_next(ix);  // instrumentation: every pass is one "call" (NN kernel, far/fallback, whole pass)
                ev_mark(ix, EV_CALL0);
                // (cell-ordered loop: the search applies the previous pass's matrix -- the identity before the first -- to the
                // queries it reads and writes them back; no transform kernel, grid.hip k_grid_nn1_flat2)
                ix->pre_transform = fold ? st->Ti : nullptr;
                PCC_TRY(nn1_packed(ix, n));  // determineCorrespondences: one NN per source point
                ix->pre_transform = nullptr;
                ix->warm_start = warm_env != 0;  // from now on out_packed holds the last pass's keys of these same points
                int nb = 0;
                unsigned int* zw = ix->engine == PCC_ENGINE_GRID ? ix->small.as<unsigned int>() + 32 : nullptr;
                // (one GPU: the sums kernel's last workgroup solves the pass itself -- [53] of `small` is its ticket word;
                // PCC_OPT_FUSE_PARAMS bit 1; the sharded loop keeps the solver's own launch: its sums pass through an all-reduce)
                const bool fuse_solve = !hooks && (ix->opt.fuse_params & 2) != 0;
                const IcpFuse fuse{ix->small.as<unsigned int>() + 53, st, max_iter, fixed, fold ? zw : nullptr};
                PCC_TRY(launch_icp_sums(ix->stream, ix->q_packed.as<float4>(), n, ix->out_packed.as<unsigned long long>(),
                                        ix->refs.as<float4>(), ix->scratch_a.as<double>(), &nb, zw,
                                        static_cast<unsigned int*>(ix->pinned) + 40, center_dev, fuse_solve ? &fuse : nullptr));
                if (fuse_solve) {
                } else if (hooks) {  // rows -> 17 sums (workgroup order, as the solver adds them) -> sum over the ranks -> solve
                    PCC_TRY(launch_icp_rows_to_sums(ix->stream, ix->scratch_a.as<double>(), nb, sums_dev));
                    PCC_TRY(hooks->allreduce_sum_f64(hooks->ctx, sums_dev, 17, ix->stream));
                    PCC_TRY(launch_icp_solve(ix->stream, sums_dev, 1, st, max_iter, fixed, center_dev, fold ? zw : nullptr));
                } else
                PCC_TRY(launch_icp_solve(ix->stream, ix->scratch_a.as<double>(), nb, st, max_iter, fixed, center_dev, fold ? zw : nullptr));
                // (the transform -- or, when the next search applies it itself, the solver -- also zeroes the counters of the next
                // pass's search)
                if (!fold) PCC_TRY(launch_transform(ix->stream, st->Ti, nullptr, ix->q_packed.p, n, sizeof(float4), ix->q_packed.p, sizeof(float4), zw));
                if (zw && n > 0) ix->fb_zeroed = true;
                ev_mark(ix, EV_CALL1);
            }
            PCC_HIP(hipMemcpyAsync(&h1, ix->icp_state.p, sizeof(h1), hipMemcpyDeviceToHost, ix->stream));
            PCC_HIP(hipStreamSynchronize(ix->stream));
        }
        memcpy(T, h1.T, sizeof(h1.T));
        it = h1.it;
        conv = h1.converged != 0;
    } else {
    double center[3] = {0, 0, 0};
    bool have_center = false;
    while (it < max_iter) {
        ev_next(ix);  // instrumentation: every pass is one "call" (NN kernel, far/fallback, whole pass)
        ev_mark(ix, EV_CALL0);
        PCC_TRY(nn1_packed(ix, n));  // determineCorrespondences: one NN per source point
        ix->warm_start = warm_env != 0;  // from now on out_packed holds the last pass's keys of these same points
        double sums[17];
        PCC_TRY(icp_reduce(ix, n, sums, center_dev));
        if (!have_center) {
            PCC_HIP(hipMemcpyAsync(center, center_dev, sizeof(center), hipMemcpyDeviceToHost, ix->stream));
            PCC_HIP(hipStreamSynchronize(ix->stream));
            have_center = true;
        }
        float Ti[16];
        if (rigid_from_sums(sums, Ti, center) != 0) { conv = false; break; }  // < 3 correspondences: not converged
        PCC_TRY(launch_transform(ix->stream, nullptr, Ti, ix->q_packed.p, n, sizeof(float4), ix->q_packed.p, sizeof(float4)));
        ev_mark(ix, EV_CALL1);
        mat4_mul_f(Ti, T, T);  // final = T_i * final
        const double mse = sums[15] / sums[16];
        ++it;
        if (it >= max_iter) { conv = true; break; }  // DefaultConvergenceCriteria: iteration cap counts as converged
        if (!fixed && std::fabs(mse - prev_mse) < 1e-12) { conv = true; break; }
        prev_mse = mse;
    }
    }
    if (iterations) *iterations = it;
    if (converged) *converged = conv ? 1 : 0;
    if (fitness) {
        // getFitnessScore: re-transform the INPUT with the final matrix, one more NN pass, mean d2
        PCC_TRY(launch_transform(ix->stream, nullptr, T, ix->icp_src.p, n, sizeof(float4), ix->q_packed.p, sizeof(float4)));
        // launch_transform writes x,y,z only: refresh the validity flags from the input
        PCC_TRY(launch_copy_w(ix->stream, ix->icp_src.as<float4>(), ix->q_packed.as<float4>(), n));
        ev_next(ix);
        ev_mark(ix, EV_CALL0);
        PCC_TRY(nn1_packed(ix, n));
        double sums[17];
        if (hooks) {  // sum of d2 and count over ALL shards
            int nb = 0;
            PCC_TRY(launch_icp_sums(ix->stream, ix->q_packed.as<float4>(), n, ix->out_packed.as<unsigned long long>(), ix->refs.as<float4>(),
                                    ix->scratch_a.as<double>(), &nb, nullptr, nullptr, nullptr));
            PCC_TRY(launch_icp_rows_to_sums(ix->stream, ix->scratch_a.as<double>(), nb, sums_dev));
            PCC_TRY(hooks->allreduce_sum_f64(hooks->ctx, sums_dev, 17, ix->stream));
            PCC_HIP(hipMemcpyAsync(sums, sums_dev, sizeof(sums), hipMemcpyDeviceToHost, ix->stream));
            PCC_HIP(hipStreamSynchronize(ix->stream));
        } else
        PCC_TRY(icp_reduce(ix, n, sums));
        ev_mark(ix, EV_CALL1);
        *fitness = sums[16] > 0 ? sums[15] / sums[16] : 1.79769313486231570e308;
    }
    PCC_HIP(hipStreamSynchronize(ix->stream));
    return PCC_OK;
}

extern "C" {
int pcc_match_knn(pcc_index* ix, const void* des2, size_t n2, size_t stride, int mem, float threshold,
                  int32_t* out, int32_t* out_size) {
    PCC_ENTER(ix);
    PCC_TRY(check_points(des2, n2, stride, mem));
    if (!out || !out_size) { set_error("null output"); return PCC_ERR_INVALID; }
    out[0] = 0;  // std::vector<int> correspondence(1) -- reference src/comparator.cpp:568
    *out_size = 1;
    if (n2 == 0) return PCC_OK;
    if (ix->n_orig == 0) { set_error("index is empty"); return PCC_ERR_EMPTY; }
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    if (small_call(ix, n2, stride, mem)) {
        PCC_TRY(small_nn1(ix, des2, n2, stride, true, true));
    } else {
        PCC_TRY(stage_queries(ix, des2, n2, stride, mem));
        PCC_TRY(nn1_packed(ix, n2));
        if (ix->tie_mode == PCC_TIES_FLANN) PCC_TRY(resolve_ties_flann(ix, ix->q_packed.as<float4>(), ix->out_packed.as<unsigned long long>(), n2, true));
        // (the unpack kernel writes the result arrays into pinned host memory itself: no copy command, one wait)
        PCC_TRY(ix->host_a.reserve(n2 * sizeof(int32_t)));
        PCC_TRY(ix->host_b.reserve(n2 * sizeof(float)));
        PCC_TRY(launch_unpack(ix->stream, ix->out_packed.as<unsigned long long>(), nullptr, n2, ix->host_a.as<int32_t>(), ix->host_b.as<float>()));
        ev_mark(ix, EV_CALL1);
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    const int32_t* hi = ix->host_a.as<int32_t>();
    const float* hd = ix->host_b.as<float>();
    int32_t c = 1;
    for (size_t i = 0; i < n2; ++i)  // neighborCount == 1 && squaredDistances[0] < threshold (:579)
        if (hi[i] >= 0 && hd[i] < threshold) out[c++] = hi[i];
    *out_size = c;
    return PCC_OK;
}

}  // extern "C"

namespace pcc {
int make_handle(int device, int engine, pcc_index** out) { return new_handle(device, engine, out); }
int need_grid(pcc_index* ix) { return ensure_grid(ix); }
}  // namespace pcc
