// api.hip -- the C-ABI of libpcc_nn (include/pcc_nn.h) on top of the gfx950 kernels.
// Host-side glue only: argument checks, H2D/D2H staging, launch order.  There is
// no CPU compute fallback: without a HIP device every entry point fails with
// PCC_ERR_DEVICE.
#include "pcc_internal.hpp"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

namespace pcc {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap && p) return PCC_OK;
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
    pcc::DeviceGuard _guard((ix)->device);                                    \
    if (!_guard.ok) { pcc::set_error("hipSetDevice(%d) failed", (ix)->device); return PCC_ERR_DEVICE; }

// Stage a caller cloud (host or device AoS) as packed float4 on the device.
// host: one H2D copy of the raw AoS, then the pack kernel.
static int stage_points(pcc_index* ix, const void* pts, size_t n, size_t stride, int mem,
                        DevBuf& raw, float4* packed, float* blk_stats = nullptr, int* n_blocks = nullptr) {
    const void* src = pts;
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(raw.reserve(n * stride));
        PCC_HIP(hipMemcpyAsync(raw.p, pts, (n - 1) * stride + 12, hipMemcpyHostToDevice, ix->stream));
        src = raw.p;
    }
    return launch_pack(ix->stream, src, n, stride, packed, blk_stats, n_blocks);
}

static int check_points(const void* pts, size_t n, size_t stride, int mem) {
    if (mem != PCC_MEM_HOST && mem != PCC_MEM_DEVICE) { set_error("bad mem space %d", mem); return PCC_ERR_INVALID; }
    if (n && !pts) { set_error("null point pointer"); return PCC_ERR_INVALID; }
    if (stride < 12 || stride % 4) { set_error("stride %zu must be a multiple of 4 and >= 12", stride); return PCC_ERR_INVALID; }
    if (n >= (1ull << 31)) { set_error("more than 2^31 points"); return PCC_ERR_UNSUPPORTED; }
    return PCC_OK;
}

// queries -> ix->q_packed (float4, w < 0 marks a non-finite query)
static int stage_queries(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem) {
    PCC_TRY(ix->q_packed.reserve(nq * sizeof(float4)));
    return stage_points(ix, q, nq, stride, mem, ix->q_raw, ix->q_packed.as<float4>());
}

// deliver device results to the caller's memory space
template <class T>
static int deliver(pcc_index* ix, const T* dev, T* user, size_t count, int mem) {
    if (!user || count == 0) return PCC_OK;
    if (mem == PCC_MEM_HOST) {
        PCC_HIP(hipMemcpyAsync(user, dev, count * sizeof(T), hipMemcpyDeviceToHost, ix->stream));
    } else if (user != dev) {
        PCC_HIP(hipMemcpyAsync(user, dev, count * sizeof(T), hipMemcpyDeviceToDevice, ix->stream));
    }
    return PCC_OK;
}

int grid_build(pcc_index* ix, const float lo[3], const float hi[3]);  // grid.hip
int grid_nn1(pcc_index* ix, const float4* q, size_t nq, unsigned long long* out);

// k = 1 search of ix->q_packed[0..nq) into ix->out_packed (u64 per query)
static int nn1_packed(pcc_index* ix, size_t nq) {
    PCC_TRY(ix->out_packed.reserve(nq * sizeof(unsigned long long)));
    auto* out = ix->out_packed.as<unsigned long long>();
    PCC_HIP(hipMemsetAsync(out, 0xff, nq * sizeof(unsigned long long), ix->stream));
    if (ix->engine == PCC_ENGINE_GRID) return grid_nn1(ix, ix->q_packed.as<float4>(), nq, out);
    ix->stats[0] = 0;
    ix->stats[1] = nq;
    ev_mark(ix, EV_MAIN0);
    int st = launch_nn1_brute(ix->stream, ix->refs.as<float4>(), ix->n_valid, ix->q_packed.as<float4>(), nq,
                              out, nullptr, nullptr, 0);
    ev_mark(ix, EV_MAIN1);
    return st;
}

static int resolve_engine(int requested, size_t n_valid) {
    // the grid build costs a few passes over the cloud; below ~4k points one exhaustive
    // sweep is cheaper than building it
    if (requested == PCC_ENGINE_AUTO) return n_valid >= 4096 ? PCC_ENGINE_GRID : PCC_ENGINE_BRUTE;
    return requested;
}

// (re)build the index over a new cloud: pack (+ bbox + invalid count in the same pass),
// optional order-preserving compaction, optional grid build.  One stream sync.
static int set_input(pcc_index* ix, const void* pts, size_t n, size_t stride, int mem) {
    ix->n_valid = 0;
    ix->has_grid = false;
    ix->n_orig = n;
    ev_next(ix);
    ev_mark(ix, EV_BUILD0);
    PCC_TRY(ix->refs.reserve(n * sizeof(float4)));
    unsigned int* d_cnt = ix->small.as<unsigned int>();
    float* d_blk = ix->blk_stats.as<float>();
    float* h_blk = static_cast<float*>(ix->pinned);
    int nblk = 0;
    PCC_TRY(stage_points(ix, pts, n, stride, mem, ix->q_raw, ix->refs.as<float4>(), d_blk, &nblk));
    PCC_HIP(hipMemcpyAsync(h_blk, d_blk, (size_t)nblk * 8 * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    PCC_HIP(hipStreamSynchronize(ix->stream));
    size_t n_invalid = 0;
    for (int a = 0; a < 3; ++a) { ix->bbox_lo[a] = INFINITY; ix->bbox_hi[a] = -INFINITY; }
    for (int b = 0; b < nblk; ++b) {
        unsigned int bad;
        memcpy(&bad, &h_blk[b * 8], 4);
        n_invalid += bad;
        for (int a = 0; a < 3; ++a) {
            ix->bbox_lo[a] = std::min(ix->bbox_lo[a], h_blk[b * 8 + 1 + a]);
            ix->bbox_hi[a] = std::max(ix->bbox_hi[a], h_blk[b * 8 + 4 + a]);
        }
    }
    size_t n_valid = n - n_invalid;
    if (n_valid == 0) { set_error("Cannot create a KDTree with an empty input cloud (all %zu points non-finite)", n); return PCC_ERR_EMPTY; }
    if (n_invalid) {
        // order-preserving compaction == PCL's index_mapping_
        DevBuf packed2;
        PCC_TRY(packed2.reserve(n_valid * sizeof(float4)));
        int st = launch_compact(ix->stream, ix->refs.as<float4>(), n, packed2.as<float4>(), d_cnt + 12, ix->scratch_a);
        if (st == PCC_OK && hipStreamSynchronize(ix->stream) != hipSuccess) { set_error("compaction failed"); st = PCC_ERR_DEVICE; }
        if (st != PCC_OK) { packed2.release(); return st; }
        ix->refs.release();
        ix->refs = packed2;
    }
    ix->n_valid = n_valid;
    ix->stats[2] = n_valid;
    ix->engine = resolve_engine(ix->engine_requested, n_valid);
    if (ix->engine == PCC_ENGINE_GRID) {
        PCC_TRY(grid_build(ix, ix->bbox_lo, ix->bbox_hi));
    }
    ev_mark(ix, EV_BUILD1);
    return PCC_OK;
}

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
                      &ix->out_idx, &ix->out_d2, &ix->scratch_a, &ix->scratch_b, &ix->scratch_c, &ix->scratch_d, &ix->scratch_e,
                      &ix->small, &ix->blk_stats};
    for (DevBuf* b : bufs) b->release();
    for (int sl = 0; sl < PCC_EV_SLOTS; ++sl)
        for (int k = 0; k < PCC_EV_KINDS; ++k)
            if (ix->ev[sl][k]) (void)hipEventDestroy(ix->ev[sl][k]);
    if (ix->pinned) (void)hipHostFree(ix->pinned);
    if (ix->own_stream) (void)hipStreamDestroy(ix->own_stream);
    delete ix;
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
    pcc_index* ix = new pcc_index();
    ix->device = device;
    DeviceGuard g(device);
    int st = PCC_OK;
    auto fail = [&](int s) { pcc_index_destroy(ix); return s; };
    if (!g.ok) { set_error("hipSetDevice(%d) failed", device); return fail(PCC_ERR_DEVICE); }
    if (hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(PCC_ERR_DEVICE); }
    ix->stream = ix->own_stream;
    if (hipHostMalloc(&ix->pinned, PACK_MAX_BLOCKS * 8 * sizeof(float) + 4096, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc failed"); return fail(PCC_ERR_DEVICE); }
    if ((st = ix->small.reserve(4096)) != PCC_OK) return fail(st);
    if ((st = ix->blk_stats.reserve(PACK_MAX_BLOCKS * 8 * sizeof(float))) != PCC_OK) return fail(st);
    ix->engine_requested = engine;
    ix->engine = engine;
    if ((st = set_input(ix, pts, n, stride, mem)) != PCC_OK) return fail(st);
    if (hipStreamSynchronize(ix->stream) != hipSuccess) { set_error("index build failed: %s", hipGetErrorString(hipGetLastError())); return fail(PCC_ERR_DEVICE); }
    *out = ix;
    return PCC_OK;
}

int pcc_index_set_input(pcc_index* ix, const void* pts, size_t n, size_t stride, int dim, int mem) {
    PCC_ENTER(ix);
    if (dim != 3) { set_error("dim %d unsupported", dim); return PCC_ERR_UNSUPPORTED; }
    PCC_TRY(check_points(pts, n, stride, mem));
    if (n == 0) { ix->n_valid = 0; set_error("Cannot create a KDTree with an empty input cloud"); return PCC_ERR_EMPTY; }
    return set_input(ix, pts, n, stride, mem);
}

int pcc_index_enable_timing(pcc_index* ix, int on) {
    PCC_ENTER(ix);
    PCC_HIP(hipStreamSynchronize(ix->stream));
    if (on) {
        for (int sl = 0; sl < PCC_EV_SLOTS; ++sl)
            for (int k = 0; k < PCC_EV_KINDS; ++k)
                if (!ix->ev[sl][k]) PCC_HIP(hipEventCreate(&ix->ev[sl][k]));
    }
    for (int sl = 0; sl < PCC_EV_SLOTS; ++sl)
        for (int k = 0; k < PCC_EV_KINDS; ++k) ix->ev_rec[sl][k] = false;
    ix->ev_slot = 0;
    ix->timing = on != 0;
    return PCC_OK;
}

int pcc_index_timing(pcc_index* ix, float ms[8]) {
    PCC_ENTER(ix);
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

int pcc_index_size(const pcc_index* ix, size_t* n_valid) {
    if (!ix || !n_valid) { set_error("null argument"); return PCC_ERR_INVALID; }
    *n_valid = ix->n_valid;
    return PCC_OK;
}
int pcc_index_set_stream(pcc_index* ix, void* s) {
    PCC_ENTER(ix);
    PCC_HIP(hipStreamSynchronize(ix->stream));
    ix->stream = s ? static_cast<hipStream_t>(s) : ix->own_stream;
    return PCC_OK;
}
int pcc_index_sync(pcc_index* ix) {
    PCC_ENTER(ix);
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
    engine = resolve_engine(engine, ix->n_valid);
    if (engine == PCC_ENGINE_GRID && !ix->has_grid && ix->n_valid) {
        PCC_TRY(grid_build(ix, ix->bbox_lo, ix->bbox_hi));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    ix->engine = engine;
    return PCC_OK;
}
int pcc_index_stats(const pcc_index* ix, uint64_t stats[8]) {
    if (!ix || !stats) { set_error("null argument"); return PCC_ERR_INVALID; }
    memcpy(stats, ix->stats, sizeof(ix->stats));
    return PCC_OK;
}

int pcc_nn1(pcc_index* ix, const void* q, size_t nq, size_t stride, int mem, int32_t* idx, float* d2) {
    PCC_ENTER(ix);
    PCC_TRY(check_points(q, nq, stride, mem));
    if (nq == 0) return PCC_OK;
    if (ix->n_valid == 0) { set_error("index is empty"); return PCC_ERR_EMPTY; }
    ev_next(ix);
    ev_mark(ix, EV_CALL0);
    PCC_TRY(stage_queries(ix, q, nq, stride, mem));
    PCC_TRY(nn1_packed(ix, nq));
    int32_t* didx = idx;
    float* dd2 = d2;
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(ix->out_idx.reserve(nq * sizeof(int32_t)));
        PCC_TRY(ix->out_d2.reserve(nq * sizeof(float)));
        didx = idx ? ix->out_idx.as<int32_t>() : nullptr;
        dd2 = d2 ? ix->out_d2.as<float>() : nullptr;
    }
    PCC_TRY(launch_unpack(ix->stream, ix->out_packed.as<unsigned long long>(), ix->q_packed.as<float4>(), nq, didx, dd2));
    ev_mark(ix, EV_CALL1);
    if (mem == PCC_MEM_HOST) {
        PCC_TRY(deliver(ix, didx, idx, nq, mem));
        PCC_TRY(deliver(ix, dd2, d2, nq, mem));
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    return PCC_OK;
}

// ---- entry points still to be wired to kernels (round-1 work in progress) ----------------
#define PCC_NOT_YET(name) pcc::set_error(name " is not implemented yet"); return PCC_ERR_UNSUPPORTED
int pcc_knn(pcc_index*, const void*, size_t, size_t, int, int, int32_t*, float*) { PCC_NOT_YET("pcc_knn"); }
int pcc_radius_count(pcc_index*, const void*, size_t, size_t, int, double, int32_t*) { PCC_NOT_YET("pcc_radius_count"); }
int pcc_radius_fill(pcc_index*, const void*, size_t, size_t, int, double, int, const int64_t*, int32_t*, float*) { PCC_NOT_YET("pcc_radius_fill"); }
int pcc_euclidean_clusters(pcc_index*, double, uint32_t, uint32_t, int, int32_t*, int32_t*, int32_t*, int) { PCC_NOT_YET("pcc_euclidean_clusters"); }
int pcc_sor(pcc_index*, int, double, int, float*, uint8_t*, double*, size_t*) { PCC_NOT_YET("pcc_sor"); }
int pcc_icp_step(pcc_index*, const void*, size_t, size_t, int, int32_t*, float*, double*) { PCC_NOT_YET("pcc_icp_step"); }
int pcc_transform(pcc_index*, const float*, const void*, size_t, size_t, void*, size_t, int) { PCC_NOT_YET("pcc_transform"); }
int pcc_icp_align(pcc_index*, const void*, size_t, size_t, int, int, int, float*, double*, int*, int*) { PCC_NOT_YET("pcc_icp_align"); }
int pcc_match_knn(pcc_index*, const void*, size_t, size_t, int, float, int32_t*, int32_t*) { PCC_NOT_YET("pcc_match_knn"); }

}  // extern "C"
