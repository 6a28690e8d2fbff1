// comm.hip -- the multi-GPU entry points of libpcc_nn: RCCL over xGMI behind the C-ABI.
//
// The path shards by independent queries: every GPU holds the whole reference cloud and its own index, queries are
// split into contiguous shards, nothing is exchanged inside a search.  What IS exchanged:
//   * the reference cloud, once: ncclBroadcast of the packed float4 array from the root's handle into every other
//     rank's, each rank then builds its own index (0.5 ms at 10M points -- cheaper than shipping the index);
//   * ICP (reference src/comparator.cpp:1089-1110) with the SOURCE sharded: one ncclAllReduce of the 17 double sums
//     per pass, on the handle's stream inside the device-resident loop, so every rank solves the same transform;
//   * SOR (src/comparator.cpp:1523-1541) with the cloud's points sharded: one all-reduce of (sum, sq_sum) and of the
//     smallest terms (MIN) that decide whether the sums are PCL's in-order ones, one of the kept count.
// One process per GPU (pcc_comm_create_rank, the unique id travels over whatever launched the ranks) or one process
// driving several GPUs from one thread per device (pcc_comm_create_local).  RCCL is loaded at the first communicator
// (dlopen: librccl.so.1 -- the copy torch has already mapped when there is one): libpcc_nn.so has no link-time
// dependency on it and single-GPU users never load it.
#include "pcc_internal.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <vector>

namespace pcc {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* h = nullptr;
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;  // a copy the process already holds (torch's)
        for (const char* n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(h, "ncclBroadcast"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.Broadcast && r.AllReduce && r.GetErrorString) r.lib = h;
    });
    return r.lib ? &r : nullptr;
}

#define PCC_NCCL(expr)                                                                                    \
    do {                                                                                                  \
        ncclResult_t _r = (expr);                                                                         \
        if (_r != ncclSuccess) {                                                                          \
            pcc::set_error("%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(_r), __FILE__, __LINE__); \
            return PCC_ERR_DEVICE;                                                                        \
        }                                                                                                 \
    } while (0)

}  // namespace pcc

// The opaque communicator of the C-ABI: one per (process, GPU).
struct pcc_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1, device = 0;
    pcc::DevBuf word;  // a few device words for scalar exchanges (cloud size, kept counts)
};

namespace pcc {

struct SetDevice {
    int prev = -1;
    bool ok = true;
    explicit SetDevice(int dev) {
        if (hipGetDevice(&prev) != hipSuccess || (prev != dev && hipSetDevice(dev) != hipSuccess)) ok = false;
    }
    ~SetDevice() {
        int cur;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

static int allreduce_sum_f64(void* ctx, double* dev, int count, hipStream_t s) {
    pcc_comm* c = static_cast<pcc_comm*>(ctx);
    PCC_NCCL(rccl()->AllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, c->nccl, s));
    return PCC_OK;
}
static int bcast_f64(void* ctx, double* dev, int count, int root, hipStream_t s) {
    pcc_comm* c = static_cast<pcc_comm*>(ctx);
    PCC_NCCL(rccl()->Broadcast(dev, dev, (size_t)count, ncclDouble, root, c->nccl, s));
    return PCC_OK;
}
static int check_comm(const pcc_comm* c) {
    if (!c || !c->nccl) { set_error("null communicator"); return PCC_ERR_INVALID; }
    return PCC_OK;
}

// One status word agreed over the ranks before a data collective: the MINIMUM of the local statuses (PCC_OK = 0, every
// error is negative), so all ranks return the same code and none is left waiting in a collective its peer never joins
// -- the reference maps every failure to a return code (src/comparator.cpp:1123,1134,1179), it never hangs.  Costs one
// 4-byte all-reduce; called at the same point of the call on every rank, with a failed rank passing its error.
static int agree_status(pcc_comm* c, int local) {
    int* w = c->word.as<int>() + 16;
    int agreed = local;
    if (hipMemcpy(w, &local, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { set_error("status exchange: hipMemcpy failed"); return PCC_ERR_DEVICE; }
    PCC_NCCL(rccl()->AllReduce(w, w, 1, ncclInt32, ncclMin, c->nccl, nullptr));
    if (hipMemcpy(&agreed, w, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { set_error("status exchange: hipMemcpy failed"); return PCC_ERR_DEVICE; }
    if (agreed != PCC_OK && local == PCC_OK) set_error("another rank of the communicator failed (status %d)", agreed);
    return agreed;
}
static int agree_hook(void* ctx, int local) { return agree_status(static_cast<pcc_comm*>(ctx), local); }

}  // namespace pcc

using namespace pcc;

extern "C" {

int pcc_comm_unique_id(void* id, size_t bytes) {
    if (!id || bytes < sizeof(ncclUniqueId)) { set_error("unique id needs %zu bytes", sizeof(ncclUniqueId)); return PCC_ERR_INVALID; }
    if (!rccl()) { set_error("librccl.so.1 not found (dlopen)"); return PCC_ERR_DEVICE; }
    ncclUniqueId u;
    PCC_NCCL(rccl()->GetUniqueId(&u));
    memset(id, 0, bytes);
    memcpy(id, &u, sizeof(u));
    return PCC_OK;
}

int pcc_comm_create_rank(const void* id, size_t bytes, int world, int rank, int device, pcc_comm** out) {
    if (!out) { set_error("null out"); return PCC_ERR_INVALID; }
    *out = nullptr;
    if (!id || bytes < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) { set_error("bad communicator arguments"); return PCC_ERR_INVALID; }
    if (!rccl()) { set_error("librccl.so.1 not found (dlopen)"); return PCC_ERR_DEVICE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { set_error("device %d out of range (%d present)", device, ndev); return PCC_ERR_INVALID; }
    SetDevice g(device);
    if (!g.ok) { set_error("hipSetDevice(%d) failed", device); return PCC_ERR_DEVICE; }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    pcc_comm* c = new pcc_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclResult_t r = rccl()->CommInitRank(&c->nccl, world, u, rank);
    if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", rccl()->GetErrorString(r)); delete c; return PCC_ERR_DEVICE; }
    if (c->word.reserve(256) != PCC_OK) { rccl()->CommDestroy(c->nccl); delete c; return PCC_ERR_NOMEM; }
    *out = c;
    return PCC_OK;
}

int pcc_comm_create_local(const int* devices, int count, pcc_comm** out) {
    if (!out || !devices || count < 1) { set_error("bad communicator arguments"); return PCC_ERR_INVALID; }
    for (int k = 0; k < count; ++k) out[k] = nullptr;
    if (!rccl()) { set_error("librccl.so.1 not found (dlopen)"); return PCC_ERR_DEVICE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (libpcc_nn has no CPU path)"); return PCC_ERR_DEVICE; }
    for (int k = 0; k < count; ++k) {
        if (devices[k] < 0 || devices[k] >= ndev) { set_error("device %d out of range (%d present)", devices[k], ndev); return PCC_ERR_INVALID; }
        for (int j = 0; j < k; ++j)
            if (devices[j] == devices[k]) { set_error("device %d listed twice: RCCL takes one rank per GPU", devices[k]); return PCC_ERR_INVALID; }
    }
    std::vector<ncclComm_t> comms((size_t)count);
    PCC_NCCL(rccl()->CommInitAll(comms.data(), count, devices));
    for (int k = 0; k < count; ++k) {
        pcc_comm* c = new pcc_comm();
        c->nccl = comms[(size_t)k];
        c->rank = k;
        c->world = count;
        c->device = devices[k];
        out[k] = c;
    }
    for (int k = 0; k < count; ++k) {  // the scalar-exchange words of every rank, or no communicator at all
        SetDevice g(devices[k]);
        if (!g.ok || out[k]->word.reserve(256) != PCC_OK) {
            if (!g.ok) set_error("hipSetDevice(%d) failed", devices[k]);
            for (int j = 0; j < count; ++j) { pcc_comm_destroy(out[j]); out[j] = nullptr; }
            return g.ok ? PCC_ERR_NOMEM : PCC_ERR_DEVICE;
        }
    }
    return PCC_OK;
}

int pcc_comm_destroy(pcc_comm* c) {
    if (!c) return PCC_OK;
    SetDevice g(c->device);
    c->word.release();
    if (c->nccl && rccl()) (void)rccl()->CommDestroy(c->nccl);
    delete c;
    return PCC_OK;
}

int pcc_comm_info(const pcc_comm* c, int* rank, int* world, int* device) {
    PCC_TRY(check_comm(c));
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    return PCC_OK;
}

// The reference cloud of `root` on every rank: root builds its index over (pts, n, stride, mem) as pcc_index_create
// does; the size and then the packed cloud (16 B per point) are broadcast over RCCL and every other rank builds its own
// index over the copy.  Collective: every rank of the communicator calls it (pts / n are read on the root only).
int pcc_index_create_broadcast(pcc_comm* c, int root, const void* pts, size_t n, size_t stride, int mem, int engine, pcc_index** out,
                               size_t* n_out) {
    if (!out) { set_error("null out"); return PCC_ERR_INVALID; }
    *out = nullptr;
    PCC_TRY(check_comm(c));
    if (root < 0 || root >= c->world) { set_error("root %d outside the communicator (%d ranks)", root, c->world); return PCC_ERR_INVALID; }
    SetDevice g(c->device);
    if (!g.ok) { set_error("hipSetDevice(%d) failed", c->device); return PCC_ERR_DEVICE; }
    pcc_index* ix = nullptr;
    unsigned long long n64 = 0;
    auto fail = [&](int st) { if (ix) pcc_index_destroy(ix); ix = nullptr; return st; };
    // step 1: the root indexes its cloud, every other rank makes its empty handle; one agreed status
    int st = c->rank == root ? pcc_index_create(pts, n, stride, 3, mem, c->device, engine, &ix) : make_handle(c->device, engine, &ix);
    if (st != PCC_OK) ix = nullptr;
    if ((st = agree_status(c, st)) != PCC_OK) return fail(st);
    // step 2: the size (a device word; the default stream orders the three steps)
    n64 = c->rank == root ? (unsigned long long)n : 0ull;
    unsigned long long* w = c->word.as<unsigned long long>();
    // (a copy that fails on one rank joins the status exchange like every other local failure: no rank enters the broadcast alone)
    st = PCC_OK;
    if (hipMemcpy(w, &n64, sizeof(n64), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed: %s", hipGetErrorString(hipGetLastError())); st = PCC_ERR_DEVICE; }
    if ((st = agree_status(c, st)) != PCC_OK) return fail(st);
    {
        ncclResult_t r = rccl()->Broadcast(w, w, 1, ncclUint64, root, c->nccl, nullptr);
        if (r != ncclSuccess) { set_error("ncclBroadcast failed: %s", rccl()->GetErrorString(r)); return fail(PCC_ERR_DEVICE); }
    }
    // step 3: room for the copy on the other ranks; one agreed status BEFORE the data collective (the read-back of the size included)
    st = PCC_OK;
    if (hipMemcpy(&n64, w, sizeof(n64), hipMemcpyDeviceToHost) != hipSuccess) { set_error("hipMemcpy failed: %s", hipGetErrorString(hipGetLastError())); st = PCC_ERR_DEVICE; }
    if (st == PCC_OK && c->rank != root) {
        std::lock_guard<std::mutex> lock(ix->mu);
        pcc::entered(ix);
        st = n64 ? ix->icp_src.reserve((size_t)n64 * sizeof(float4)) : PCC_ERR_EMPTY;
    }
    if ((st = agree_status(c, st)) != PCC_OK) return fail(st);
    // step 4: the packed cloud, 16 B per point, in one broadcast on the handle's stream; every other rank builds over its copy
    if (c->rank == root) {
        std::lock_guard<std::mutex> lock(ix->mu);
        pcc::entered(ix);
        ncclResult_t r = rccl()->Broadcast(ix->refs.p, ix->refs.p, (size_t)n64 * 4, ncclFloat, root, c->nccl, ix->stream);
        if (r != ncclSuccess) { set_error("ncclBroadcast failed: %s", rccl()->GetErrorString(r)); st = PCC_ERR_DEVICE; }
        else if (hipStreamSynchronize(ix->stream) != hipSuccess) { set_error("broadcast failed: %s", hipGetErrorString(hipGetLastError())); st = PCC_ERR_DEVICE; }
    } else {
        std::lock_guard<std::mutex> lock(ix->mu);
        pcc::entered(ix);
        ncclResult_t r = rccl()->Broadcast(ix->icp_src.p, ix->icp_src.p, (size_t)n64 * 4, ncclFloat, root, c->nccl, ix->stream);
        if (r != ncclSuccess) { set_error("ncclBroadcast failed: %s", rccl()->GetErrorString(r)); st = PCC_ERR_DEVICE; }
        // (non-finite points get their NaN back so that the build sees what the root's upload saw; then the usual build)
        if (st == PCC_OK) st = launch_nanify(ix->stream, ix->icp_src.as<float4>(), (size_t)n64);
        if (st == PCC_OK) st = set_input(ix, ix->icp_src.p, (size_t)n64, sizeof(float4), PCC_MEM_DEVICE);
        if (st == PCC_OK) st = sync_info(ix);
        if (st == PCC_OK && hipStreamSynchronize(ix->stream) != hipSuccess) { set_error("index build failed: %s", hipGetErrorString(hipGetLastError())); st = PCC_ERR_DEVICE; }
    }
    // step 5: every rank holds an index, or none does
    if ((st = agree_status(c, st)) != PCC_OK) return fail(st);
    *out = ix;
    if (n_out) *n_out = (size_t)n64;
    return PCC_OK;
}

// pcc_icp_align with the SOURCE cloud sharded over the ranks of `comm` (every rank: its own shard, its own handle over the
// same target cloud).  Collective.  T, fitness (mean d2 over ALL shards), iterations, converged: the same on every rank.
int pcc_icp_align_sharded(pcc_index* ix, pcc_comm* c, const void* src_shard, size_t n, size_t stride, int mem, int max_iter, int fixed,
                          float T[16], double* fitness, int* iterations, int* converged) {
    PCC_TRY(check_comm(c));
    // (a rank that fails here joins the one status exchange icp_align_impl makes before its first collective)
    if (!ix) { set_error("null index"); return agree_status(c, PCC_ERR_INVALID); }
    if (ix->device != c->device) { set_error("index on device %d, communicator on device %d", ix->device, c->device); return agree_status(c, PCC_ERR_INVALID); }
    const IcpHooks hooks{c, allreduce_sum_f64, bcast_f64, agree_hook};
    return icp_align_impl(ix, &hooks, src_shard, n, stride, mem, max_iter, fixed, T, fitness, iterations, converged);
}

// ---- SOR over a shard of the indexed cloud --------------------------------------------------------------------------
// mean distances of the points [start, start + count) of the indexed cloud (self query with mean_k + 1 neighbours, as
// pcc_sor) and this shard's share of PCL's statistics: sums[0] = sum of the means, [1] = sum of their float squares,
// [2], [3] = bit patterns (as doubles) of the smallest positive term of either sum (+inf's pattern when there is none).
// Combine over shards with (+, +, min, min) and hand the result to pcc_sor_threshold.
static int sor_shard_means(pcc_index* ix, size_t start, size_t count, int mean_k, float** dmean_out) {
    if (mean_k < 1 || mean_k + 1 > PCC_KNN_MAX_K) { set_error("mean_k=%d outside [1, %d]", mean_k, PCC_KNN_MAX_K - 1); return PCC_ERR_UNSUPPORTED; }
    if (start > ix->n_orig || count > ix->n_orig - start) { set_error("shard [%zu, %zu) outside the cloud (%zu points)", start, start + count, ix->n_orig); return PCC_ERR_INVALID; }
    PCC_TRY(need_grid(ix));
    const int K = mean_k + 1;
    const bool d2_only = grid_knn_delivers(K) && (size_t)2 * 64 * (K + 1) * sizeof(unsigned int) <= 64 * 1024;
    PCC_TRY(ix->out_packed.reserve((count + 1) * (size_t)K * (d2_only ? sizeof(float) : sizeof(unsigned long long))));
    auto* keys = d2_only ? nullptr : ix->out_packed.as<unsigned long long>();
    float* d2_rows = d2_only ? ix->out_packed.as<float>() : nullptr;
    const float4* q = ix->refs.as<float4>() + start;
    PCC_TRY(ix->out_d2.reserve((count + 1) * sizeof(float)));
    float* dmean = ix->out_d2.as<float>();
    if (count) {
        PCC_TRY(grid_knn(ix, q, count, K, keys, nullptr, d2_rows));
        PCC_HIP(hipMemsetAsync(dmean, 0, count * sizeof(float), ix->stream));
        PCC_TRY(launch_sor_mean(ix->stream, keys, q, count, K, dmean, d2_rows));
    }
    *dmean_out = dmean;
    return PCC_OK;
}

int pcc_sor_partial(pcc_index* ix, size_t start, size_t count, int mean_k, int mem, float* mean_dist, double sums[4]) {
    if (!ix) { set_error("null index"); return PCC_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(ix->mu);
    pcc::entered(ix);
    SetDevice g(ix->device);
    if (mem != PCC_MEM_HOST && mem != PCC_MEM_DEVICE) { set_error("bad mem space"); return PCC_ERR_INVALID; }
    if (!sums) { set_error("null sums"); return PCC_ERR_INVALID; }
    float* dmean = nullptr;
    PCC_TRY(sor_shard_means(ix, start, count, mean_k, &dmean));
    PCC_TRY(ix->scratch_a.reserve((size_t)(3 * 1024 + 4) * sizeof(double) + 64));
    double* out4 = ix->scratch_a.as<double>() + 3 * 1024;
    PCC_TRY(launch_sor_partial(ix->stream, dmean, count, ix->scratch_a.as<double>(), out4));
    PCC_HIP(hipMemcpyAsync(sums, out4, 4 * sizeof(double), hipMemcpyDeviceToHost, ix->stream));
    if (mean_dist && count)
        PCC_HIP(hipMemcpyAsync(mean_dist, dmean, count * sizeof(float), mem == PCC_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ix->stream));
    PCC_HIP(hipStreamSynchronize(ix->stream));
    return PCC_OK;
}

// PCL's threshold from the combined sums of all shards (pure host arithmetic, no handle): *exact = 0 says that PCL's
// in-order additions would round -- the combined sums then need not be PCL's bits and the caller should take the
// sums of all mean distances in index order instead (pcc_sor does that on one GPU).
int pcc_sor_threshold(const double sums[4], uint64_t n_valid, int mean_k, double stddev_mult, double* threshold, int* exact) {
    if (!sums || !threshold || !exact) { set_error("null argument"); return PCC_ERR_INVALID; }
    sor_threshold_host(sums, (double)n_valid, mean_k + 1, stddev_mult, threshold, exact);
    return PCC_OK;
}

// pcc_sor over the ranks of `comm`: this rank filters the points [start, start + count) of the indexed cloud (the shards
// of all ranks tile the cloud), the statistics are PCL's over the WHOLE cloud.  mean_dist / inlier: `count` entries in
// memory space `mem`; *threshold and *kept_total (inliers over all shards) are the same on every rank.  Collective.
int pcc_sor_sharded(pcc_index* ix, pcc_comm* c, size_t start, size_t count, int mean_k, double stddev_mult, int mem,
                    float* mean_dist, uint8_t* inlier, double* threshold, size_t* kept_total) {
    PCC_TRY(check_comm(c));
    // (every failure of one rank alone ends in the status exchange below, never in a peer waiting for its all-reduce)
    if (!ix) { set_error("null index"); return agree_status(c, PCC_ERR_INVALID); }
    if (ix->device != c->device) { set_error("index on device %d, communicator on device %d", ix->device, c->device); return agree_status(c, PCC_ERR_INVALID); }
    if (mem != PCC_MEM_HOST && mem != PCC_MEM_DEVICE) { set_error("bad mem space"); return agree_status(c, PCC_ERR_INVALID); }
    std::lock_guard<std::mutex> lock(ix->mu);
    pcc::entered(ix);
    SetDevice g(ix->device);
    hipStream_t s = ix->stream;
    const int K = mean_k + 1;
    float* dmean = nullptr;
    double* out4 = nullptr;
    uint8_t* dmask = nullptr;
    auto prepare = [&]() -> int {  // shard search + every allocation of the exact path
        if (!g.ok) { set_error("hipSetDevice(%d) failed", ix->device); return PCC_ERR_DEVICE; }
        PCC_TRY(sor_shard_means(ix, start, count, mean_k, &dmean));
        PCC_TRY(ix->scratch_a.reserve((size_t)(3 * 1024 + 4) * sizeof(double) + 64));
        out4 = ix->scratch_a.as<double>() + 3 * 1024;
        PCC_TRY(ix->scratch_b.reserve(count + 64));
        dmask = mem == PCC_MEM_DEVICE && inlier ? inlier : ix->scratch_b.as<uint8_t>();
        PCC_TRY(launch_sor_partial(s, dmean, count, ix->scratch_a.as<double>(), out4));
        return PCC_OK;
    };
    {
        const int st = agree_status(c, prepare());
        if (st != PCC_OK) return st;
    }
    // (A collective RCCL refuses to enqueue -- PCC_NCCL -- is not a failure of this rank alone: the communicator is broken for
    // every rank and the call returns PCC_ERR_DEVICE; what CAN fail locally between two collectives -- a launch, a copy, an
    // allocation -- is folded into an agreed status before the next one.)
    PCC_NCCL(rccl()->AllReduce(out4, out4, 2, ncclDouble, ncclSum, c->nccl, s));
    PCC_NCCL(rccl()->AllReduce(out4 + 2, out4 + 2, 2, ncclDouble, ncclMin, c->nccl, s));
    struct { double sum, sq, thr; unsigned long long kept; unsigned int exact, pad; } hs{};
    void* st_dev = ix->small.as<char>() + 256;
    {
        const int st = agree_status(c, launch_sor_threshold_mask(s, dmean, count, ix->d_grid.as<GridDev>(), K, stddev_mult, out4, st_dev, dmask));
        if (st != PCC_OK) return st;
    }
    unsigned long long* kept_dev = reinterpret_cast<unsigned long long*>(static_cast<char*>(st_dev) + 24);
    PCC_NCCL(rccl()->AllReduce(kept_dev, kept_dev, 1, ncclUint64, ncclSum, c->nccl, s));
    {   // (hs.exact decides whether the big all-reduce below happens: every rank must have read it)
        int st = PCC_OK;
        if (hipMemcpyAsync(&hs, st_dev, sizeof(hs), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            set_error("SOR statistics read-back failed: %s", hipGetErrorString(hipGetLastError()));
            st = PCC_ERR_DEVICE;
        }
        if ((st = agree_status(c, st)) != PCC_OK) return st;
    }
    double thr = hs.thr;
    size_t kept = (size_t)hs.kept;
    if (!hs.exact) {
        // (the same verdict on every rank) PCL's order decides the last bits: every rank gets ALL mean distances -- its own
        // shard in a zeroed array of the cloud's size, summed over the ranks (adding zeros is exact) -- and walks them in order
        const size_t no = ix->n_orig;
        {  // (hs.exact is the same on every rank, so every rank is here: agree on the allocations before the big all-reduce)
            int st = ix->scratch_c.reserve(no * sizeof(float) + 64);
            if (st == PCC_OK) st = ix->host_a.reserve(no * sizeof(float));
            if (st == PCC_OK) st = ix->host_b.reserve(count + 64);
            float* all0 = ix->scratch_c.as<float>();
            if (st == PCC_OK && (hipMemsetAsync(all0, 0, no * sizeof(float), s) != hipSuccess ||
                                 (count && hipMemcpyAsync(all0 + start, dmean, count * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess))) {
                set_error("SOR: staging the shard's mean distances failed: %s", hipGetErrorString(hipGetLastError()));
                st = PCC_ERR_DEVICE;
            }
            if ((st = agree_status(c, st)) != PCC_OK) return st;
        }
        float* all = ix->scratch_c.as<float>();
        PCC_NCCL(rccl()->AllReduce(all, all, no, ncclFloat, ncclSum, c->nccl, s));
        float* hm = ix->host_a.as<float>();
        PCC_HIP(hipMemcpyAsync(hm, all, no * sizeof(float), hipMemcpyDeviceToHost, s));
        PCC_HIP(hipStreamSynchronize(s));
        PCC_TRY(sync_info(ix));
        const size_t valid = ix->n_valid >= (size_t)K ? ix->n_valid : 0;
        double sum = 0, sq = 0;
        for (size_t i = 0; i < no; ++i) { const float f = hm[i]; sum += f; sq += (double)(f * f); }
        const double mean = sum / (double)valid;
        const double var = (sq - sum * sum / (double)valid) / ((double)valid - 1);
        thr = mean + stddev_mult * std::sqrt(var);
        kept = 0;
        for (size_t i = 0; i < no; ++i) kept += !(hm[i] > thr);
        uint8_t* hin = ix->host_b.as<uint8_t>();
        for (size_t i = 0; i < count; ++i) hin[i] = !(hm[start + i] > thr);
        if (count) PCC_HIP(hipMemcpyAsync(dmask, hin, count, hipMemcpyHostToDevice, s));
        PCC_HIP(hipStreamSynchronize(s));
    }
    ix->sor_exact_last = hs.exact != 0;
    if (threshold) *threshold = thr;
    if (kept_total) *kept_total = kept;
    if (count) {
        const hipMemcpyKind kind = mem == PCC_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
        if (mean_dist) PCC_HIP(hipMemcpyAsync(mean_dist, dmean, count * sizeof(float), kind, s));
        if (inlier && dmask != inlier) PCC_HIP(hipMemcpyAsync(inlier, dmask, count, kind, s));
        PCC_HIP(hipStreamSynchronize(s));
    }
    return PCC_OK;
}

}  // extern "C"
