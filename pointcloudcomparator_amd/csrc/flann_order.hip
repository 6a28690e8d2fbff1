// flann_order.hip -- PCC_TIES_FLANN on the device: the kernels that flag the queries whose minimum distance is shared by
// a second reference, and the kernel that walks those through a kd-tree of FLANN's shape (flann_tree.hpp: built on the
// host once per indexed cloud, uploaded as two flat arrays).  Nothing is copied back per call and the host never waits:
// a search in FLANN mode costs the two kernels.  Compiled with -ffp-contract=off: the distances must round like FLANN's
// L2_Simple<float>.  Reference call sites: src/comparator.cpp:576-580 (the indices matchRIFTFeaturesKnn hands on).
#include "pcc_internal.hpp"
#include "grid_device.hpp"
#include <algorithm>
#include <cfloat>
#include <cstring>
#include <thread>
#include <vector>

namespace pcc {

// ---- which queries have a tie ------------------------------------------------------------------------------
// keys[i] = (d2 bits << 32 | index) of the nearest reference.  A query is TIED when some OTHER reference has exactly
// the same d2 (or when that cannot be decided cheaply: ball wider than the cell walk allows).  The tied queries are
// appended to a compact list (one returning atomic per wave, PCC_TIE_SHARDS counters each owning a slice of the list):
// the tree walk then runs with every lane busy instead of one lane in ten.
__device__ __forceinline__ void tie_append(bool tie, unsigned int i, unsigned int* __restrict__ list, unsigned int* __restrict__ counters,
                                           unsigned int shard_cap) {
    // every lane of the wave must arrive here together (the callers do not return before it); the leader is still taken
    // from the lanes that ARE here -- readfirstlane reads the first active lane -- so a caller that diverged would lose
    // nothing: lane 0 need not be among them
    const unsigned long long m = __ballot(tie);
    if (m == 0ull) return;
    const unsigned int shard = blockIdx.x % PCC_TIE_SHARDS;
    const unsigned int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const unsigned int leader = (unsigned int)__builtin_ctzll(__ballot(true));
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(counters + shard * PCC_OPEN_CTR_STRIDE, (unsigned int)__popcll(m));
    base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
    if (tie) list[shard * shard_cap + base + __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] = i;
}

__global__ void __launch_bounds__(256)
k_tie_flags_grid(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
                 const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned long long* __restrict__ keys,
                 unsigned int n, unsigned int* __restrict__ list, unsigned int* __restrict__ counters, unsigned int shard_cap) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    const GridParams g = gd->g;
    const float4 qv = i < n ? q[i] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    const unsigned long long key = i < n ? keys[i] : ~0ull;
    // a non-finite query or one without a neighbour has no tie; it skips the scan but NOT the wave-wide append below
    const bool valid = !(__float_as_int(qv.w) < 0 || key_none(key));
    const float bd = __uint_as_float((unsigned int)(key >> 32));
    const unsigned int bi = (unsigned int)key;
    const float rb = valid ? sqrtf(bd) * 1.00001f + gd->slack : 0.f;
    int x0 = 0, x1 = -1, y0 = 0, y1 = -1, z0 = 0, z1 = -1;
    bool tie = false;
    if (valid) {
        float ux, uy, uz;  // the query in the grid's frame (grid_device.hpp)
        grid_frame(g, qv.x, qv.y, qv.z, ux, uy, uz);
        cell_range(ux, rb, g.org[0], g.inv_h, g.dim[0], x0, x1);
        cell_range(uy, rb, g.org[1], g.inv_h, g.dim[1], y0, y1);
        cell_range(uz, rb, g.org[2], g.inv_h, g.dim[2], z0, z1);
        tie = !(rb < __builtin_inff()) || (long long)(x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1) > 4096;
    }
    for (int z = z0; z <= z1 && !tie; ++z)
        for (int y = y0; y <= y1 && !tie; ++y) {
            const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
            const unsigned int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
            for (unsigned int p = s; p < e; ++p) {
                const float4 r = cell_refs[p];
                if (dist2(qv.x, qv.y, qv.z, r) == bd && (unsigned int)__float_as_int(r.w) != bi) { tie = true; break; }
            }
        }
    tie_append(tie, i, list, counters, shard_cap);
}

// exhaustive form (BRUTE engine: small clouds): one lane per query over all references
__global__ void __launch_bounds__(256)
k_tie_flags_brute(const float4* __restrict__ refs, unsigned int m, const float4* __restrict__ q,
                  const unsigned long long* __restrict__ keys, unsigned int n, unsigned int* __restrict__ list,
                  unsigned int* __restrict__ counters, unsigned int shard_cap) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 qv = i < n ? q[i] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    const unsigned long long key = i < n ? keys[i] : ~0ull;
    const bool valid = !(__float_as_int(qv.w) < 0 || key_none(key));  // (no early return: the append below is wave-wide)
    const float bd = __uint_as_float((unsigned int)(key >> 32));
    const unsigned int bi = (unsigned int)key;
    bool tie = false;
    for (unsigned int p = 0; p < m; ++p) {  // wave-uniform address: one broadcast load per reference
        const float4 r = refs[p];
        if (__float_as_int(r.w) >= 0 && p != bi && dist2(qv.x, qv.y, qv.z, r) == bd) tie = true;
    }
    tie_append(tie && valid, i, list, counters, shard_cap);
}

// list: PCC_TIE_SHARDS slices of shard_cap entries; counters[shard * PCC_OPEN_CTR_STRIDE] = entries of the slice (pre-zeroed)
static int launch_tie_list(pcc_index* ix, const float4* q, const unsigned long long* keys, size_t nq, unsigned int* list,
                           unsigned int* counters, unsigned int shard_cap) {
    const unsigned int n = (unsigned int)nq, blocks = (n + 255) / 256;
    if (ix->engine == PCC_ENGINE_GRID && ix->has_grid)
        hipLaunchKernelGGL(k_tie_flags_grid, dim3(blocks), dim3(256), 0, ix->stream, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, keys, n, list, counters, shard_cap);
    else
        hipLaunchKernelGGL(k_tie_flags_brute, dim3(blocks), dim3(256), 0, ix->stream, ix->refs.as<float4>(),
                           (unsigned int)ix->n_orig, q, keys, n, list, counters, shard_cap);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}


// flagged queries through the tree: the index part of a key is replaced when FLANN's walk names another reference at
// the SAME distance bits (same arithmetic => same minimum; should the bits ever differ the GPU result stands).
// counters: PCC_TIE_SHARDS x {flagged, changed, too deep}, one 128-byte line per shard
template <int STACK>
__global__ void __launch_bounds__(256)
k_tie_walk(const FlannNode* __restrict__ nodes, const float* __restrict__ leaf_pts, FlannBox root, unsigned int n_valid,
           const float4* __restrict__ q, unsigned long long* __restrict__ keys, const unsigned int* __restrict__ list,
           unsigned int shard_cap, unsigned int* __restrict__ counters) {
    // block b works on slice b % PCC_TIE_SHARDS of the list, as chunk b / PCC_TIE_SHARDS of gridDim / PCC_TIE_SHARDS
    const unsigned int shard = blockIdx.x % PCC_TIE_SHARDS, chunk = blockIdx.x / PCC_TIE_SHARDS, nchunk = gridDim.x / PCC_TIE_SHARDS;
    unsigned int* c = counters + shard * PCC_OPEN_CTR_STRIDE;
    const unsigned int cnt = c[0];
    for (unsigned int j0 = chunk * blockDim.x; j0 < cnt; j0 += nchunk * blockDim.x) {  // (block-uniform trip count)
        const unsigned int j = j0 + threadIdx.x;
        bool changed = false;
        if (j < cnt) {
            const unsigned int i = list[shard * shard_cap + j];
            const float4 qv = q[i];
            const unsigned long long key = keys[i];
            // the short walk first (the minimum distance is known: straight to the first reference at it); the full walk
            // only where the short one cannot vouch for its answer (flann_tree.hpp)
            const float bd = __uint_as_float((unsigned int)(key >> 32));
            bool unc = true;
            int32_t fi = flann_walk_tied<24>(nodes, leaf_pts, root, n_valid, qv.x, qv.y, qv.z, bd, &unc);
            float d2 = bd;
            if (unc) fi = flann_walk<STACK>(nodes, leaf_pts, root, n_valid, qv.x, qv.y, qv.z, &d2);
            if (fi >= 0 && __float_as_uint(d2) == (unsigned int)(key >> 32) && (unsigned int)fi != (unsigned int)key) {
                keys[i] = (key & 0xffffffff00000000ull) | (unsigned int)fi;
                changed = true;
            }
        }
        const unsigned long long mc = __ballot(changed);
        if ((threadIdx.x & 63) == 0 && mc) atomicAdd(c + 1, (unsigned int)__popcll(mc));
    }
}

// FLANN's tree over the indexed cloud, from the packed copy on the device; uploaded as flat arrays
static int ensure_tree(pcc_index* ix) {
    if (ix->flann_valid) return PCC_OK;
    PCC_TRY(ix->host_a.reserve(ix->n_orig * sizeof(float4)));
    PCC_HIP(hipMemcpyAsync(ix->host_a.p, ix->refs.p, ix->n_orig * sizeof(float4), hipMemcpyDeviceToHost, ix->stream));
    PCC_HIP(hipStreamSynchronize(ix->stream));
    unsigned int threads = std::thread::hardware_concurrency();
    threads = threads < 1 ? 1 : (threads > 32 ? 32 : threads);
    // (threads: the build forks at its top levels while a side holds more than 4096 points; 18 381 records 1.3 -> 0.55 ms with four)
    ix->flann.build(ix->host_a.as<float>(), ix->n_orig, ix->opt.flann_split, ix->n_orig >= 50000 ? threads : (ix->n_orig >= 6000 ? std::min(threads, 4u) : 1u));
    PCC_TRY(ix->flann_nodes.reserve(ix->flann.nodes.size() * sizeof(FlannNode) + 16));
    PCC_TRY(ix->flann_leaf.reserve(ix->flann.leaf_pts.size() * sizeof(float) + 16));
    PCC_HIP(hipMemcpyAsync(ix->flann_nodes.p, ix->flann.nodes.data(), ix->flann.nodes.size() * sizeof(FlannNode), hipMemcpyHostToDevice, ix->stream));
    PCC_HIP(hipMemcpyAsync(ix->flann_leaf.p, ix->flann.leaf_pts.data(), ix->flann.leaf_pts.size() * sizeof(float), hipMemcpyHostToDevice, ix->stream));
    PCC_HIP(hipStreamSynchronize(ix->stream));  // (the vectors are pageable host memory)
    if (ix->flann.depth <= FLANN_DEV_STACK_MAX) {  // the host copy is only needed for trees the device stack cannot hold
        std::vector<FlannNode>().swap(ix->flann.nodes);
        std::vector<float>().swap(ix->flann.leaf_pts);
    }
    ix->flann_valid = true;
    return PCC_OK;
}

// PCC_TIES_FLANN: rewrite the index part of the keys of tied queries (keys of the queries in q, nq of them)
// may_wait: the caller waits for the stream anyway (results in host memory).  Then, as long as the indexed cloud has no tree yet, the
// number of tied queries is read back first and a call without any skips the tree altogether -- the host build is 1.3 ms for the
// reference's largest descriptor cloud (18 381 records), per indexed cloud, i.e. per call in matchRIFTFeaturesKnn's pattern.
int resolve_ties_flann(pcc_index* ix, const float4* q, unsigned long long* keys, size_t nq, bool may_wait) {
    if (nq == 0) return PCC_OK;
    const unsigned int n = (unsigned int)nq, blocks = (n + 255) / 256;
    const unsigned int shard_cap = (blocks + PCC_TIE_SHARDS - 1) / PCC_TIE_SHARDS * 256;  // what a slice's blocks could append
    PCC_TRY(ix->tie_buf.reserve((size_t)shard_cap * PCC_TIE_SHARDS * sizeof(unsigned int) + 256));
    unsigned int* list = ix->tie_buf.as<unsigned int>();
    unsigned int* counters = ix->small.as<unsigned int>() + PCC_TIE_CTR0;
    PCC_HIP(hipMemsetAsync(counters, 0, PCC_TIE_SHARDS * PCC_OPEN_CTR_STRIDE * 4, ix->stream));
    PCC_TRY(launch_tie_list(ix, q, keys, nq, list, counters, shard_cap));
    if (may_wait && !ix->flann_valid) {
        unsigned int h[PCC_TIE_SHARDS * PCC_OPEN_CTR_STRIDE];
        PCC_HIP(hipMemcpyAsync(h, counters, sizeof(h), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipStreamSynchronize(ix->stream));
        unsigned int flagged = 0;
        for (int sh = 0; sh < PCC_TIE_SHARDS; ++sh) flagged += h[sh * PCC_OPEN_CTR_STRIDE];
        if (flagged == 0) {  // every minimum is unique: the lowest index is FLANN's answer
            ix->ties_pending = true;  // (the counters are all zero)
            return PCC_OK;
        }
    }
    PCC_TRY(ensure_tree(ix));
    const FlannNode* nodes = ix->flann_nodes.as<FlannNode>();
    const float* leaf = ix->flann_leaf.as<float>();
    const unsigned int nv = (unsigned int)ix->flann.n_valid;
    // blocks per slice: enough for every query to be listed without a second trip, at most 32
    unsigned int chunks = (shard_cap / 8 + 255) / 256;
    chunks = chunks < 1 ? 1 : (chunks > 32 ? 32 : chunks);
    const dim3 wg(chunks * PCC_TIE_SHARDS);
    // the walk defers one far child per level: a stack of the tree's depth always suffices
    if (ix->flann.depth <= 48)
        hipLaunchKernelGGL(k_tie_walk<48>, wg, dim3(256), 0, ix->stream, nodes, leaf, ix->flann.root, nv, q, keys, list, shard_cap, counters);
    else if (ix->flann.depth <= 128)
        hipLaunchKernelGGL(k_tie_walk<128>, wg, dim3(256), 0, ix->stream, nodes, leaf, ix->flann.root, nv, q, keys, list, shard_cap, counters);
    else if (ix->flann.depth <= FLANN_DEV_STACK_MAX)
        hipLaunchKernelGGL(k_tie_walk<FLANN_DEV_STACK_MAX>, wg, dim3(256), 0, ix->stream, nodes, leaf, ix->flann.root, nv, q, keys, list, shard_cap, counters);
    else {
        // a tree deeper than any device stack (coordinates spread over hundreds of binades): the host walks the listed
        // queries -- synchronous, and only here
        std::vector<unsigned int> hl((size_t)shard_cap * PCC_TIE_SHARDS), hc(PCC_TIE_SHARDS * PCC_OPEN_CTR_STRIDE);
        std::vector<float4> hq(nq);
        std::vector<unsigned long long> hk(nq);
        PCC_HIP(hipMemcpyAsync(hl.data(), list, hl.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipMemcpyAsync(hc.data(), counters, hc.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipMemcpyAsync(hq.data(), q, nq * sizeof(float4), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipMemcpyAsync(hk.data(), keys, nq * sizeof(unsigned long long), hipMemcpyDeviceToHost, ix->stream));
        PCC_HIP(hipStreamSynchronize(ix->stream));
        unsigned int n_changed = 0;
        for (int sh = 0; sh < PCC_TIE_SHARDS; ++sh)
            for (unsigned int j = 0; j < hc[sh * PCC_OPEN_CTR_STRIDE]; ++j) {
                const unsigned int i = hl[(size_t)sh * shard_cap + j];
                const float qv[3] = {hq[i].x, hq[i].y, hq[i].z};
                float d2 = 0.f;
                const int32_t fi = ix->flann.nearest(qv, &d2);
                uint32_t bits;
                memcpy(&bits, &d2, 4);
                if (fi >= 0 && bits == (uint32_t)(hk[i] >> 32) && (uint32_t)fi != (uint32_t)hk[i]) {
                    hk[i] = (hk[i] & 0xffffffff00000000ull) | (uint32_t)fi;
                    ++n_changed;
                }
            }
        if (n_changed) {
            PCC_HIP(hipMemcpyAsync(keys, hk.data(), nq * sizeof(unsigned long long), hipMemcpyHostToDevice, ix->stream));
            PCC_HIP(hipMemcpyAsync(counters + 1, &n_changed, 4, hipMemcpyHostToDevice, ix->stream));
        }
        PCC_HIP(hipStreamSynchronize(ix->stream));
    }
    PCC_HIP(hipGetLastError());
    ix->ties_pending = true;  // pcc_index_stats adds the shards up
    return PCC_OK;
}

// ---- the small-call form ------------------------------------------------------------------------------------------------
// one lane per query of the call; the flagged ones take k_tie_walk's walk, statement for statement
template <int STACK>
__global__ void __launch_bounds__(64)
k_small_tie_walk(const FlannNode* __restrict__ nodes, const float* __restrict__ leaf_pts, FlannBox root, unsigned int n_valid,
                 const float4* __restrict__ q, unsigned long long* __restrict__ keys, const unsigned char* __restrict__ tie_q, unsigned int nq,
                 int32_t* __restrict__ idx_host, unsigned int* __restrict__ changed_blocks) {
    const unsigned int i = blockIdx.x * 64u + threadIdx.x;
    bool changed = false;
    if (i < nq && tie_q[i]) {
        const float4 qv = q[i];
        const unsigned long long key = keys[i];
        const float bd = __uint_as_float((unsigned int)(key >> 32));
        bool unc = true;
        int32_t fi = flann_walk_tied<24>(nodes, leaf_pts, root, n_valid, qv.x, qv.y, qv.z, bd, &unc);
        float d2 = bd;
        if (unc) fi = flann_walk<STACK>(nodes, leaf_pts, root, n_valid, qv.x, qv.y, qv.z, &d2);
        if (fi >= 0 && __float_as_uint(d2) == (unsigned int)(key >> 32) && (unsigned int)fi != (unsigned int)key) {
            keys[i] = (key & 0xffffffff00000000ull) | (unsigned int)fi;
            idx_host[i] = fi;
            changed = true;
        }
    }
    const unsigned long long mc = __ballot(changed);
    if (threadIdx.x == 0) changed_blocks[blockIdx.x] = (unsigned int)__popcll(mc);
}

int small_tie_replay(pcc_index* ix, const void* raw_refs, size_t nq, const unsigned char* tie_q, int32_t* idx_host,
                     unsigned int* changed_blocks, bool* done) {
    *done = false;
    if (!ix->flann_valid) {
        // FLANN's tree from the raw records the small-call path keeps in pinned memory (k_pack's finite test, k_pack's flags): no
        // download; the flat arrays go up from a pinned staging buffer without a wait
        const size_t n = ix->n_orig, stride = ix->small_raw_stride;
        std::vector<float> packed(n * 4);
        const char* raw = static_cast<const char*>(raw_refs);
        for (size_t i = 0; i < n; ++i) {
            const float* p = reinterpret_cast<const float*>(raw + i * stride);
            const float x = p[0], y = p[1], z = p[2];
            const bool fin = (x - x) == 0.0f && (y - y) == 0.0f && (z - z) == 0.0f;
            const int32_t w = fin ? (int32_t)i : -1;
            float* o = &packed[i * 4];
            o[0] = fin ? x : 0.f;
            o[1] = fin ? y : 0.f;
            o[2] = fin ? z : 0.f;
            memcpy(o + 3, &w, 4);
        }
        ix->flann.build(packed.data(), n, ix->opt.flann_split, 1);
        if (ix->flann.depth > 128) return PCC_OK;  // (resolve_ties_flann builds it again its own way: deep trees are its business)
        const size_t nb = ix->flann.nodes.size() * sizeof(FlannNode), lb = ix->flann.leaf_pts.size() * sizeof(float);
        const size_t nb_al = (nb + 15) & ~(size_t)15;
        PCC_TRY(ix->host_c.reserve(nb_al + lb + 16));
        PCC_TRY(ix->flann_nodes.reserve(nb + 16));
        PCC_TRY(ix->flann_leaf.reserve(lb + 16));
        memcpy(ix->host_c.p, ix->flann.nodes.data(), nb);
        memcpy(ix->host_c.as<char>() + nb_al, ix->flann.leaf_pts.data(), lb);
        if (nb) PCC_HIP(hipMemcpyAsync(ix->flann_nodes.p, ix->host_c.p, nb, hipMemcpyHostToDevice, ix->stream));
        if (lb) PCC_HIP(hipMemcpyAsync(ix->flann_leaf.p, ix->host_c.as<char>() + nb_al, lb, hipMemcpyHostToDevice, ix->stream));
        std::vector<FlannNode>().swap(ix->flann.nodes);
        std::vector<float>().swap(ix->flann.leaf_pts);
        ix->flann_valid = true;  // (valid for whoever is enqueued behind the two copies; the staging buffer is not touched before the
                                 // next set_input, and every small call ends with a wait)
    }
    if (ix->flann.depth > 128) return PCC_OK;
    const unsigned int blocks = (unsigned int)((nq + 63) / 64);
    const FlannNode* nodes = ix->flann_nodes.as<FlannNode>();
    const float* leaf = ix->flann_leaf.as<float>();
    const unsigned int nv = (unsigned int)ix->flann.n_valid;
    if (ix->flann.depth <= 48)
        hipLaunchKernelGGL(k_small_tie_walk<48>, dim3(blocks), dim3(64), 0, ix->stream, nodes, leaf, ix->flann.root, nv, ix->q_packed.as<float4>(),
                           ix->out_packed.as<unsigned long long>(), tie_q, (unsigned int)nq, idx_host, changed_blocks);
    else
        hipLaunchKernelGGL(k_small_tie_walk<128>, dim3(blocks), dim3(64), 0, ix->stream, nodes, leaf, ix->flann.root, nv, ix->q_packed.as<float4>(),
                           ix->out_packed.as<unsigned long long>(), tie_q, (unsigned int)nq, idx_host, changed_blocks);
    PCC_HIP(hipGetLastError());
    *done = true;
    return PCC_OK;
}

PCC_PAIRS_TAKE(flann)

}  // namespace pcc
