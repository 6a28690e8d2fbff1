// small.hip -- the SMALL-CALL form of the k = 1 search (gfx950): one launch per query call.
//
// The reference's descriptor matching (matchRIFTFeaturesKnn, src/comparator.cpp:560-588) builds a tree over 4 ... 18 381
// descriptors and asks once per descriptor of a second cloud, up to 300 times per comparison.  At these sizes a call is not
// arithmetic, it is launches: pack the queries, preset the keys, search, unpack -- four launches of a few microseconds each
// behind the index's pack, 60 us a call against 8 us for a 100-point kd-tree on the CPU (profiles/r06_exp_small_calls.txt).
// k_small_nn1 does the query side in ONE launch: a workgroup takes 64 queries straight from the pinned host copy of the
// caller's array (wave 0 packs them: the first three floats of every record, non-finite ones flagged, as k_pack), its 16 waves
// share the references (a wave's reference is wave-uniform: scalar loads), the partial minima meet in LDS, and wave 0 writes the
// result arrays into pinned host memory as k_unpack would -- plus q_packed / out_packed, so that the handle is left in the state
// the four launches leave it in.
//
// Same arithmetic as the exhaustive kernel (nn1_brute.hip): d = dx * dx; d += dy * dy; d += dz * dz, every operation rounded
// (-ffp-contract=off), key = (bits(d) << 32) | position, smallest key wins -- the smallest distance, the lowest index among equals;
// non-finite references never take part; distances that overflow to +inf leave the first valid reference, which k_unpack's rule
// (d2 bits >= 0x7f7fffff: nothing found) then reports as -1 / +inf, exactly as the exhaustive path does.
#include "pcc_internal.hpp"

namespace pcc {

constexpr int SMALL_WAVES = 16;  // waves of a workgroup: each takes a sixteenth of the references for the same 64 queries

// TIES (PCC_TIES_FLANN): a query whose minimum distance is shared by a second reference is TIED -- FLANN's tree walk may return
// another index than the lowest (flann_order.hip).  The kernel counts the tied queries of every workgroup into tie_blocks[] (pinned
// host memory); with none anywhere -- real descriptors: exact float ties are rare -- the lowest index IS FLANN's answer and the call
// is complete; otherwise the flagged queries (tie_q[], device memory) are walked through FLANN's tree by one more launch
// (flann_order.hip: small_tie_replay), or the caller replays the ties on q_packed / out_packed as after the separate launches.
template <bool TIES>
__global__ void __launch_bounds__(SMALL_WAVES * 64)
k_small_nn1(const char* __restrict__ raw_q, unsigned int nq, size_t stride, const float4* __restrict__ refs, unsigned int n,
            float4* __restrict__ q_packed, unsigned long long* __restrict__ out_packed, int32_t* __restrict__ idx,
            float* __restrict__ d2, unsigned int* __restrict__ tie_blocks, unsigned char* __restrict__ tie_q) {
    __shared__ float4 sq[64];
    __shared__ unsigned long long sk[SMALL_WAVES][64];
    __shared__ unsigned char st[SMALL_WAVES][64];
    const unsigned int lane = threadIdx.x & 63;
    const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned int i = blockIdx.x * 64u + lane;
    if (wave == 0) {
        float4 o = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        if (i < nq) {
            const float* p = reinterpret_cast<const float*>(raw_q + (size_t)i * stride);
            const float x = p[0], y = p[1], z = p[2];
            if ((x - x) == 0.0f && (y - y) == 0.0f && (z - z) == 0.0f) o = make_float4(x, y, z, __int_as_float((int)i));
            q_packed[i] = o;
        }
        sq[lane] = o;
    }
    __syncthreads();
    const float4 qv = sq[lane];
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    const unsigned int per = (n + SMALL_WAVES - 1) / SMALL_WAVES;
    const unsigned int r0 = min(n, wave * per), r1 = min(n, r0 + per);
    unsigned long long key = ~0ull;
    bool tie = false;  // a second reference of this wave's slice at the slice's minimum distance
#pragma unroll 4
    for (unsigned int j = r0; j < r1; ++j) {  // (wave-uniform: the reference arrives by a scalar load)
        const float4 r = refs[j];
        if (__float_as_int(r.w) < 0) continue;
        const float dx = qx - r.x, dy = qy - r.y, dz = qz - r.z;
        float d = dx * dx;
        d = d + dy * dy;
        d = d + dz * dz;
        const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | j;
        // (positions ascend: an equal distance never replaces the key)
        if (TIES) tie = k < key ? false : (tie || (unsigned int)(k >> 32) == (unsigned int)(key >> 32));
        key = k < key ? k : key;
    }
    sk[wave][lane] = key;
    if (TIES) st[wave][lane] = tie ? 1 : 0;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < SMALL_WAVES; ++w) {
        const unsigned long long k = sk[w][lane];
        key = k < key ? k : key;
    }
    const bool finite_q = __float_as_int(qv.w) >= 0;
    if (!finite_q) key = ~0ull;  // (k_pack's preset for a non-finite query: nothing found)
    const bool ok = finite_q && !key_none(key);
    if (TIES) {
        bool tied = false;
#pragma unroll
        for (int w = 0; w < SMALL_WAVES; ++w) {
            const unsigned long long k = sk[w][lane];
            tied = tied || ((unsigned int)(k >> 32) == (unsigned int)(key >> 32) && (k != key || st[w][lane] != 0));
        }
        tied = tied && ok && i < nq;
        const unsigned long long m = __ballot(tied);
        if (lane == 0) tie_blocks[blockIdx.x] = (unsigned int)__popcll(m);
        if (i < nq) tie_q[i] = tied ? 1 : 0;
    }
    if (i >= nq) return;
    out_packed[i] = key;
    if (idx) idx[i] = ok ? (int32_t)(unsigned int)(key & 0xffffffffull) : -1;
    if (d2) d2[i] = ok ? __uint_as_float((unsigned int)(key >> 32)) : __builtin_inff();
}

int launch_small_nn1(hipStream_t s, const void* raw_q, size_t nq, size_t stride, const float4* refs, size_t n, float4* q_packed,
                     unsigned long long* out_packed, int32_t* idx, float* d2, unsigned int* tie_blocks, unsigned char* tie_q) {
    if (nq == 0) return PCC_OK;
    const dim3 wg((unsigned int)((nq + 63) / 64)), th(SMALL_WAVES * 64);
    if (tie_blocks)
        hipLaunchKernelGGL(k_small_nn1<true>, wg, th, 0, s, static_cast<const char*>(raw_q), (unsigned int)nq, stride, refs, (unsigned int)n,
                           q_packed, out_packed, idx, d2, tie_blocks, tie_q);
    else
        hipLaunchKernelGGL(k_small_nn1<false>, wg, th, 0, s, static_cast<const char*>(raw_q), (unsigned int)nq, stride, refs, (unsigned int)n,
                           q_packed, out_packed, idx, d2, tie_blocks, tie_q);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

}  // namespace pcc
