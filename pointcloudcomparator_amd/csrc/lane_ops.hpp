// lane_ops.hpp -- cross-lane exchanges of a wave64 without the LDS crossbar (gfx950).
// __shfl_xor compiles to ds_bpermute_b32: an LDS-pipeline instruction with ~100 clk of latency.  A
// bitonic network is a chain of dependent exchanges, so the k-NN merge spent its time waiting on them.
// xor masks 1, 2, 8 are one DPP mov (quad_perm / row_ror:8), 4 is two row rotations and a select, 16 and 32
// are the gfx950 row / half swaps (v_permlane16_swap, v_permlane32_swap) and a select.  Semantics checked
// on the device for every lane.
#pragma once
#include <hip/hip_runtime.h>

namespace pcc {

// Lanes of ONE wave exchanging data through LDS: the DS operations of a wave execute in issue order, so no hardware
// barrier is needed -- but the compiler must keep that order and must not forward a lane's own store to its later load
// of the same address when another lane's store or atomic may have landed in between (it reasons per thread:
// __builtin_amdgcn_wave_barrier alone is "no memory effect" to it, and a load was seen sunk into the branch of the
// lane's own atomic).  Release / acquire fences at wavefront scope emit no instruction and say exactly that.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value of lane (lane ^ m); m must fold to a constant in {1, 2, 4, 8, 16, 32}
__device__ __forceinline__ unsigned int xor_lane_u32(unsigned int x, int m, unsigned int lane) {
    switch (m) {
        case 1: return (unsigned int)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
        case 2: return (unsigned int)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
        case 4: {
            const unsigned int a = (unsigned int)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x124, 0xf, 0xf, true);  // row_ror:4
            const unsigned int b = (unsigned int)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x12C, 0xf, 0xf, true);  // row_ror:12
            return (lane & 4u) ? a : b;
        }
        case 8: return (unsigned int)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xf, 0xf, true);  // row_ror:8
        case 16: {
            const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
            return (lane & 16u) ? r[0] : r[1];
        }
        default: {
            const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
            return (lane & 32u) ? r[0] : r[1];
        }
    }
}
__device__ __forceinline__ unsigned long long xor_lane_u64(unsigned long long v, int m, unsigned int lane) {
    const unsigned int lo = xor_lane_u32((unsigned int)v, m, lane);
    const unsigned int hi = xor_lane_u32((unsigned int)(v >> 32), m, lane);
    return ((unsigned long long)hi << 32) | lo;
}
// value of lane (63 - lane)
__device__ __forceinline__ unsigned int reverse_lanes_u32(unsigned int x, unsigned int lane) {
    unsigned int m = (unsigned int)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, true);  // row_mirror
    m = xor_lane_u32(m, 16, lane);
    return xor_lane_u32(m, 32, lane);
}
__device__ __forceinline__ unsigned long long reverse_lanes_u64(unsigned long long v, unsigned int lane) {
    const unsigned int lo = reverse_lanes_u32((unsigned int)v, lane);
    const unsigned int hi = reverse_lanes_u32((unsigned int)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

}  // namespace pcc

namespace pcc {

// inclusive scans over the 64 lanes (DPP row shifts inside the rows of 16, row_bcast:15 / :31 across them)
#define PCC_DPP_STEP(OP, X, CTRL, ROWMASK) \
    X = OP(X, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)X, CTRL, ROWMASK, 0xf, false))
__device__ __forceinline__ unsigned int lane_add_(unsigned int a, unsigned int b) { return a + b; }
__device__ __forceinline__ unsigned int lane_max_(unsigned int a, unsigned int b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned int wave_incl_scan_add(unsigned int x) {
    PCC_DPP_STEP(lane_add_, x, 0x111, 0xf);  // row_shr:1
    PCC_DPP_STEP(lane_add_, x, 0x112, 0xf);  // row_shr:2
    PCC_DPP_STEP(lane_add_, x, 0x114, 0xf);  // row_shr:4
    PCC_DPP_STEP(lane_add_, x, 0x118, 0xf);  // row_shr:8
    PCC_DPP_STEP(lane_add_, x, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    PCC_DPP_STEP(lane_add_, x, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3
    return x;
}
// identity 0: for values >= 0 only
__device__ __forceinline__ unsigned int wave_incl_scan_max(unsigned int x) {
    PCC_DPP_STEP(lane_max_, x, 0x111, 0xf);
    PCC_DPP_STEP(lane_max_, x, 0x112, 0xf);
    PCC_DPP_STEP(lane_max_, x, 0x114, 0xf);
    PCC_DPP_STEP(lane_max_, x, 0x118, 0xf);
    PCC_DPP_STEP(lane_max_, x, 0x142, 0xa);
    PCC_DPP_STEP(lane_max_, x, 0x143, 0xc);
    return x;
}
#undef PCC_DPP_STEP

}  // namespace pcc
