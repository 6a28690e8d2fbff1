// test_lane_ops.hip -- device check of csrc/lane_ops.hpp: every DPP / permlane exchange against its definition,
// for every lane.  Exit code 0 = pass.  Built by `make hosttest`, run by tests/test_host_mirror_gpu.py.
#include <cstdio>
#include "lane_ops.hpp"
using namespace pcc;

__global__ void k(unsigned int* o) {
    const unsigned int lane = threadIdx.x;
    const unsigned int v = lane * 3 + 7;
    const unsigned long long w = ((unsigned long long)(lane * 5 + 1) << 32) | (lane * 11 + 2);
    o[0 * 64 + lane] = xor_lane_u32(v, 1, lane);
    o[1 * 64 + lane] = xor_lane_u32(v, 2, lane);
    o[2 * 64 + lane] = xor_lane_u32(v, 4, lane);
    o[3 * 64 + lane] = xor_lane_u32(v, 8, lane);
    o[4 * 64 + lane] = xor_lane_u32(v, 16, lane);
    o[5 * 64 + lane] = xor_lane_u32(v, 32, lane);
    o[6 * 64 + lane] = reverse_lanes_u32(v, lane);
    const unsigned long long x = xor_lane_u64(w, 16, lane), r = reverse_lanes_u64(w, lane);
    o[7 * 64 + lane] = (unsigned int)(x >> 32);
    o[8 * 64 + lane] = (unsigned int)x;
    o[9 * 64 + lane] = (unsigned int)(r >> 32);
    o[10 * 64 + lane] = (unsigned int)r;
    o[11 * 64 + lane] = wave_incl_scan_add((lane * 37u + 11u) % 23u);
    o[12 * 64 + lane] = wave_incl_scan_max((lane * 29u + 5u) % 31u);
}

int main() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::printf("no HIP device: skipped\n"); return 77; }
    unsigned int* d = nullptr;
    if (hipMalloc(&d, 13 * 64 * 4) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned int h[13 * 64];
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const int ms[6] = {1, 2, 4, 8, 16, 32};
    int bad = 0;
    unsigned int s = 0, m = 0;
    for (int l = 0; l < 64; ++l) {
        for (int t = 0; t < 6; ++t) bad += h[t * 64 + l] != (unsigned)((l ^ ms[t]) * 3 + 7);
        bad += h[6 * 64 + l] != (unsigned)((63 - l) * 3 + 7);
        bad += h[7 * 64 + l] != (unsigned)((l ^ 16) * 5 + 1) || h[8 * 64 + l] != (unsigned)((l ^ 16) * 11 + 2);
        bad += h[9 * 64 + l] != (unsigned)((63 - l) * 5 + 1) || h[10 * 64 + l] != (unsigned)((63 - l) * 11 + 2);
        s += ((unsigned)l * 37u + 11u) % 23u;
        const unsigned int x = ((unsigned)l * 29u + 5u) % 31u;
        m = x > m ? x : m;
        bad += h[11 * 64 + l] != s;
        bad += h[12 * 64 + l] != m;
    }
    std::printf("lane ops: %d mismatches\n", bad);
    return bad != 0;
}
