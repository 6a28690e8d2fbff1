// ubench_gather.hip -- what does a wave64 gather cost the CU?  Standalone (make ubench), run on the GPU box.
// The grid-search kernels issue one 16-byte global load per (lane, candidate); DESIGN.md 4.2 prices them at
// ~16 clk of the CU's vector-memory path per wave instruction.  This measures that price and how it moves
// with the number of active lanes, the access width and the locality of the addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 2048;

// MODE 0: 64 lanes, 16 B, every lane its own line inside a 4 KB window per wave (L1 resident)
// MODE 1:  8 lanes active, otherwise as 0
// MODE 2: 64 lanes, 16 B, lanes read 4 consecutive float4 (one 64 B segment per 4 lanes), L1 resident
// MODE 3: 64 lanes,  8 B, as 0
// MODE 4: 64 lanes, 16 B, pseudo-random lines over a 16 MB array (L2 / Infinity Cache)
// MODE 5: 16 lanes active, as 4
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const float4* __restrict__ data, unsigned int n_mask, float4* out) {
    const unsigned int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const bool active = MODE == 1 ? lane < 8 : (MODE == 5 ? lane < 16 : true);
    float4 acc = make_float4(0, 0, 0, 0);
    unsigned int idx = (wave * 2654435761u) & n_mask;
    if (active) {
        for (int i = 0; i < ITERS; ++i) {
            unsigned int a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE <= 1 || MODE == 3) a[u] = ((idx & ~255u) + ((lane * 4 + u * 67 + i * 13) & 255u)) & n_mask;  // 256 float4 = 4 KB window
                else if (MODE == 2) a[u] = ((idx & ~255u) + ((lane + u * 64 + i * 4) & 255u)) & n_mask;
                else a[u] = (idx + lane * 7919u + u * 104729u + (unsigned int)i * 1299709u) & n_mask;
            }
            if (MODE == 3) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float2 v = *reinterpret_cast<const float2*>(data + a[u]);
                    acc.x += v.x; acc.y += v.y;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 v = data[a[u]];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
            if (MODE >= 4) idx = idx * 1664525u + 1013904223u;
        }
    }
    if (acc.x == 12345.678f) out[wave] = acc;
}

template <int MODE>
static void run(const char* name, const float4* d, unsigned int mask, float4* out, int cus) {
    const int blocks = cus * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, d, mask, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, d, mask, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves_per_cu = 8.0 * 4.0;  // blocks per CU * waves per block
    const double instr_per_cu = waves_per_cu * ITERS * 4.0;
    printf("%-58s %8.3f ms  %6.2f ns per wave-load per CU (%5.1f clk at 2.1 GHz)\n", name, ms, ms * 1e6 / instr_per_cu,
           ms * 1e6 / instr_per_cu * 2.1);
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const unsigned int n = 1u << 20;  // 1M float4 = 16 MB
    float4* d; float4* out;
    CHECK(hipMalloc(&d, (size_t)n * sizeof(float4)));
    CHECK(hipMalloc(&out, (size_t)cus * 64 * sizeof(float4)));
    CHECK(hipMemset(d, 0, (size_t)n * sizeof(float4)));
    printf("%s, %d CUs\n", p.name, cus);
    run<0>("64 lanes x 16 B, own lines, 4 KB window (L1)", d, n - 1, out, cus);
    run<1>(" 8 lanes x 16 B, own lines, 4 KB window (L1)", d, n - 1, out, cus);
    run<2>("64 lanes x 16 B, 4 lanes per 64 B segment (L1)", d, n - 1, out, cus);
    run<3>("64 lanes x  8 B, own lines, 4 KB window (L1)", d, n - 1, out, cus);
    run<4>("64 lanes x 16 B, random over 16 MB (L2 / MALL)", d, n - 1, out, cus);
    run<5>("16 lanes x 16 B, random over 16 MB (L2 / MALL)", d, n - 1, out, cus);
    return 0;
}
