// ubench_scatter.hip -- what a random permutation of n results costs on this chip, by form (make ubench):
//   A  out[perm[t]] = key[t]                 one 8-byte scattered store per element (what k_grid_nn1_flat2 does at its end)
//   B  idx[perm[t]] = lo(key[t]); d2[perm[t]] = hi(key[t])     two 4-byte scattered stores (a fused permute + unpack)
//   C  key[t] = out[perm[t]]                 the 8-byte gather
//   D  A with non-temporal stores
// perm is a random permutation (a bijective hash of t), read coalesced like the search kernel's order[].
// Second part, the GRANULARITY of a scattered 16-byte read (what the query gather q[order[t]] pays per element): a table of n 128-byte
// lines (8 float4 each, 1.28 GB at n = 10M: five times the Infinity Cache), every line visited once in random order:
//   L1  one float4 of the line                       L2s  two float4 of the same 64-byte half
//   L2d two float4, one from each 64-byte half       L2x  one float4 each of TWO different random lines
// With 128-byte fills L1 = L2s = L2d and L2x costs twice; with 64-byte fills L2d would cost what L2x does.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int FORM>
__global__ void __launch_bounds__(256) k_perm(const unsigned int* __restrict__ perm, const unsigned long long* __restrict__ src,
                                              unsigned long long* __restrict__ dst, int* __restrict__ idx, float* __restrict__ d2, unsigned int n) {
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const unsigned int p = perm[t];
        if (FORM == 0) dst[p] = src[t];
        if (FORM == 1) { const unsigned long long k = src[t]; idx[p] = (int)(unsigned int)k; d2[p] = __uint_as_float((unsigned int)(k >> 32)); }
        if (FORM == 2) dst[t] = src[p];
        if (FORM == 3) __builtin_nontemporal_store(src[t], &dst[p]);
    }
}

template <int FORM>
__global__ void __launch_bounds__(256) k_lines(const unsigned int* __restrict__ perm, const float4* __restrict__ table, float* __restrict__ dst,
                                               unsigned int n) {
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const unsigned int l = perm[t];
        float4 a = table[(size_t)l * 8u], b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FORM == 1) b = table[(size_t)l * 8u + 1u];
        if (FORM == 2) b = table[(size_t)l * 8u + 4u];
        if (FORM == 3) b = table[(size_t)perm[t + n / 2 < n ? t + n / 2 : t + n / 2 - n] * 8u + 4u];
        dst[t] = (a.x + a.w) + (b.y + b.z);
    }
}

int main(int argc, char** argv) {
    const unsigned int n = argc > 1 ? (unsigned int)atof(argv[1]) : 10000000u;
    std::vector<unsigned int> h(n);
    std::iota(h.begin(), h.end(), 0u);
    std::mt19937 rng(12345);
    std::shuffle(h.begin(), h.end(), rng);
    unsigned int* perm; unsigned long long *src, *dst; int* idx; float* d2;
    CK(hipMalloc(&perm, n * 4ull)); CK(hipMalloc(&src, n * 8ull)); CK(hipMalloc(&dst, n * 8ull)); CK(hipMalloc(&idx, n * 4ull)); CK(hipMalloc(&d2, n * 4ull));
    CK(hipMemcpy(perm, h.data(), n * 4ull, hipMemcpyHostToDevice));
    CK(hipMemset(src, 1, n * 8ull));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"A 8-byte scatter", "B 2 x 4-byte scatter", "C 8-byte gather", "D 8-byte scatter, non-temporal"};
    for (int grid : {0, 2048, 8192}) {
        const unsigned int g = grid ? (unsigned int)grid : (n + 255) / 256;
        for (int form = 0; form < 4; ++form) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                if (form == 0) hipLaunchKernelGGL(k_perm<0>, dim3(g), dim3(256), 0, 0, perm, src, dst, idx, d2, n);
                if (form == 1) hipLaunchKernelGGL(k_perm<1>, dim3(g), dim3(256), 0, 0, perm, src, dst, idx, d2, n);
                if (form == 2) hipLaunchKernelGGL(k_perm<2>, dim3(g), dim3(256), 0, 0, perm, src, dst, idx, d2, n);
                if (form == 3) hipLaunchKernelGGL(k_perm<3>, dim3(g), dim3(256), 0, 0, perm, src, dst, idx, d2, n);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("n=%u grid=%u  %-32s %8.1f us\n", n, g, names[form], best * 1e3f);
        }
    }
    // ---- granularity of a scattered 16-byte read ----
    float4* table; float* out4;
    CK(hipMalloc(&table, (size_t)n * 128ull)); CK(hipMalloc(&out4, n * 4ull));
    CK(hipMemset(table, 0, (size_t)n * 128ull));
    const char* lnames[4] = {"L1  one float4 per line", "L2s two float4, same 64-B half", "L2d two float4, both halves", "L2x two float4, two lines"};
    const unsigned int g = (n + 255) / 256;
    for (int form = 0; form < 4; ++form) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            if (form == 0) hipLaunchKernelGGL(k_lines<0>, dim3(g), dim3(256), 0, 0, perm, table, out4, n);
            if (form == 1) hipLaunchKernelGGL(k_lines<1>, dim3(g), dim3(256), 0, 0, perm, table, out4, n);
            if (form == 2) hipLaunchKernelGGL(k_lines<2>, dim3(g), dim3(256), 0, 0, perm, table, out4, n);
            if (form == 3) hipLaunchKernelGGL(k_lines<3>, dim3(g), dim3(256), 0, 0, perm, table, out4, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("n=%u lines of 128 B (table %.2f GB)  %-32s %8.1f us   %6.2f TB/s if 128 B per line touched, %6.2f if 64 B per half touched\n", n,
               n * 128.0 / 1e9, lnames[form], best * 1e3f, (form == 3 ? 2.0 : 1.0) * n * 128.0 / (best * 1e-3) / 1e12,
               (form >= 2 ? 2.0 : 1.0) * n * 64.0 / (best * 1e-3) / 1e12);
    }
    return 0;
}
