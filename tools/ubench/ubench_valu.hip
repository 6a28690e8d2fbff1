// ubench_valu.hip -- issue-rate microbenchmark for the instructions of the NN inner loop.
// Not part of libpcc_nn; built as a standalone executable (make ubench) and run on the
// GPU box to pin the VALU ceiling DESIGN.md prices the exhaustive kernel against:
// is a wave64 v_mul_f32 2 cycles, and does v_pk_mul_f32 retire two results in the
// same slot or take twice as long?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 4096;

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * 0.5f, b5 = a5 * 0.5f, b6 = a6 * 0.5f, b7 = a7 * 0.5f;
    float c = 1.0000001f;
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {  // 16 independent v_mul_f32
            asm volatile(
                "v_mul_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n"
                "v_mul_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_mul_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n"
                "v_mul_f32 %8, %8, %16\n v_mul_f32 %9, %9, %16\n v_mul_f32 %10, %10, %16\n v_mul_f32 %11, %11, %16\n"
                "v_mul_f32 %12, %12, %16\n v_mul_f32 %13, %13, %16\n v_mul_f32 %14, %14, %16\n v_mul_f32 %15, %15, %16\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                  "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                : "v"(c));
        } else if (MODE == 1) {  // 8 independent v_pk_mul_f32 (16 results)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
            f2 cc = {c, c};
            asm volatile(
                "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                : "v"(cc));
            a0 = p0.x; b0 = p0.y; a1 = p1.x; b1 = p1.y; a2 = p2.x; b2 = p2.y; a3 = p3.x; b3 = p3.y;
            a4 = p4.x; b4 = p4.y; a5 = p5.x; b5 = p5.y; a6 = p6.x; b6 = p6.y; a7 = p7.x; b7 = p7.y;
        } else if (MODE == 2) {  // 16 v_min3_f32
            asm volatile(
                "v_min3_f32 %0, %0, %8, %16\n v_min3_f32 %1, %1, %9, %16\n v_min3_f32 %2, %2, %10, %16\n v_min3_f32 %3, %3, %11, %16\n"
                "v_min3_f32 %4, %4, %12, %16\n v_min3_f32 %5, %5, %13, %16\n v_min3_f32 %6, %6, %14, %16\n v_min3_f32 %7, %7, %15, %16\n"
                "v_min3_f32 %8, %8, %0, %16\n v_min3_f32 %9, %9, %1, %16\n v_min3_f32 %10, %10, %2, %16\n v_min3_f32 %11, %11, %3, %16\n"
                "v_min3_f32 %12, %12, %4, %16\n v_min3_f32 %13, %13, %5, %16\n v_min3_f32 %14, %14, %6, %16\n v_min3_f32 %15, %15, %7, %16\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                  "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                : "v"(c));
        } else if (MODE == 3) {  // the scalar inner-loop mix: 2 pairs = 6 sub, 6 mul, 4 add, 1 min3 (17 ops)
            asm volatile(
                "v_sub_f32 %8, %0, %16\n v_sub_f32 %9, %1, %16\n v_sub_f32 %10, %2, %16\n"
                "v_sub_f32 %11, %3, %16\n v_sub_f32 %12, %4, %16\n v_sub_f32 %13, %5, %16\n"
                "v_mul_f32 %8, %8, %8\n v_mul_f32 %9, %9, %9\n v_mul_f32 %10, %10, %10\n"
                "v_mul_f32 %11, %11, %11\n v_mul_f32 %12, %12, %12\n v_mul_f32 %13, %13, %13\n"
                "v_add_f32 %8, %8, %9\n v_add_f32 %11, %11, %12\n"
                "v_add_f32 %8, %8, %10\n v_add_f32 %11, %11, %13\n"
                "v_min3_f32 %6, %6, %8, %11\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                  "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                : "v"(c));
        } else if (MODE == 4) {  // packed mix: 2 pairs = 3 pk_add(sub), 3 pk_mul, 2 pk_add, 1 min3 (9 ops)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 qx = {a0, b0}, qy = {a1, b1}, qz = {a2, b2}, t0 = {a3, b3}, t1 = {a4, b4}, t2 = {a5, b5};
            f2 cc = {c, c};
            asm volatile(
                "v_pk_add_f32 %3, %0, %7 neg_lo:[0,1] neg_hi:[0,1]\n"
                "v_pk_add_f32 %4, %1, %7 neg_lo:[0,1] neg_hi:[0,1]\n"
                "v_pk_add_f32 %5, %2, %7 neg_lo:[0,1] neg_hi:[0,1]\n"
                "v_pk_mul_f32 %3, %3, %3\n v_pk_mul_f32 %4, %4, %4\n v_pk_mul_f32 %5, %5, %5\n"
                "v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %3, %3, %5\n"
                : "+v"(qx), "+v"(qy), "+v"(qz), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(a6)
                : "v"(cc));
            asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(t0.x), "v"(t0.y));
            a3 = t0.x; b3 = t0.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, double results_per_iter, double instrs_per_iter) {
    int ncu = 256;
    int grid = ncu * blocks_per_cu;
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * grid * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    const int reps = 10;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double t = ms * 1e-3 / reps;
    double lanes = (double)grid * 256;
    double res = lanes * ITERS * results_per_iter / t;
    double ins = (double)grid * 4 * ITERS * instrs_per_iter / t;  // wave-instructions / s
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    double cyc = 2.4e9 * (256.0 * 4) / ins;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f T results/s  %.2f G wave-instr/s  ~%.2f cyc/instr/SIMD@2.4GHz\n",
           name, blocks_per_cu, t * 1e3, res * 1e-12, ins * 1e-9, cyc);
    CHECK(hipFree(out));
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        if (w == 1) run<0>("v_mul_f32 x16", 1, 16, 16);
        if (w == 2) run<0>("v_mul_f32 x16", 2, 16, 16);
        if (w == 4) run<0>("v_mul_f32 x16", 4, 16, 16);
        if (w == 8) run<0>("v_mul_f32 x16", 8, 16, 16);
    }
    run<1>("v_pk_mul_f32 x16", 1, 32, 16);
    run<1>("v_pk_mul_f32 x16", 2, 32, 16);
    run<1>("v_pk_mul_f32 x16", 4, 32, 16);
    run<2>("v_min3_f32 x16", 2, 16, 16);
    run<2>("v_min3_f32 x16", 4, 16, 16);
    run<3>("scalar mix 17 ops / 2 pairs", 2, 2, 17);
    run<3>("scalar mix 17 ops / 2 pairs", 4, 2, 17);
    run<4>("packed mix 9 ops / 2 pairs", 2, 2, 9);
    run<4>("packed mix 9 ops / 2 pairs", 4, 2, 9);
    return 0;
}
