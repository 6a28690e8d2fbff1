// test_libm.cpp -- csrc/libm_f32.hpp (glibc 2.35's sinf / cosf / atan2f restated for the device) against the HOST's libm,
// bit for bit: what the oracle's pcl::eigen33 restatement calls (oracle/pcc_oracle.c, std::atan2 / cos / sin on floats as
// PCL does, reference src/segmentation.cpp:232-241) must be what the device evaluates.  CPU only.
//   sinf, cosf : every float of [2^-13, 1.2] (theta = atan2f(..) / 3 never leaves [0, pi / 3]; below 2^-12 both return at once),
//                every 64th float below, 40M arguments drawn over [0, 120) for the other quadrants of the reduction
//   atanf      : every 64th float of the whole line (both signs, NaN and Inf included) and three whole binades
//   atan2f     : 20M pairs, half of them with exponents within 4 of each other, plus the special cases
// usage: test_libm [quick]   prints "libm ok" and the counts
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <initializer_list>
#include <cstring>
#include "libm_f32.hpp"

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float flt(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static bool same(float a, float b) { return bits(a) == bits(b) || (a != a && b != b); }
static uint64_t g_s = 88172645463325252ull;
static uint64_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }

int main(int argc, char** argv) {
    const bool quick = argc > 1;
    unsigned long bad = 0, n_sc = 0, n_at = 0, n_at2 = 0;
    auto sc = [&](float f) {
        ++n_sc;
        if (!same(sinf(f), pcc::lm_sinf(f))) { if (bad++ < 5) printf("sinf(%a): libm %a restated %a\n", f, sinf(f), pcc::lm_sinf(f)); }
        if (!same(cosf(f), pcc::lm_cosf(f))) { if (bad++ < 5) printf("cosf(%a): libm %a restated %a\n", f, cosf(f), pcc::lm_cosf(f)); }
    };
    const uint32_t lo = bits(0x1p-13f), hi = bits(1.2f);
    for (uint32_t u = 0; u < lo; u += 64) sc(flt(u));
    for (uint32_t u = lo; u <= hi; u += quick ? 16 : 1) sc(flt(u));
    // (beyond pi / 4 + the callers' range the FMA build and the baseline build of glibc differ for one argument in two
    // million: the restatement follows the FMA build, so the sweep is only meaningful on a CPU whose libm selects it)
    if (__builtin_cpu_supports("fma"))
        for (long i = 0; i < (quick ? 2000000L : 40000000L); ++i) {
            const float f = (float)((double)(rnd() >> 11) * (1.0 / 9007199254740992.0) * 119.99);
            sc(f);
            sc(-f);
        }
    auto at = [&](float f) {
        ++n_at;
        if (!same(atanf(f), pcc::lm_atanf(f))) { if (bad++ < 5) printf("atanf(%a): libm %a restated %a\n", f, atanf(f), pcc::lm_atanf(f)); }
    };
    for (uint64_t u = 0; u <= 0xffffffffull; u += quick ? 1024 : 64) at(flt((uint32_t)u));
    for (uint32_t e : {0x3e800000u, 0x3f000000u, 0x3f800000u})  // [0.25, 2): the four reduction intervals meet here
        for (uint32_t m = 0; m < (1u << 23); m += quick ? 16 : 1) at(flt(e + m));
    auto at2 = [&](float y, float x) {
        ++n_at2;
        if (!same(atan2f(y, x), pcc::lm_atan2f(y, x))) { if (bad++ < 5) printf("atan2f(%a, %a): libm %a restated %a\n", y, x, atan2f(y, x), pcc::lm_atan2f(y, x)); }
    };
    const float sp[] = {0.f, -0.f, 1.f, -1.f, INFINITY, -INFINITY, NAN, 1e-45f, -1e-45f, 3.4e38f, -3.4e38f, 1e-30f, 1e30f, 0.5f, 2.f};
    for (float y : sp) for (float x : sp) at2(y, x);
    for (long i = 0; i < (quick ? 1000000L : 20000000L); ++i) {
        const uint64_t r = rnd();
        float y = flt((uint32_t)r), x = flt((uint32_t)(r >> 32));
        if (i & 1) {  // exponents close together: the quotient lands in the reduction intervals
            const int ey = (int)((bits(y) >> 23) & 0xff);
            int ex = ey + (int)((r >> 20) % 9) - 4;
            ex = ex < 1 ? 1 : (ex > 254 ? 254 : ex);
            x = flt((bits(x) & 0x807fffffu) | ((uint32_t)ex << 23));
        }
        at2(y, x);
        at2(fabsf(y), x);  // (the callers' half plane: y = sqrtf(-q) >= 0)
    }
    printf("sinf/cosf %lu arguments, atanf %lu, atan2f %lu pairs: %lu mismatches\n", n_sc, n_at, n_at2, bad);
    if (bad) return 1;
    printf("libm ok\n");
    return 0;
}
