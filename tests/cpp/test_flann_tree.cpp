// CPU self-test of flann_tree.hpp (the PCC_TIES_FLANN tree of the product): host build (threads 1 and 8) and the
// iterative walk, on clouds with many exact ties.  Usage: test_flann_tree <rule> <n> <seed> reads nothing, prints the
// indices of nq queries one per line so that the Python side can compare them with the oracle's recursive kd-tree
// (tests/test_host_cpu.py).  Also run under -fsanitize=address,undefined by `make asan`.
#include "flann_tree.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <points.bin> <queries.bin> <rule> [threads]\n", argv[0]); return 2; }
    const int rule = atoi(argv[3]);
    const unsigned int threads = argc > 4 ? (unsigned int)atoi(argv[4]) : 1u;
    auto slurp = [](const char* path, std::vector<float>& v) {
        FILE* f = fopen(path, "rb");
        if (!f) { perror(path); exit(2); }
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        v.resize((size_t)sz / 4);
        if (fread(v.data(), 4, v.size(), f) != v.size()) { fprintf(stderr, "short read\n"); exit(2); }
        fclose(f);
    };
    std::vector<float> pts, qs;
    slurp(argv[1], pts);  // n x 3 floats
    slurp(argv[2], qs);
    const size_t n = pts.size() / 3, nq = qs.size() / 3;
    std::vector<float> packed(n * 4);
    for (size_t i = 0; i < n; ++i) {
        const float* p = &pts[i * 3];
        const bool ok = std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
        packed[i * 4 + 0] = ok ? p[0] : 0.f;
        packed[i * 4 + 1] = ok ? p[1] : 0.f;
        packed[i * 4 + 2] = ok ? p[2] : 0.f;
        const int32_t w = ok ? (int32_t)i : -1;
        memcpy(&packed[i * 4 + 3], &w, 4);
    }
    pcc::FlannTree t;
    t.build(packed.data(), n, rule, threads);
    printf("# n_valid %zu nodes %zu depth %d\n", t.n_valid, t.nodes.size(), t.depth);
    size_t full_walks = 0;
    for (size_t j = 0; j < nq; ++j) {
        float d2 = 0.f;
        const int32_t idx = t.nearest(&qs[j * 3], &d2);
        uint32_t bits;
        memcpy(&bits, &d2, 4);
        // the short walk for a known minimum distance must name the same reference (or hand over to the full walk)
        bool full = false;
        const int32_t tied = t.nearest_tied(&qs[j * 3], d2, &full);
        full_walks += full ? 1 : 0;
        printf("%d %u %d\n", idx, bits, tied);
    }
    printf("# short walks that handed over to the full walk: %zu of %zu\n", full_walks, nq);
    return 0;
}
