// asan_driver.cpp -- `make asan`: the host-side code of the product (flann_tree.hpp, rigid_solve.hpp, plane_fit.hpp,
// host/ply_io.hpp) and the oracle (oracle/pcc_oracle.c) under -fsanitize=address,undefined with a CPU-only driver: random
// and degenerate clouds through every oracle entry, the PCC_TIES_FLANN tree (forked build, three split rules, deep and
// empty trees), degenerate ICP sums and covariances, and the PLY reader on valid, truncated and corrupt files.
// CPU build only -- sanitizers never run on the GPU box.  Prints "asan driver ok".
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <vector>
#include "flann_tree.hpp"
#include "plane_fit.hpp"
#include "rigid_solve.hpp"
#include "ply_io.hpp"
extern "C" {
#include "pcc_oracle.h"
}

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static float rnd() {
    g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((g_state >> 40) * (1.0 / (1 << 24)));
}

static std::vector<float> cloud(size_t n, int kind) {
    std::vector<float> p(n * 3);
    for (size_t i = 0; i < n; ++i) {
        float x = rnd(), y = rnd(), z = rnd();
        if (kind == 1) z = 0.5f;                                       // plane
        if (kind == 2) { x = std::floor(x * 6) * 0.25f; y = std::floor(y * 6) * 0.25f; z = std::floor(z * 6) * 0.25f; }  // lattice: ties
        if (kind == 3) { x = y = z = 0.25f; }                          // one pile
        if (kind == 4) { x = std::pow(x, 8.f) * 1e4f; y = std::pow(y, 8.f) * 1e-3f; }  // lopsided: deep trees
        p[i * 3] = x; p[i * 3 + 1] = y; p[i * 3 + 2] = z;
    }
    if (kind == 5 && n > 20) { p[3] = NAN; p[16] = INFINITY; p[31] = -INFINITY; }
    return p;
}

static int exercise_oracle(const std::vector<float>& a, const std::vector<float>& q) {
    const size_t m = a.size() / 3, n = q.size() / 3;
    std::vector<int32_t> idx(n * 8), cnt(n);
    std::vector<float> d2(n * 8);
    orc_nn1_exhaustive(a.data(), m, 12, q.data(), n, 12, idx.data(), d2.data());
    orc_knn_exhaustive(a.data(), m, 12, q.data(), n, 12, 8, idx.data(), d2.data());
    orc_radius_count_exhaustive(a.data(), m, 12, q.data(), n, 12, 0.01f, cnt.data());
    for (int rule = 0; rule < 3; ++rule) {
        orc_set_split_rule(rule);
        orc_kdtree* t = orc_kdtree_build(a.data(), m, 12);
        if (!t) continue;
        std::vector<int32_t> ti(n);
        std::vector<float> td(n);
        orc_kdtree_nn1_batch(t, q.data(), n, 12, ti.data(), td.data());
        orc_kdtree_nn1_batch_mt(t, q.data(), n, 12, ti.data(), td.data(), 3);
        std::vector<float> packed(m * 4);
        for (size_t i = 0; i < m; ++i) {
            const bool ok = std::isfinite(a[i * 3]) && std::isfinite(a[i * 3 + 1]) && std::isfinite(a[i * 3 + 2]);
            for (int d = 0; d < 3; ++d) packed[i * 4 + d] = ok ? a[i * 3 + d] : 0.f;
            const int32_t w = ok ? (int32_t)i : -1;
            memcpy(&packed[i * 4 + 3], &w, 4);
        }
        for (unsigned int threads : {1u, 5u}) {  // the product's tree against the oracle's, same rule
            pcc::FlannTree ft;
            ft.build(packed.data(), m, rule, threads);
            for (size_t j = 0; j < n; ++j) {
                if (!(std::isfinite(q[j * 3]) && std::isfinite(q[j * 3 + 1]) && std::isfinite(q[j * 3 + 2]))) continue;
                float fd = 0.f;
                const int32_t fi = ft.nearest(&q[j * 3], &fd);
                REQUIRE(fi == ti[j] && memcmp(&fd, &td[j], 4) == 0);
                REQUIRE(ft.nearest_tied(&q[j * 3], fd) == fi);  // the short walk for a known minimum distance (what the device runs)
            }
        }
        int32_t ki[16];
        float kd[16];
        for (size_t j = 0; j < n && j < 64; ++j) {
            if (!std::isfinite(q[j * 3])) continue;
            orc_kdtree_knn(t, &q[j * 3], 16, ki, kd);
            orc_kdtree_radius(t, &q[j * 3], 0.04f, 1, ki, kd, 16);
        }
        orc_kdtree_free(t);
    }
    orc_set_split_rule(0);
    std::vector<int32_t> labels(m), sizes(64);
    orc_euclidean_clusters(a.data(), m, 12, 0.05f, 2, 100000, labels.data(), sizes.data(), 64);
    {   // RegionGrowingRGB's restatement: few colours (many small segments, folds), rows from the oracle's own tree
        bool finite = true;
        for (float v : a) finite = finite && std::isfinite(v);
        if (finite && m <= 1000) {
            std::vector<uint8_t> rgb(m * 3);
            for (size_t i = 0; i < m * 3; ++i) rgb[i] = (uint8_t)(((i * 2654435761u) >> 28) * 16);
            std::vector<int32_t> cl(m);
            for (int mn : {1, 7, 200}) orc_region_growing_rgb(a.data(), m, 12, rgb.data(), nullptr, nullptr, 0, 10.f, 6.f, 5.f, mn, 1 << 30, 30, 100, cl.data());
        }
    }
    std::vector<float> md(m);
    std::vector<uint8_t> inl(m);
    double thr = 0;
    orc_sor(a.data(), m, 12, 8, 1.5, md.data(), inl.data(), &thr);
    const float vp[3] = {0, 0, 0};
    std::vector<float> nrm(m * 4);
    orc_normals(a.data(), m, 12, 10, vp, nrm.data());
    orc_normals_radius(a.data(), m, 12, 0.1, vp, nrm.data());
    std::vector<int32_t> inliers(m);
    float coeff[4];
    int its = 0;
    orc_sac_plane(a.data(), m, 12, 50, 0.02, 0.99, 1, inliers.data(), coeff, &its);
    std::vector<float> vox(m * 3);
    orc_voxel_grid(a.data(), m, 12, 0.1f, 0, vox.data(), 12);
    std::vector<int32_t> fw(n);
    orc_first_within(a.data(), m, 12, q.data(), n, 12, 0.05, fw.data());
    float T[16];
    double fit = 0;
    std::vector<int32_t> corr(n);
    std::vector<double> mse(5);
    orc_icp(q.data(), n, 12, a.data(), m, 12, 5, 0, T, &fit, corr.data(), mse.data());
    return 0;
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    // 1. oracle + the product's tree on random and degenerate clouds (sizes around the leaf size too)
    for (int kind = 0; kind <= 5; ++kind)
        for (size_t m : {(size_t)1, (size_t)14, (size_t)15, (size_t)16, (size_t)33, (size_t)700, (size_t)5000}) {
            const std::vector<float> a = cloud(m, kind);
            std::vector<float> q = cloud(200, kind == 3 ? 0 : kind);
            q[0] = NAN;  // a non-finite query: idx -1, never a crash
            if (exercise_oracle(a, q)) return 1;
        }
    {   // an all-invalid cloud: no tree
        std::vector<float> a(30, NAN);
        REQUIRE(orc_kdtree_build(a.data(), 10, 12) == nullptr);
        std::vector<float> packed(40, 0.f);
        for (int i = 0; i < 10; ++i) { const int32_t w = -1; memcpy(&packed[i * 4 + 3], &w, 4); }
        pcc::FlannTree ft;
        ft.build(packed.data(), 10, 0, 4);
        float d = 0.f, qq[3] = {0, 0, 0};
        REQUIRE(ft.n_valid == 0 && ft.nearest(qq, &d) == -1);
    }
    // 2. rigid_from_sums: too few pairs, collinear pairs, zeros, huge values
    {
        double sums[17] = {0};
        float T[16];
        REQUIRE(pcc::rigid_from_sums(sums, T) != 0);
        sums[16] = 2;
        REQUIRE(pcc::rigid_from_sums(sums, T) != 0);
        sums[16] = 5;  // five coincident pairs at the origin: any rotation is optimal, the result must be finite
        REQUIRE(pcc::rigid_from_sums(sums, T) == 0);
        for (int i = 0; i < 16; ++i) REQUIRE(std::isfinite(T[i]));
        for (int k = 0; k < 15; ++k) sums[k] = 1e300;
        (void)pcc::rigid_from_sums(sums, T);
        const double c[3] = {1e6, -1e6, 3};
        for (int k = 0; k < 15; ++k) sums[k] = 0.5 * k;
        REQUIRE(pcc::rigid_from_sums(sums, T, c) == 0);
    }
    // 3. plane fit: zero, rank-1 and NaN covariances
    {
        float n3[3], curv = 0.f;
        const float zero[9] = {0};
        pcc::plane_from_covariance(zero, n3, &curv);
        const float rank1[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0};
        pcc::plane_from_covariance(rank1, n3, &curv);
        float nan9[9];
        for (float& v : nan9) v = NAN;
        pcc::plane_from_covariance(nan9, n3, &curv);
        float acc[9] = {1, 2, 3, 4, 5, 6, 7, 8, 9}, cov[9];
        pcc::covariance_from_sums(acc, 3, cov);
    }
    // 4. the PLY reader: valid ascii / binary, then every truncation of them and a header that lies about its size
    {
        auto write = [&](const std::string& name, const std::string& bytes) {
            std::ofstream f((dir + "/" + name).c_str(), std::ios::binary);
            f.write(bytes.data(), (std::streamsize)bytes.size());
        };
        std::string ascii = "ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                            "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n0 0 0 1 2 3\n1 0.5 2 4 5 6\nnan 1 1 7 8 9\n";
        std::string bin = "ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty double x\nproperty double y\nproperty double z\n"
                          "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
        for (int v = 0; v < 2; ++v) {
            const double xyz[3] = {1.5 + v, -2.0, 3.25};
            bin.append(reinterpret_cast<const char*>(xyz), 24);
            bin.append("\x0a\x14\x1e", 3);
        }
        pcc::PointCloud<pcc::PointXYZRGB> c;
        write("a.ply", ascii);
        REQUIRE(pcc::io::loadPLYFile(dir + "/a.ply", c) == 0 && c.size() == 3);
        write("b.ply", bin);
        REQUIRE(pcc::io::loadPLYFile(dir + "/b.ply", c) == 0 && c.size() == 2 && c.points[1].x == 2.5f);
        for (const std::string* src : {&ascii, &bin})
            for (size_t cut = 0; cut < src->size(); cut += 3) {
                write("t.ply", src->substr(0, cut));
                (void)pcc::io::loadPLYFile(dir + "/t.ply", c);  // any outcome but a memory error
            }
        std::string liar = ascii;
        liar.replace(liar.find("vertex 3"), 8, "vertex 4000000000");
        write("l.ply", liar);
        REQUIRE(pcc::io::loadPLYFile(dir + "/l.ply", c) == -1);
        std::string junk = "ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty list uchar int vertex_indices\nproperty float x\nend_header\n\x01\x02";
        write("j.ply", junk);
        (void)pcc::io::loadPLYFile(dir + "/j.ply", c);
        REQUIRE(pcc::io::loadPLYFile(dir + "/does_not_exist.ply", c) == -1);
    }
    std::printf("asan driver ok\n");
    return 0;
}
