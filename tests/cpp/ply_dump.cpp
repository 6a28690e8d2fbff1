// ply_dump.cpp -- test helper: load a PLY with pointcloudcomparator_amd/host/ply_io.hpp, strip NaNs,
// print "n_loaded n_finite" and the first points as "x y z r g b" (full float precision).  No GPU.
#include <cstdio>
#include "ply_io.hpp"
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    pcc::PointCloud<pcc::PointXYZRGB> c;
    if (pcc::io::loadPLYFile(argv[1], c) == -1) { std::printf("LOAD_FAILED\n"); return 1; }
    size_t n = c.size();
    std::vector<int> idx;
    pcc::io::removeNaNFromPointCloud(c, idx);
    std::printf("%zu %zu\n", n, c.size());
    for (size_t i = 0; i < c.size() && i < 8; ++i)
        std::printf("%.9g %.9g %.9g %u %u %u\n", c[i].x, c[i].y, c[i].z, c[i].r, c[i].g, c[i].b);
    if (argc > 2) return pcc::io::savePLYFileBinary(argv[2], c) == 0 ? 0 : 3;
    return 0;
}
