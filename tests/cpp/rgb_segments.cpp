// rgb_segments.cpp -- test helper: color_growing_segmentation (include/pcc/region_growing_rgb.hpp: pcl::RegionGrowingRGB over
// the GPU k-NN rows) on a PLY; prints the number of colour segments, then "label" per point of the NaN-stripped cloud
// (cluster in PCL's order, -1 = dropped).  tests/test_rgb_gpu.py compares it with the oracle's restatement.
#include <cstdio>
#include <cstdlib>
#include "ply_io.hpp"
#include "pcc/region_growing_rgb.hpp"
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    pcc::PointCloud<pcc::PointXYZRGB>::Ptr c(new pcc::PointCloud<pcc::PointXYZRGB>);
    if (pcc::io::loadPLYFile(argv[1], *c) == -1) { std::printf("LOAD_FAILED\n"); return 1; }
    std::vector<int> idx;
    pcc::io::removeNaNFromPointCloud(*c, idx);
    pcc::RegionGrowingRGB<pcc::PointXYZRGB> reg;
    reg.setInputCloud(c);
    reg.setDistanceThreshold(argc > 2 ? (float)std::atof(argv[2]) : 10.f);
    reg.setPointColorThreshold(argc > 3 ? (float)std::atof(argv[3]) : 6.f);
    reg.setRegionColorThreshold(argc > 4 ? (float)std::atof(argv[4]) : 5.f);
    reg.setMinClusterSize(argc > 5 ? std::atoi(argv[5]) : 200);
    std::vector<pcc::PointIndices> clusters;
    reg.extract(clusters);
    std::vector<int> label(c->size(), -1);
    for (size_t k = 0; k < clusters.size(); ++k)
        for (int i : clusters[k].indices) label[(size_t)i] = (int)k;
    std::printf("%zu %zu\n", clusters.size(), c->size());
    for (int v : label) std::printf("%d\n", v);
    // and through the reference-shaped function (defaults of src/segmentation.cpp:161-216)
    std::printf("segments %zu\n", pcc::color_growing_segmentation<pcc::PointXYZRGB>(c).size());
    return 0;
}
