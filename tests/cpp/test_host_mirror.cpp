// test_host_mirror.cpp -- the C++ host mirror (include/pcc/*.hpp) exercised the way the
// reference's call sites use PCL: matchRIFTFeaturesKnn, performICP, SOR, EuclideanClusterExtraction,
// per-point KdTree calls.  Checks are self-consistency properties (the bit-level parity against the
// oracle lives in the Python GPU tests, through the same C-ABI).  Exit code 0 = pass, 77 = no GPU.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <chrono>
#include <hip/hip_runtime_api.h>
#include "pcc/comparator_nn.hpp"
#include "pcc/multi_device.hpp"

using namespace pcc;

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static float brute_d2(const PointXYZRGB& a, const PointXYZRGB& b) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    float d = dx * dx; d = d + dy * dy; d = d + dz * dz;
    return d;
}

int main() {
    int ndev = 0;
    if (pcc_device_count(&ndev) != PCC_OK || ndev == 0) { std::printf("no HIP device: skipped\n"); return 77; }
    std::mt19937 rng(42);
    std::uniform_real_distribution<float> U(0.f, 1.f);

    // clouds: three blobs of 400 points + 50 stragglers
    PointCloud<PointXYZRGB>::Ptr cloud(new PointCloud<PointXYZRGB>);
    const float centres[3][3] = {{0, 0, 0}, {2, 0, 0}, {0, 2, 1}};
    for (int b = 0; b < 3; ++b)
        for (int i = 0; i < 400; ++i) {
            PointXYZRGB p;
            p.x = centres[b][0] + 0.1f * U(rng); p.y = centres[b][1] + 0.1f * U(rng); p.z = centres[b][2] + 0.1f * U(rng);
            cloud->push_back(p);
        }
    for (int i = 0; i < 50; ++i) { PointXYZRGB p; p.x = 5 + 3 * U(rng); p.y = 5 + 3 * U(rng); p.z = 3 * U(rng); cloud->push_back(p); }

    // per-point KdTree calls (src/comparator.cpp:571-577 shape)
    search::KdTree<PointXYZRGB>::Ptr tree(new search::KdTree<PointXYZRGB>);
    tree->setInputCloud(cloud);
    std::vector<int> idx; std::vector<float> d2;
    REQUIRE(tree->nearestKSearch(cloud->points[7], 1, idx, d2) == 1 && idx[0] == 7 && d2[0] == 0.f);
    REQUIRE(tree->nearestKSearch(7, 5, idx, d2) == 5 && idx[0] == 7);
    for (int j = 1; j < 5; ++j) REQUIRE(d2[j] >= d2[j - 1] && d2[j] == brute_d2(cloud->points[7], cloud->points[idx[j]]));
    int nr = tree->radiusSearch(cloud->points[7], 0.05, idx, d2);
    REQUIRE(nr >= 1 && idx[0] == 7);
    int cnt = 0;
    for (auto& p : cloud->points) cnt += brute_d2(cloud->points[7], p) < (float)(0.05 * 0.05);
    REQUIRE(cnt == nr);
    REQUIRE(tree->nearestKSearch(cloud->points[0], 5000, idx, d2) == (int)cloud->size());  // k clamped

    // setInputCloud(cloud, indices) (pcl::search::Search): the tree holds the listed points only,
    // results name points of the whole cloud, nearestKSearch(int) takes cloud[indices[i]] as the query
    {
        std::shared_ptr<std::vector<int>> sel(new std::vector<int>);
        for (int i = 1; i < (int)cloud->size(); i += 3) sel->push_back(i);
        search::KdTree<PointXYZRGB> sub;
        sub.setInputCloud(cloud, sel);
        REQUIRE(sub.nearestKSearch(cloud->points[7], 1, idx, d2) == 1 && idx[0] == 7 && d2[0] == 0.f);  // 7 = 1 + 2*3 is listed
        REQUIRE(sub.nearestKSearch(cloud->points[8], 4, idx, d2) == 4);
        for (int j = 0; j < 4; ++j) {
            REQUIRE(idx[j] % 3 == 1 && d2[j] == brute_d2(cloud->points[8], cloud->points[idx[j]]));
            REQUIRE(j == 0 || d2[j] >= d2[j - 1]);
        }
        float best = 1e30f; int arg = -1;
        for (int i : *sel) { float d = brute_d2(cloud->points[8], cloud->points[i]); if (d < best) { best = d; arg = i; } }
        REQUIRE(idx[0] == arg && d2[0] == best);
        REQUIRE(sub.nearestKSearch(2, 1, idx, d2) == 1 && idx[0] == (*sel)[2] && d2[0] == 0.f);  // query = cloud[indices[2]]
        int nrs = sub.radiusSearch(cloud->points[7], 0.05, idx, d2);
        int want = 0;
        for (int i : *sel) want += brute_d2(cloud->points[7], cloud->points[i]) < (float)(0.05 * 0.05);
        REQUIRE(nrs == want && idx[0] == 7);
        for (int v : idx) REQUIRE(v % 3 == 1);
        REQUIRE(sub.nearestKSearch(cloud->points[0], 5000, idx, d2) == (int)sel->size());  // k clamped to the subset
        sub.setInputCloud(cloud);  // back to the whole cloud
        REQUIRE(sub.nearestKSearch(cloud->points[8], 1, idx, d2) == 1 && idx[0] == 8);
    }
    // pcl::PointXYZ: 16-byte stride
    {
        static_assert(sizeof(PointXYZ) == 16, "pcl::PointXYZ is 16 bytes");
        PointCloud<PointXYZ>::Ptr xyz(new PointCloud<PointXYZ>);
        for (auto& p : cloud->points) { PointXYZ v; v.x = p.x; v.y = p.y; v.z = p.z; xyz->push_back(v); }
        KdTreeFLANN<PointXYZ> t16;
        t16.setInputCloud(xyz);
        std::vector<int> i16; std::vector<float> d16;
        t16.nearestKSearchBatch(*xyz, i16, d16);
        for (size_t i = 0; i < xyz->size(); ++i) REQUIRE(d16[i] == 0.f);
        REQUIRE(t16.nearestKSearch(xyz->points[11], 3, idx, d2) == 3 && idx[0] == 11);
    }

    // one index per device, queries sharded -- with one GPU: two handles on device 0
    {
        std::vector<int> i1, i2; std::vector<float> d1, d2b;
        tree->nearestKSearchBatch(*cloud, i1, d1);
        ShardedKdTree<PointXYZRGB> sharded(std::vector<int>{0, 0, 0});
        sharded.setInputCloud(cloud);
        REQUIRE(sharded.shards() == 3 && sharded.handle(0) != sharded.handle(1));
        size_t nv = 0;
        REQUIRE(pcc_index_size(sharded.handle(2), &nv) == PCC_OK && nv == cloud->size());
        sharded.nearestKSearchBatch(*cloud, i2, d2b);
        REQUIRE(i1 == i2 && d1 == d2b);
        std::vector<int> k1, k2; std::vector<float> e1, e2;
        tree->nearestKSearchBatch(*cloud, 6, k1, e1);
        sharded.nearestKSearchBatch(*cloud, 6, k2, e2);
        REQUIRE(k1 == k2 && e1 == e2);
        // a clone keeps the indices of the original cloud, non-finite points included
        PointCloud<PointXYZRGB>::Ptr holes(new PointCloud<PointXYZRGB>(*cloud));
        holes->points[3].x = NAN; holes->points[700].z = INFINITY;
        pcc_index *a = nullptr, *b = nullptr;
        REQUIRE(pcc_index_create(holes->points.data(), holes->size(), sizeof(PointXYZRGB), 3, PCC_MEM_HOST, 0, PCC_ENGINE_AUTO, &a) == PCC_OK);
        REQUIRE(pcc_index_clone_to_device(a, 0, &b) == PCC_OK);
        REQUIRE(pcc_index_size(b, &nv) == PCC_OK && nv == holes->size() - 2);
        std::vector<int32_t> ia(cloud->size()), ib(cloud->size()); std::vector<float> da(cloud->size()), db(cloud->size());
        REQUIRE(pcc_nn1(a, cloud->points.data(), cloud->size(), sizeof(PointXYZRGB), PCC_MEM_HOST, ia.data(), da.data()) == PCC_OK);
        REQUIRE(pcc_nn1(b, cloud->points.data(), cloud->size(), sizeof(PointXYZRGB), PCC_MEM_HOST, ib.data(), db.data()) == PCC_OK);
        REQUIRE(ia == ib && da == db && ia[3] != 3 && ia[700] != 700 && ia[5] == 5);
        REQUIRE(pcc_index_clone_to_device(a, 99, &b) == PCC_ERR_INVALID);
        pcc_index_destroy(a); pcc_index_destroy(b);
        // replicas: two jobs, each on "its" device, objects created inside pick the device up
        int seen[2] = {-1, -1};
        onDevices(2, std::vector<int>{0}, [&](int k) {
            search::KdTree<PointXYZRGB> t;
            t.setInputCloud(cloud);
            std::vector<int> ii; std::vector<float> dd;
            seen[k] = t.nearestKSearch(cloud->points[k], 1, ii, dd) == 1 && ii[0] == k ? t.device() : -2;
        });
        REQUIRE(seen[0] == 0 && seen[1] == 0);
        size_t s0, c0;
        shardRange(10, 3, 4, s0, c0);
        REQUIRE(s0 == 8 && c0 == 2);
    }

    // concurrent multi-device set-up: four handles from one upload, every peer copy issued before any wait
    // (pcc_index_clone_to_devices) against four clones made one after the other; options and tie order carry over;
    // a shard searched through device pointers (nothing crosses PCIe)
    {
        const size_t big = 2000000;
        PointCloud<PointXYZRGB>::Ptr bc(new PointCloud<PointXYZRGB>);
        bc->points.resize(big);
        uint64_t st = 0x1234567ull;
        auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) * (1.0 / (1 << 24))); };
        for (size_t i = 0; i < big; ++i) { bc->points[i].x = rnd() * 10.f; bc->points[i].y = rnd() * 8.f; bc->points[i].z = rnd() * 3.f; }
        pcc_index* src = nullptr;
        REQUIRE(pcc_index_create(bc->points.data(), big, sizeof(PointXYZRGB), 3, PCC_MEM_HOST, 0, PCC_ENGINE_GRID, &src) == PCC_OK);
        REQUIRE(pcc_index_set_tie_order(src, PCC_TIES_FLANN) == PCC_OK);
        REQUIRE(pcc_index_set_option(src, PCC_OPT_FAR_MODE, 1) == PCC_OK);
        const int devs[4] = {0, 0, 0, 0};
        pcc_index* seq[4] = {nullptr, nullptr, nullptr, nullptr};
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 4; ++k) REQUIRE(pcc_index_clone_to_device(src, devs[k], &seq[k]) == PCC_OK);
        auto t1 = std::chrono::steady_clock::now();
        pcc_index* par[4] = {nullptr, nullptr, nullptr, nullptr};
        REQUIRE(pcc_index_clone_to_devices(src, devs, 4, par) == PCC_OK);
        auto t2 = std::chrono::steady_clock::now();
        printf("multi-device set-up, 4 handles over %zu points on device 0: one after the other %.2f ms, pcc_index_clone_to_devices %.2f ms\n",
               big, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
        double v = -5;
        REQUIRE(pcc_index_get_option(par[3], PCC_OPT_FAR_MODE, &v) == PCC_OK && v == 1);
        size_t nv = 0;
        for (int k = 0; k < 4; ++k) REQUIRE(pcc_index_size(par[k], &nv) == PCC_OK && nv == big);
        const size_t nq = 50000;
        std::vector<int32_t> ia(nq), ib(nq); std::vector<float> da(nq), db(nq);
        REQUIRE(pcc_nn1(src, bc->points.data() + 1000, nq, sizeof(PointXYZRGB), PCC_MEM_HOST, ia.data(), da.data()) == PCC_OK);
        // the same queries through device pointers on a clone
        void *dq = nullptr, *di = nullptr, *dd = nullptr;
        REQUIRE(hipMalloc(&dq, nq * sizeof(PointXYZRGB)) == hipSuccess && hipMalloc(&di, nq * 4) == hipSuccess && hipMalloc(&dd, nq * 4) == hipSuccess);
        REQUIRE(hipMemcpy(dq, bc->points.data() + 1000, nq * sizeof(PointXYZRGB), hipMemcpyHostToDevice) == hipSuccess);
        REQUIRE(pcc_nn1(par[2], dq, nq, sizeof(PointXYZRGB), PCC_MEM_DEVICE, (int32_t*)di, (float*)dd) == PCC_OK);
        REQUIRE(pcc_index_sync(par[2]) == PCC_OK);
        REQUIRE(hipMemcpy(ib.data(), di, nq * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(db.data(), dd, nq * 4, hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(ia == ib && da == db);
        for (int k = 0; k < 4; ++k) REQUIRE(ia[k] == 1000 + k);
        const int bad[2] = {0, 99};
        pcc_index* none[2] = {src, src};
        REQUIRE(pcc_index_clone_to_devices(src, bad, 2, none) == PCC_ERR_INVALID && none[0] == nullptr && none[1] == nullptr);
        // the C++ mirror on top of it: ShardedKdTree with its shards searched through device pointers
        ShardedKdTree<PointXYZRGB> sh(std::vector<int>{0, 0});
        sh.setInputCloud(bc);
        sh.nearestKSearchShardDevice(1, dq, nq, sizeof(PointXYZRGB), (int*)di, (float*)dd);
        sh.syncShards();
        REQUIRE(hipMemcpy(ib.data(), di, nq * 4, hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(ia == ib);
        (void)hipFree(dq); (void)hipFree(di); (void)hipFree(dd);
        for (int k = 0; k < 4; ++k) { pcc_index_destroy(seq[k]); pcc_index_destroy(par[k]); }
        pcc_index_destroy(src);
    }

    // EuclideanClusterExtraction (src/segmentation.cpp:125-131)
    EuclideanClusterExtraction<PointXYZRGB> ec;
    ec.setClusterTolerance(0.05); ec.setMinClusterSize(100); ec.setMaxClusterSize(250000);
    ec.setSearchMethod(tree); ec.setInputCloud(cloud);
    std::vector<PointIndices> clusters;
    ec.extract(clusters);
    REQUIRE(clusters.size() == 3);
    for (auto& c : clusters) { REQUIRE(c.indices.size() == 400); REQUIRE(std::is_sorted(c.indices.begin(), c.indices.end())); }

    // StatisticalOutlierRemoval (src/comparator.cpp:1523-1527)
    StatisticalOutlierRemoval<PointXYZRGB> sor;
    PointCloud<PointXYZRGB> filtered;
    sor.setInputCloud(cloud); sor.setMeanK(50); sor.setStddevMulThresh(1.5); sor.filter(filtered);
    REQUIRE(filtered.size() < cloud->size() && filtered.size() >= 1200);  // stragglers are the outliers

    // performICP (src/comparator.cpp:1089-1110): cloud2 = cloud1 shifted slightly
    PointCloud<PointXYZRGB>::Ptr moved(new PointCloud<PointXYZRGB>(*cloud));
    for (auto& p : moved->points) { p.x += 0.01f; p.y -= 0.005f; }
    REQUIRE(performICP(moved, cloud));

    // the RCCL form of the sharded tree on the one GPU present (a one-rank communicator: broadcast and all-reduces run for
    // real): sharded ICP == pcl-shaped IterativeClosestPoint, sharded SOR == pcc_sor, bit for bit
    {
        ShardedKdTree<PointXYZRGB> rt(std::vector<int>{0});
        rt.setUseRccl(true);
        rt.setInputCloud(cloud);
        REQUIRE(rt.hasCollectives() && rt.shards() == 1);
        std::vector<int> i1, i2; std::vector<float> d1, d2b;
        tree->nearestKSearchBatch(*cloud, i1, d1);
        rt.nearestKSearchBatch(*cloud, i2, d2b);
        REQUIRE(i1 == i2 && d1 == d2b);
        IterativeClosestPoint<PointXYZRGB, PointXYZRGB> icp;
        icp.setMaximumIterations(20); icp.setInputSource(moved); icp.setInputTarget(cloud);
        PointCloud<PointXYZRGB> fin;
        icp.align(fin);
        std::array<float, 16> Ts; double fit = 0; int its = 0; bool conv = false;
        rt.icpAlign(*moved, 20, Ts, fit, its, conv);
        REQUIRE(Ts == icp.getFinalTransformation() && fit == icp.getFitnessScore() && its == icp.getIterations() && conv == icp.hasConverged());
        std::vector<float> m1(cloud->size()), m2; std::vector<uint8_t> in1(cloud->size()), in2; double t1 = 0, t2 = 0; size_t k1 = 0, k2 = 0;
        REQUIRE(pcc_sor(rt.handle(0), 50, 1.5, PCC_MEM_HOST, m1.data(), in1.data(), &t1, &k1) == PCC_OK);
        rt.sor(50, 1.5, m2, in2, t2, k2);
        REQUIRE(m1 == m2 && in1 == in2 && t1 == t2 && k1 == k2);
        std::printf("sharded tree over RCCL (1 rank): ICP %d iterations, fitness %.6g; SOR kept %zu of %zu\n", its, fit, k2, cloud->size());
    }

    // matchRIFTFeaturesKnn (src/comparator.cpp:560-588): identical descriptor sets match 1:1
    PointCloud<RIFT32>::Ptr d1(new PointCloud<RIFT32>), dd(new PointCloud<RIFT32>);
    for (int i = 0; i < 300; ++i) { RIFT32 h; for (float& v : h.histogram) v = U(rng); d1->push_back(h); }
    *dd = *d1;
    std::vector<int> m = matchRIFTFeaturesKnn(d1, dd);
    REQUIRE(m.size() == 301 && m[0] == 0);
    for (int i = 0; i < 300; ++i) {
        // duplicates in the first three bins aside, element i matches itself
        const float* a = d1->points[i].histogram; const float* b = d1->points[m[i + 1]].histogram;
        REQUIRE(a[0] == b[0] && a[1] == b[1] && a[2] == b[2]);
    }
    PointCloud<RIFT32>::Ptr empty(new PointCloud<RIFT32>);
    REQUIRE(matchRIFTFeaturesKnn(empty, dd).size() == 1);  // empty tree: only the dummy element

    // NormalEstimation + RegionGrowing (src/segmentation.cpp:232-271): two perpendicular plates
    PointCloud<PointXYZRGB>::Ptr plates(new PointCloud<PointXYZRGB>);
    for (int i = 0; i < 3000; ++i) { PointXYZRGB p; p.x = 1 + U(rng); p.y = 1 + U(rng); p.z = 1 + 0.0005f * U(rng); plates->push_back(p); }
    for (int i = 0; i < 3000; ++i) { PointXYZRGB p; p.x = 1 + U(rng); p.y = 3 + 0.0005f * U(rng); p.z = 1.2f + U(rng); plates->push_back(p); }
    search::KdTree<PointXYZRGB>::Ptr ptree(new search::KdTree<PointXYZRGB>);
    PointCloud<Normal>::Ptr normals(new PointCloud<Normal>);
    NormalEstimation<PointXYZRGB, Normal> ne;
    ne.setSearchMethod(ptree); ne.setInputCloud(plates); ne.setKSearch(50);
    ne.compute(*normals);
    REQUIRE(normals->size() == plates->size() && normals->is_dense);
    for (int i = 0; i < 6000; ++i) {
        const Normal& q = normals->points[i];
        // flipped towards the origin: the z plate looks down, the y plate looks towards -y
        if (i < 3000) REQUIRE(q.normal_z < -0.99f); else REQUIRE(q.normal_y < -0.99f);
        REQUIRE(q.curvature >= 0.f && q.curvature < 0.01f);
    }
    NormalEstimation<PointXYZRGB, Normal> ner;  // the RIFT pipeline's form: radius search (src/comparator.cpp:628-635)
    PointCloud<Normal>::Ptr normals_r(new PointCloud<Normal>);
    ner.setSearchMethod(ptree); ner.setInputCloud(plates); ner.setRadiusSearch(0.08);
    ner.compute(*normals_r);
    REQUIRE(normals_r->size() == plates->size());
    for (int i = 0; i < 6000; i += 7) {
        const Normal& q = normals_r->points[i];
        if (std::isfinite(q.normal_x)) { if (i < 3000) REQUIRE(q.normal_z < -0.99f); else REQUIRE(q.normal_y < -0.99f); }
    }
    RegionGrowing<PointXYZRGB, Normal> reg;
    reg.setMinClusterSize(50); reg.setMaxClusterSize(1000000); reg.setSearchMethod(ptree);
    reg.setNumberOfNeighbours(100); reg.setInputCloud(plates); reg.setInputNormals(normals);
    reg.setSmoothnessThreshold(3.0 / 180.0 * M_PI); reg.setCurvatureThreshold(1);
    std::vector<PointIndices> regions;
    reg.extract(regions);
    REQUIRE(regions.size() == 2);
    for (auto& c : regions) { REQUIRE(c.indices.size() == 3000); REQUIRE(std::is_sorted(c.indices.begin(), c.indices.end())); }
    REQUIRE((regions[0].indices[0] < 3000) != (regions[1].indices[0] < 3000));
    RegionGrowing<PointXYZRGB, Normal> reg0;  // no normals set: PCL returns no clusters
    reg0.setInputCloud(plates);
    reg0.extract(regions);
    REQUIRE(regions.empty());

    // SACSegmentation + ExtractIndices (src/segmentation.cpp:79-117): the z plate is the larger plane of `plates`
    for (int i = 0; i < 2000; ++i) { PointXYZRGB p; p.x = 1 + U(rng); p.y = 1 + U(rng); p.z = 1 + 0.0005f * U(rng); plates->push_back(p); }
    SACSegmentation<PointXYZRGB> seg;
    seg.setOptimizeCoefficients(true); seg.setModelType(SACMODEL_PLANE); seg.setMethodType(SAC_RANSAC);
    seg.setMaxIterations(100); seg.setDistanceThreshold(0.02); seg.setInputCloud(plates);
    std::shared_ptr<PointIndices> inl(new PointIndices);
    ModelCoefficients coef;
    seg.segment(*inl, coef);
    REQUIRE(inl->indices.size() == 5000 && coef.values.size() == 4);
    REQUIRE(std::fabs(std::fabs(coef.values[2]) - 1.f) < 1e-3f && std::fabs(std::fabs(coef.values[3]) - 1.f) < 2e-2f);
    REQUIRE(std::is_sorted(inl->indices.begin(), inl->indices.end()));
    ExtractIndices<PointXYZRGB> ex;
    PointCloud<PointXYZRGB> plane_pts, rest;
    ex.setInputCloud(plates); ex.setIndices(inl);
    ex.setNegative(false); ex.filter(plane_pts);
    ex.setNegative(true); ex.filter(rest);
    REQUIRE(plane_pts.size() == 5000 && rest.size() == 3000);
    for (auto& p : rest.points) REQUIRE(p.y > 2.9f);
    std::printf("host mirror ok\n");
    return 0;
}
