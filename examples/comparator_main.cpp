// examples/comparator_main.cpp -- an EXAMPLE of the integration INTEGRATION.md describes, not product source: the
// reference's own call sites (its two segmentation functions, its option parsing, its printed lines) re-pointed at the
// pcc:: shim, so that the drop-in claim can be exercised end to end (tests/test_cli_gpu.py).  What follows mirrors the
// reference statement by statement on purpose -- that is what "the maintainer changes the namespace and nothing else"
// looks like.  The product is include/ + pointcloudcomparator_amd/csrc/ + host/ply_io.hpp + host/report.hpp.
//
// The `./comparator [-n] [-v] [-i] [-e] scene1.ply scene2.ply` front end of the
// reference (src/comparator.cpp:1641-1705 main, :1112-1636 computeSimilarity) for the stages that sit
// on the nearest-neighbour path this repository accelerates:
//   load 2 PLY (:1119,:1130; -2 on failure) -> NaN strip (:1144-1148) -> [-i] ICP gate (:1152-1186, -1
//   when it does not converge) -> [-e] Euclidean clustering (src/segmentation.cpp:119-156) -> point /
//   cluster counts (:1199-1205) -> [-n] noise pass (:1520-1568) -> results file.
// Same flags, banner lines, section strings and exit code (always 1, :1704).  NOT part of this build
// (out of scope): descriptor pipelines (SIFT/RIFT) and the per-cluster matching /
// scoring that needs them, and the viewer (-v is accepted and ignored).  Both segmentation paths run: region growing (default, src/segmentation.cpp:218-327) and
// Euclidean clustering (-e, :64-156), each behind the VoxelGrid the reference applies first.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <algorithm>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include "pcc/comparator_nn.hpp"
#include "pcc/multi_device.hpp"
#include "ply_io.hpp"
#include "report.hpp"

using namespace pcc;

static bool seeClusters = false, noise = false, euclidean = false, icp = false;
// not in the reference: --gpus N (the two clouds are segmented / filtered as replicas on two devices),
// --descriptors1/2 FILE (precomputed RIFT32 descriptors per cluster: the descriptor pipeline itself is out of scope),
// --dump-clusters PREFIX (the clusters as PLY files, so that descriptors can be computed for them elsewhere)
static int n_gpus = 1;
static std::string descriptors_path[2], dump_prefix;

static void printUsage() {
    std::cout << "\n\nUsage: [options] </pathToScene1.ply> </pathToScene2.ply>\n\n"
              << "Options:\n" << "-------------------------------------------\n"
              << "-n activate noise analysis \n"
              << "-v activate visualization of clusters and matches\n"
              << "-i activate ICP algorithm to know if both point clouds are enough similar \n"
              << "-e activate euclidean cluster segmentation as main segmentation algorithm; region growing segmentation is default\n"
              << "-h show this help\n"
              << "--gpus N           (this build) N devices: ICP source sharded, the two clouds segmented / filtered as replicas\n"
              << "--descriptors1 F   (this build) precomputed RIFT32 descriptors of the clusters of scene 1\n"
              << "--descriptors2 F   (this build) ... of scene 2\n"
              << "--dump-clusters P  (this build) write the clusters as P_<scene>_<cluster>.ply\n"
              << "--results F        (this build) results file (default ../../PointCloudComparatorResults/results.txt)\n" << "\n\n";
}

// src/segmentation.cpp:64-156 euclidean_cluster_segmentation: VoxelGrid 0.025 (:69-76) -> RANSAC plane
// removal until <= 30 % of the points remain (:79-117) -> KdTree + EuclideanClusterExtraction (:119-156)
static std::vector<PointCloud<PointXYZRGB>::Ptr> euclidean_cluster_segmentation(const PointCloud<PointXYZRGB>::Ptr& point_cloud_ptr) {
    VoxelGrid<PointXYZRGB> vg;
    PointCloud<PointXYZRGB>::Ptr cloud_filtered(new PointCloud<PointXYZRGB>);
    vg.setInputCloud(point_cloud_ptr);
    vg.setLeafSize(0.025f, 0.025f, 0.025f);
    vg.filter(*cloud_filtered);
    std::cout << "PointCloud after filtering has: " << cloud_filtered->points.size() << " data points." << std::endl;
    SACSegmentation<PointXYZRGB> seg;
    std::shared_ptr<PointIndices> inliers(new PointIndices);
    ModelCoefficients coefficients;
    PointCloud<PointXYZRGB>::Ptr cloud_plane(new PointCloud<PointXYZRGB>), cloud_f(new PointCloud<PointXYZRGB>);
    seg.setOptimizeCoefficients(true);
    seg.setModelType(SACMODEL_PLANE);
    seg.setMethodType(SAC_RANSAC);
    seg.setMaxIterations(100);
    seg.setDistanceThreshold(0.02);
    const int nr_points = (int)cloud_filtered->points.size();
    while (cloud_filtered->points.size() > 0.3 * nr_points) {
        seg.setInputCloud(cloud_filtered);
        seg.segment(*inliers, coefficients);
        if (inliers->indices.size() == 0) {
            std::cout << "Could not estimate a planar model for the given dataset." << std::endl;
            break;
        }
        ExtractIndices<PointXYZRGB> extract;
        extract.setInputCloud(cloud_filtered);
        extract.setIndices(inliers);
        extract.setNegative(false);
        extract.filter(*cloud_plane);
        std::cout << "PointCloud representing the planar component: " << cloud_plane->points.size() << " data points." << std::endl;
        extract.setNegative(true);
        extract.filter(*cloud_f);
        // the reference assigns *cloud_filtered = *cloud_f; a fresh cloud keeps the pointer-keyed caches honest
        cloud_filtered.reset(new PointCloud<PointXYZRGB>(*cloud_f));
    }
    search::KdTree<PointXYZRGB>::Ptr tree(new search::KdTree<PointXYZRGB>);
    tree->setInputCloud(cloud_filtered);
    std::vector<PointIndices> cluster_indices;
    EuclideanClusterExtraction<PointXYZRGB> ec;
    ec.setClusterTolerance(0.05);
    ec.setMinClusterSize(100);
    ec.setMaxClusterSize(250000);
    ec.setSearchMethod(tree);
    ec.setInputCloud(cloud_filtered);
    ec.extract(cluster_indices);
    std::vector<PointCloud<PointXYZRGB>::Ptr> clusters_pcl;
    for (const PointIndices& it : cluster_indices) {
        PointCloud<PointXYZRGB>::Ptr cloud_cluster(new PointCloud<PointXYZRGB>);
        for (int pit : it.indices) cloud_cluster->points.push_back(cloud_filtered->points[pit]);
        cloud_cluster->width = (std::uint32_t)cloud_cluster->points.size();
        cloud_cluster->height = 1;
        cloud_cluster->is_dense = true;
        std::cout << "PointCloud representing the Cluster: " << cloud_cluster->points.size() << " data points." << std::endl;
        clusters_pcl.push_back(cloud_cluster);
    }
    return clusters_pcl;
}

// src/segmentation.cpp:218-327 region_growing_segmentation: VoxelGrid 0.025 -> NormalEstimation K = 50 ->
// RegionGrowing (50..1000000 points, 100 neighbours, 3 degrees, curvature 1)
static std::vector<PointCloud<PointXYZRGB>::Ptr> region_growing_segmentation(const PointCloud<PointXYZRGB>::Ptr& point_cloud_ptr) {
    VoxelGrid<PointXYZRGB> vg;
    PointCloud<PointXYZRGB>::Ptr cloud_filtered(new PointCloud<PointXYZRGB>);
    vg.setInputCloud(point_cloud_ptr);
    vg.setLeafSize(0.025f, 0.025f, 0.025f);
    vg.filter(*cloud_filtered);
    std::cout << "PointCloud after filtering has: " << cloud_filtered->points.size() << " data points." << std::endl;
    search::KdTree<PointXYZRGB>::Ptr tree(new search::KdTree<PointXYZRGB>);
    tree->setOption(PCC_OPT_KNN_CACHE_K, 100);  // normals (50) and region growing (100) share one 100-neighbour search
    PointCloud<Normal>::Ptr normals(new PointCloud<Normal>);
    NormalEstimation<PointXYZRGB, Normal> normal_estimator;
    normal_estimator.setSearchMethod(tree);
    normal_estimator.setInputCloud(cloud_filtered);
    normal_estimator.setKSearch(50);
    normal_estimator.compute(*normals);
    RegionGrowing<PointXYZRGB, Normal> reg;
    reg.setMinClusterSize(50);
    reg.setMaxClusterSize(1000000);
    reg.setSearchMethod(tree);
    reg.setNumberOfNeighbours(100);
    reg.setInputCloud(cloud_filtered);
    reg.setInputNormals(normals);
    reg.setSmoothnessThreshold(3.0 / 180.0 * M_PI);
    reg.setCurvatureThreshold(1);
    std::vector<PointIndices> clusters;
    reg.extract(clusters);
    std::cout << "Number of clusters is equal to " << clusters.size() << std::endl;
    std::vector<PointCloud<PointXYZRGB>::Ptr> clusters_pcl;
    for (const PointIndices& c : clusters) {
        PointCloud<PointXYZRGB>::Ptr cloud_cluster(new PointCloud<PointXYZRGB>);
        for (int j : c.indices) cloud_cluster->points.push_back(cloud_filtered->points[j]);
        cloud_cluster->width = (std::uint32_t)cloud_cluster->points.size();
        cloud_cluster->height = 1;
        cloud_cluster->is_dense = true;
        clusters_pcl.push_back(cloud_cluster);
    }
    return clusters_pcl;
}

static double computeSimilarity(const std::string& file1, const std::string& file2, const std::string& results_path) {
    PointCloud<PointXYZRGB>::Ptr point_cloud1_ptr(new PointCloud<PointXYZRGB>), point_cloud2_ptr(new PointCloud<PointXYZRGB>);
    if (io::loadPLYFile(file1, *point_cloud1_ptr) == -1) {
        std::cerr << "Was not able to open file \"" << file1 << "\".\n";
        return -2;
    }
    if (io::loadPLYFile(file2, *point_cloud2_ptr) == -1) {
        std::cerr << "Was not able to open file \"" << file2 << "\".\n";
        return -2;
    }
    report::Writer out(results_path);
    std::ofstream& myfile = out.file();
    out.header(file1, file2);
    std::vector<int> indices2;
    io::removeNaNFromPointCloud(*point_cloud1_ptr, indices2);
    io::removeNaNFromPointCloud(*point_cloud2_ptr, indices2);

    if (icp) {
        // --gpus N: the source cloud of the ICP gate is sharded over the devices (17 sums all-reduced per pass over RCCL)
        std::vector<int> icp_devices;
        for (int d = 0; d < std::min(n_gpus, deviceCount()); ++d) icp_devices.push_back(d);
        const bool icp_ok = icp_devices.size() > 1 ? performICP(point_cloud1_ptr, point_cloud2_ptr, icp_devices)
                                                   : performICP(point_cloud1_ptr, point_cloud2_ptr);
        if (!icp_ok) {
            myfile << "----------------------------" << "\n\n";
            myfile << "ICP could not match the point clouds. They are probably too dissimilar.\n Brief comparison:\n";
            const size_t n1 = point_cloud1_ptr->points.size(), n2 = point_cloud2_ptr->points.size();
            if (n1 > n2) {
                std::cout << "PCL1 has more points: " << n1 << " over: " << n2 << std::endl;
                myfile << "PCL1 has more points: " << n1 << " over: " << n2 << "\n";
            } else if (n2 > n1) {
                std::cout << "PCL2 has more points: " << n2 << " over: " << n1 << std::endl;
                myfile << "PCL2 has more points: " << n2 << " over: " << n1 << "\n";
            } else {
                std::cout << "Both PCL have the same number of points" << std::endl;
                myfile << "Both PCL have the same number of points" << "\n";
            }
            myfile.close();
            return -1;
        } else {
            std::cout << "ICP has converged. Point clouds segmentation is as follows" << std::endl;
            myfile << "ICP has converged. Point clouds segmentation is as follows: \n";
        }
    }

    // the two clouds are independent until the cluster matching: replicas, one device each when there are several
    // (clustering itself does not shard).  With one device the two jobs run one after the other, in
    // the reference's order, so its printed lines keep their order.
    std::vector<int> devices;
    for (int d = 0; d < std::max(1, std::min(n_gpus, deviceCount())); ++d) devices.push_back(d);
    std::vector<PointCloud<PointXYZRGB>::Ptr> clusters[2];
    const PointCloud<PointXYZRGB>::Ptr clouds[2] = {point_cloud1_ptr, point_cloud2_ptr};
    auto segment = [&](int k) { clusters[k] = euclidean ? euclidean_cluster_segmentation(clouds[k]) : region_growing_segmentation(clouds[k]); };
    if (devices.size() > 1) onDevices(2, devices, segment);
    else { segment(0); segment(1); }
    std::vector<PointCloud<PointXYZRGB>::Ptr>&clusters_pcl_1 = clusters[0], &clusters_pcl_2 = clusters[1];

    report::Writer& w = out;
    w.counts(point_cloud1_ptr->points.size(), point_cloud2_ptr->points.size(), clusters_pcl_1.size(), clusters_pcl_2.size());
    if (!dump_prefix.empty())
        for (int k = 0; k < 2; ++k)
            for (size_t j = 0; j < clusters[k].size(); ++j) {
                std::ostringstream name;
                name << dump_prefix << "_" << k + 1 << "_" << j << ".ply";
                io::savePLYFileBinary(name.str(), *clusters[k][j]);
            }

    // descriptors per cluster: from files, or none (the SIFT / RIFT pipeline is not part of this build)
    std::vector<report::DescPtr> des[2];
    bool have_descriptors = true;
    for (int k = 0; k < 2; ++k) {
        if (descriptors_path[k].empty()) {
            have_descriptors = false;
            des[k].assign(clusters[k].size(), report::DescPtr());
            for (report::DescPtr& d : des[k]) d.reset(new PointCloud<RIFT32>);
        } else if (!report::loadDescriptors(descriptors_path[k], clusters[k].size(), des[k])) {
            std::cerr << "Was not able to read descriptors \"" << descriptors_path[k] << "\".\n";
            return -2;
        }
    }
    std::vector<int> matches;
    const report::Scores scores = report::clusterSections(w, clusters_pcl_1, clusters_pcl_2, des[0], des[1], matches);

    if (noise) {
        PointCloud<PointXYZRGB> nonoise[2];
        auto filter = [&](int k) {
            StatisticalOutlierRemoval<PointXYZRGB> sor;
            sor.setInputCloud(clouds[k]); sor.setMeanK(50); sor.setStddevMulThresh(1.5); sor.filter(nonoise[k]);
        };
        if (devices.size() > 1) onDevices(2, devices, filter);
        else { filter(0); filter(1); }
        // size_t integer division, as in the reference (:1533-1535): the ratio is always 0
        double noise1 = (point_cloud1_ptr->points.size() - nonoise[0].points.size()) / point_cloud1_ptr->points.size();
        double noise2 = (point_cloud2_ptr->points.size() - nonoise[1].points.size()) / point_cloud2_ptr->points.size();
        std::cout << "Noise pass removed " << point_cloud1_ptr->points.size() - nonoise[0].points.size() << " / "
                  << point_cloud2_ptr->points.size() - nonoise[1].points.size() << " points" << std::endl;
        myfile << "----------------------------------------\n Noise analysis: \n";
        if (noise1 > noise2) {
            std::cout << "PCL1 has more noisy points: (%) " << noise1 * 100 << " over: (%) " << noise2 * 100 << std::endl;
            myfile << "\tPCL1 has more noisy points: (%) " << noise1 * 100 << " over: (%) " << noise2 * 100 << "\n";
        } else if (noise1 < noise2) {
            std::cout << "PCL2 has more noisy points: (%) " << noise2 * 100 << " over: (%) " << noise1 * 100 << std::endl;
            myfile << "\tPCL2 has more noisy points: (%) " << noise2 * 100 << " over: (%) " << noise1 * 100 << "\n";
        } else {
            std::cout << "Both pcl have the same percentage of noisy points: " << noise1 * 100 << std::endl;
            myfile << "Both pcl have the same percentage of noisy points: " << noise1 * 100 << "\n";
        }
    }
    const int verdict = report::scoreSections(w, scores, clusters_pcl_2.size());
    if (!have_descriptors) {
        // nothing was scored: saying "same information" would report a comparison that never happened
        myfile << "\n(no descriptor files given: clusters could not be matched, no verdict)\n";
        w.close();
        return -3;
    }
    w.close();
    return verdict;
}

int main(int argc, char** argv) {
    std::cout << "------------------------------------" << std::endl;
    std::vector<std::string> plys;
    std::string results_path = "../../PointCloudComparatorResults/results.txt";  // reference :1138
    bool help = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h") help = true;
        else if (a == "-v") seeClusters = true;
        else if (a == "-n") noise = true;
        else if (a == "-i") icp = true;
        else if (a == "-e") euclidean = true;
        else if (a == "--results" && i + 1 < argc) results_path = argv[++i];
        else if (a == "--gpus" && i + 1 < argc) n_gpus = std::atoi(argv[++i]);
        else if (a == "--descriptors1" && i + 1 < argc) descriptors_path[0] = argv[++i];
        else if (a == "--descriptors2" && i + 1 < argc) descriptors_path[1] = argv[++i];
        else if (a == "--dump-clusters" && i + 1 < argc) dump_prefix = argv[++i];
        else if (a.size() > 4 && a.substr(a.size() - 4) == ".ply") plys.push_back(a);
    }
    if (help) { printUsage(); return 1; }
    std::cout << (seeClusters ? "Visualization of clusters is on." : "Visualization of clusters is off.") << std::endl;
    std::cout << (noise ? "Noise analysis is on." : "Noise analysis is off.") << std::endl;
    std::cout << (icp ? "ICP matching pre-comparison is on." : "ICP matching pre-comparison is off.") << std::endl;
    if (euclidean) std::cout << "Euclidean cluster segmentation was selected as main segmentation algorithm." << std::endl;
    else std::cout << "Region growing segmentation was selected (default) as main segmentation algorithm." << std::endl;
    std::cout << "------------------------------------" << std::endl;
    if (plys.size() < 2) { printUsage(); return 1; }  // the reference dereferences blindly (:1130); we print the usage
    double similarity = -3;
    try {
        similarity = computeSimilarity(plys[0], plys[1], results_path);
    } catch (const std::exception& e) {  // pcc::Error, and whatever a broken input makes the standard library throw
        std::cerr << e.what() << std::endl;
    }
    std::cout << "--------------------------------\n" << std::endl;
    if (similarity == 1) std::cout << "The first point cloud has more information" << std::endl;
    else if (similarity == 2) std::cout << "The second point cloud has more information" << std::endl;
    else if (similarity == 0) std::cout << "Both point clouds have the same information" << std::endl;
    else if (similarity == -3) std::cout << "No descriptor files were given (--descriptors1/2): clusters were not matched, no verdict" << std::endl;
    std::cout << "--------------------------------\n" << std::endl;
    return 1;  // reference :1704
}
