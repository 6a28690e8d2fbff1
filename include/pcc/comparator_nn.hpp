// comparator_nn.hpp -- host-side mirror of the reference's functions and PCL algorithm objects
// that sit on the nearest-neighbour path, re-hosted on libpcc_nn (same names, argument meaning,
// printed lines and error behaviour):
//   matchRIFTFeaturesKnn            reference src/comparator.cpp:560-588
//   performICP                      reference src/comparator.cpp:1089-1110
//   IterativeClosestPoint           the PCL object performICP drives (:1091-1099)
//   StatisticalOutlierRemoval       the -n noise pass (:1520-1549)
//   EuclideanClusterExtraction      the -e path (src/segmentation.cpp:119-131)
// Everything numerical happens behind the C-ABI; this header is plumbing a maintainer of the
// reference can include instead of the PCL headers for these five call sites.
#pragma once
#include <algorithm>
#include <array>
#include <cfloat>
#include <iostream>
#include "pcc/search.hpp"

namespace pcc {

typedef Histogram<32> RIFT32;  // reference src/comparator.cpp:9

// ---- matchRIFTFeaturesKnn (src/comparator.cpp:560-588) --------------------------------------
// Tree on descriptors1, one k=1 query per element of descriptors2, match kept when
// neighborCount == 1 && d2 < 0.05f.  The returned vector STARTS WITH ONE DUMMY 0 (:568), so
// size() == matches + 1 -- callers divide size() by descriptor counts (:1336-1338).
inline std::vector<int> matchRIFTFeaturesKnn(const PointCloud<RIFT32>::Ptr& descriptors1,
                                             const PointCloud<RIFT32>::Ptr& descriptors2) {
    std::vector<int> correspondence(1);
    // `= new KdTreeFLANN<RIFT32>(false)` in the reference: sorted == true, and one (leaked) tree per
    // call.  Here one tree per thread is re-pointed at each descriptor cloud: the reference calls this
    // up to three times per cluster (:1322) and a fresh device handle per call would cost ~1 ms of
    // allocations for microseconds of work.
    static thread_local KdTreeFLANN<RIFT32> matching;
    // the matched indices go back to the caller (:580): among descriptors at exactly the same distance name the one
    // PCL's FLANN tree would (descriptor clouds are full of duplicates in their first three bins)
    matching.setTieOrder(PCC_TIES_FLANN);
    matching.setInputCloud(descriptors1);
    if (!matching.handle() || !descriptors2 || descriptors2->empty()) return correspondence;
    std::vector<int> out(descriptors2->size() + 1);
    int32_t n = 0;
    check(pcc_match_knn(matching.handle(), descriptors2->points.data(), descriptors2->size(), sizeof(RIFT32),
                        PCC_MEM_HOST, 0.05f, out.data(), &n));
    out.resize(n);
    return out;
}

// ---- pcl::VoxelGrid (src/segmentation.cpp:69-74, 224-229) -------------------------------------------
template <class PointT>
class VoxelGrid {
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
    void setLeafSize(float lx, float ly, float lz) {
        if (lx != ly || ly != lz) throw Error(PCC_ERR_UNSUPPORTED, "anisotropic leaf sizes are not supported");
        leaf_ = lx;
    }
    void filter(PointCloud<PointT>& output) {
        PointCloud<PointT> result;
        if (input_ && !input_->empty()) {
            search::KdTree<PointT> ctx;  // supplies device, stream and scratch
            ctx.setInputCloud(input_);
            if (ctx.handle()) {
                result.points.resize(input_->size());
                size_t nv = 0;
                const int has_rgb = sizeof(PointT) >= 20 ? 1 : 0;
                int st = pcc_voxel_grid(ctx.handle(), input_->points.data(), input_->size(), sizeof(PointT), PCC_MEM_HOST,
                                        leaf_, has_rgb, result.points.data(), sizeof(PointT), &nv);
                if (st == PCC_ERR_UNSUPPORTED) { result = *input_; nv = result.size(); }  // PCL: "leaf size is too small", output = input
                else check(st);
                result.points.resize(nv);
            }
        }
        result.width = (std::uint32_t)result.points.size();
        result.height = 1;
        result.is_dense = true;
        output = result;
    }

private:
    typename PointCloud<PointT>::ConstPtr input_;
    float leaf_ = 0.f;
};

// ---- keypoint snap (src/comparator.cpp:696-713, inside processRIFTwithSIFT) -----------------------
// For every keypoint the FIRST cloud point within `radius` (0.05 in the reference; float differences,
// double-precision distance) is appended to the result, keypoints without one are skipped.
template <class KeyPointT>
inline PointCloud<PointXYZRGB>::Ptr snapKeypointsToCloud(const PointCloud<PointXYZRGB>::Ptr& cloud,
                                                         const PointCloud<KeyPointT>& keypoints, double radius = 0.05,
                                                         search::KdTree<PointXYZRGB>* tree = nullptr) {
    PointCloud<PointXYZRGB>::Ptr out(new PointCloud<PointXYZRGB>);
    if (!cloud || cloud->empty() || keypoints.empty()) return out;
    search::KdTree<PointXYZRGB> local;
    if (!tree) { local.setInputCloud(cloud); tree = &local; }
    if (!tree->handle()) return out;
    std::vector<int32_t> idx(keypoints.size());
    check(pcc_first_within(tree->handle(), keypoints.points.data(), keypoints.size(), sizeof(KeyPointT), PCC_MEM_HOST,
                           radius, idx.data()));
    for (int32_t j : idx)
        if (j >= 0) out->push_back(cloud->points[j]);
    return out;
}

// ---- pcl::IterativeClosestPoint ---------------------------------------------------------------
template <class PointSource, class PointTarget>
class IterativeClosestPoint {
public:
    typedef std::array<float, 16> Matrix4;  // row-major 4x4
    void setMaximumIterations(int n) { max_iterations_ = n; }
    void setInputSource(const typename PointCloud<PointSource>::ConstPtr& c) { source_ = c; }
    void setInputTarget(const typename PointCloud<PointTarget>::ConstPtr& c) { target_ = c; tree_.setInputCloud(c); }
    // align: the ICP loop with PCL's defaults (no distance threshold, SVD/Umeyama estimate,
    // DefaultConvergenceCriteria: iteration cap or |mse - prev| < 1e-12); output = transformed source
    void align(PointCloud<PointSource>& output) {
        converged_ = false;
        final_ = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        fitness_ = DBL_MAX;
        output = *source_;
        if (!tree_.handle() || source_->empty()) return;
        int conv = 0, it = 0;
        check(pcc_icp_align(tree_.handle(), source_->points.data(), source_->size(), sizeof(PointSource), PCC_MEM_HOST,
                            max_iterations_, 0, final_.data(), &fitness_, &it, &conv));
        converged_ = conv != 0;
        iterations_ = it;
        check(pcc_transform(tree_.handle(), final_.data(), source_->points.data(), source_->size(), sizeof(PointSource),
                            output.points.data(), sizeof(PointSource), PCC_MEM_HOST));
    }
    bool hasConverged() const { return converged_; }
    double getFitnessScore() const { return fitness_; }
    Matrix4 getFinalTransformation() const { return final_; }
    int getIterations() const { return iterations_; }

private:
    int max_iterations_ = 10;  // PCL default; the reference sets 20
    int iterations_ = 0;
    bool converged_ = false;
    double fitness_ = DBL_MAX;
    Matrix4 final_{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    typename PointCloud<PointSource>::ConstPtr source_;
    typename PointCloud<PointTarget>::ConstPtr target_;
    search::KdTree<PointTarget> tree_;
};

// ---- performICP (src/comparator.cpp:1089-1110), same printed lines ----------------------------
inline bool performICP(const PointCloud<PointXYZRGB>::Ptr& point_cloud1, const PointCloud<PointXYZRGB>::Ptr& point_cloud2) {
    IterativeClosestPoint<PointXYZRGB, PointXYZRGB> icp;
    icp.setMaximumIterations(20);
    icp.setInputSource(point_cloud1);
    icp.setInputTarget(point_cloud2);
    PointCloud<PointXYZRGB> Final;
    icp.align(Final);
    std::cout << "has converged:" << icp.hasConverged() << " ICP fitness score: " << icp.getFitnessScore() << std::endl;
    if (icp.hasConverged() == 1)
        std::cout << "ICP has converged; starting comparison of point clouds" << std::endl;
    else
        std::cout << "ICP has not converged; point clouds too much different to perform a specific comparison" << std::endl;
    return icp.hasConverged() == 1;
}

// ---- pcl::StatisticalOutlierRemoval (src/comparator.cpp:1523-1527) ---------------------------------
template <class PointT>
class StatisticalOutlierRemoval {
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
    void setMeanK(int k) { mean_k_ = k; }
    void setStddevMulThresh(double m) { std_mul_ = m; }
    void filter(PointCloud<PointT>& output) {
        output.points.clear();
        if (!input_ || input_->empty()) return;
        search::KdTree<PointT> tree(false);
        tree.setInputCloud(input_);
        if (!tree.handle()) return;
        std::vector<std::uint8_t> inlier(input_->size());
        size_t kept = 0;
        check(pcc_sor(tree.handle(), mean_k_, std_mul_, PCC_MEM_HOST, nullptr, inlier.data(), &threshold_, &kept));
        output.points.reserve(kept);
        for (size_t i = 0; i < input_->size(); ++i)
            if (inlier[i]) output.points.push_back(input_->points[i]);
        output.width = (std::uint32_t)output.points.size();
        output.height = 1;
    }
    double getThreshold() const { return threshold_; }

private:
    typename PointCloud<PointT>::ConstPtr input_;
    int mean_k_ = 1;
    double std_mul_ = 0.0, threshold_ = 0.0;
};

// ---- pcl::EuclideanClusterExtraction (src/segmentation.cpp:125-131) ------------------------------------
template <class PointT>
class EuclideanClusterExtraction {
public:
    void setClusterTolerance(double t) { tolerance_ = t; }
    void setMinClusterSize(int n) { min_ = n; }
    void setMaxClusterSize(int n) { max_ = n; }
    void setSearchMethod(const typename search::KdTree<PointT>::Ptr& tree) { tree_ = tree; }
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
    // PCL re-runs tree_->setInputCloud(input_) inside extract; an index already
    // built on the same cloud is reused here.
    void extract(std::vector<PointIndices>& clusters) {
        clusters.clear();
        if (!input_ || input_->empty()) return;
        if (!tree_) tree_.reset(new search::KdTree<PointT>(false));
        if (!tree_->covers(input_)) tree_->setInputCloud(input_);
        if (!tree_->handle()) return;
        std::vector<int32_t> labels(input_->size());
        int32_t ncl = 0;
        check(pcc_euclidean_clusters(tree_->handle(), tolerance_, (uint32_t)min_, (uint32_t)max_, PCC_MEM_HOST,
                                     labels.data(), &ncl, nullptr, 0));
        clusters.resize(ncl);
        for (size_t i = 0; i < labels.size(); ++i)  // ascending i == PCL's sorted indices per cluster
            if (labels[i] >= 0) clusters[labels[i]].indices.push_back((int)i);
    }

private:
    double tolerance_ = 0.0;
    int min_ = 1, max_ = 0x7fffffff;
    typename search::KdTree<PointT>::Ptr tree_;
    typename PointCloud<PointT>::ConstPtr input_;
};

// ---- pcl::NormalEstimation (src/segmentation.cpp:232-241) ----------------------------------------------
template <class PointT, class NormalT = Normal>
class NormalEstimation {
public:
    void setSearchMethod(const typename search::KdTree<PointT>::Ptr& tree) { tree_ = tree; }
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
    void setKSearch(int k) { k_ = k; }
    void setRadiusSearch(double r) { radius_ = r; }  // as in PCL, the radius is used when no K was set
    void setViewPoint(float x, float y, float z) { vp_[0] = x; vp_[1] = y; vp_[2] = z; }
    void compute(PointCloud<NormalT>& out) {
        out.points.clear();
        out.width = 0;
        out.height = 1;
        if (!input_ || input_->empty() || (k_ < 1 && !(radius_ > 0))) return;
        if (!tree_) tree_.reset(new search::KdTree<PointT>(false));
        if (!tree_->covers(input_)) tree_->setInputCloud(input_);
        if (!tree_->handle()) return;
        std::vector<float> nc(input_->size() * 4);
        if (k_ >= 1) check(pcc_normals(tree_->handle(), k_, vp_, PCC_MEM_HOST, nc.data()));
        else check(pcc_normals_radius(tree_->handle(), radius_, vp_, PCC_MEM_HOST, nc.data()));
        out.points.resize(input_->size());
        out.width = (std::uint32_t)input_->size();
        out.is_dense = true;
        for (size_t i = 0; i < out.points.size(); ++i) {
            NormalT& q = out.points[i];
            q.normal_x = nc[4 * i]; q.normal_y = nc[4 * i + 1]; q.normal_z = nc[4 * i + 2]; q.curvature = nc[4 * i + 3];
            if (!std::isfinite(q.normal_x)) out.is_dense = false;  // as PCL flags NaN normals
        }
    }

private:
    typename search::KdTree<PointT>::Ptr tree_;
    typename PointCloud<PointT>::ConstPtr input_;
    int k_ = 0;
    double radius_ = 0.0;
    float vp_[3] = {0.f, 0.f, 0.f};
};

// ---- pcl::RegionGrowing (src/segmentation.cpp:259-271) -------------------------------------------------
template <class PointT, class NormalT = Normal>
class RegionGrowing {
public:
    void setMinClusterSize(int n) { min_ = n; }
    void setMaxClusterSize(int n) { max_ = n; }
    void setSearchMethod(const typename search::KdTree<PointT>::Ptr& tree) { tree_ = tree; }
    void setNumberOfNeighbours(unsigned int k) { k_ = k; }
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
    void setInputNormals(const typename PointCloud<NormalT>::ConstPtr& n) { normals_ = n; }
    void setSmoothnessThreshold(float theta) { theta_ = theta; }
    void setCurvatureThreshold(float c) { curvature_ = c; }
    void extract(std::vector<PointIndices>& clusters) {
        clusters.clear();
        // PCL's prepareForSegmentation: no cloud, no normals or a size mismatch -> empty result
        if (!input_ || input_->empty() || !normals_ || normals_->size() != input_->size() || k_ == 0) return;
        if (!tree_) tree_.reset(new search::KdTree<PointT>(false));
        if (!tree_->covers(input_)) tree_->setInputCloud(input_);
        if (!tree_->handle()) return;
        const size_t n = input_->size();
        std::vector<float> nc(n * 4);
        for (size_t i = 0; i < n; ++i) {
            const NormalT& q = normals_->points[i];
            nc[4 * i] = q.normal_x; nc[4 * i + 1] = q.normal_y; nc[4 * i + 2] = q.normal_z; nc[4 * i + 3] = q.curvature;
        }
        std::vector<int32_t> labels(n);
        int32_t ncl = 0;
        check(pcc_region_growing(tree_->handle(), nc.data(), PCC_MEM_HOST, (int)k_, theta_, curvature_, (uint32_t)min_,
                                 (uint32_t)max_, labels.data(), &ncl));
        clusters.resize(ncl);
        for (size_t i = 0; i < n; ++i)  // assembleRegions fills every cluster in ascending point order
            if (labels[i] >= 0) clusters[labels[i]].indices.push_back((int)i);
    }

private:
    int min_ = 1, max_ = 0x7fffffff;
    unsigned int k_ = 30;
    float theta_ = 30.0f / 180.0f * 3.14159265358979f, curvature_ = 0.05f;
    typename search::KdTree<PointT>::Ptr tree_;
    typename PointCloud<PointT>::ConstPtr input_;
    typename PointCloud<NormalT>::ConstPtr normals_;
};

// ---- pcl::SACSegmentation, plane + RANSAC (src/segmentation.cpp:79-99) --------------------------------
enum SacModel { SACMODEL_PLANE = 0 };
enum SacMethod { SAC_RANSAC = 0 };

template <class PointT>
class SACSegmentation {
public:
    void setOptimizeCoefficients(bool b) { optimize_ = b; }
    void setModelType(int m) { model_ = m; }
    void setMethodType(int m) { method_ = m; }
    void setMaxIterations(int n) { max_iterations_ = n; }
    void setDistanceThreshold(double t) { threshold_ = t; }
    void setProbability(double p) { probability_ = p; }
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
    // inliers.indices empty + coefficients.values empty: no model (as PCL reports it)
    void segment(PointIndices& inliers, ModelCoefficients& coefficients) {
        inliers.indices.clear();
        coefficients.values.clear();
        if (!input_ || input_->empty()) return;
        if (model_ != SACMODEL_PLANE || method_ != SAC_RANSAC) throw Error(PCC_ERR_UNSUPPORTED, "only SACMODEL_PLANE + SAC_RANSAC");
        if (!ctx_) {
            // any index handle supplies device, stream and scratch; a one-point index is the cheapest
            const float one[3] = {0.f, 0.f, 0.f};
            check(pcc_index_create(one, 1, 12, 3, PCC_MEM_HOST, 0, PCC_ENGINE_BRUTE, &ctx_));
        }
        std::vector<int32_t> idx(input_->size());
        size_t m = 0;
        float c[4];
        check(pcc_sac_plane(ctx_, input_->points.data(), input_->size(), sizeof(PointT), PCC_MEM_HOST, max_iterations_,
                            threshold_, probability_, optimize_ ? 1 : 0, idx.data(), &m, c, nullptr));
        if (m == 0) return;
        inliers.indices.assign(idx.begin(), idx.begin() + m);
        coefficients.values.assign(c, c + 4);
    }
    SACSegmentation() = default;
    SACSegmentation(const SACSegmentation&) = delete;
    SACSegmentation& operator=(const SACSegmentation&) = delete;
    ~SACSegmentation() { if (ctx_) pcc_index_destroy(ctx_); }

private:
    bool optimize_ = false;
    int model_ = SACMODEL_PLANE, method_ = SAC_RANSAC, max_iterations_ = 50;
    double threshold_ = 0.0, probability_ = 0.99;
    typename PointCloud<PointT>::ConstPtr input_;
    pcc_index* ctx_ = nullptr;
};

// ---- pcl::ExtractIndices (src/segmentation.cpp:103-116): index bookkeeping, host only ---------------------
template <class PointT>
class ExtractIndices {
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
    void setIndices(const std::shared_ptr<const PointIndices>& i) { indices_ = i; }
    void setNegative(bool n) { negative_ = n; }
    void filter(PointCloud<PointT>& out) {
        std::vector<PointT> pts;
        if (input_) {
            const size_t n = input_->size();
            if (!negative_) {
                if (indices_) for (int i : indices_->indices) pts.push_back(input_->points[i]);
            } else {
                std::vector<char> drop(n, 0);
                if (indices_) for (int i : indices_->indices) drop[i] = 1;
                for (size_t i = 0; i < n; ++i) if (!drop[i]) pts.push_back(input_->points[i]);
            }
        }
        out.points.swap(pts);
        out.width = (std::uint32_t)out.points.size();
        out.height = 1;
        out.is_dense = true;
    }

private:
    typename PointCloud<PointT>::ConstPtr input_;
    std::shared_ptr<const PointIndices> indices_;
    bool negative_ = false;
};

}  // namespace pcc
