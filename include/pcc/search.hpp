// search.hpp -- pcc::search::KdTree<PointT> / pcc::KdTreeFLANN<PointT>: the PCL-shaped search
// object the reference constructs (src/comparator.cpp:564-577 pcl::KdTreeFLANN<RIFT32>;
// src/segmentation.cpp:120-122,129 pcl::search::KdTree<pcl::PointXYZRGB> injected with
// setSearchMethod).  Same method names, argument meaning and return conventions
// (SURVEY.md 8b): returns the number of neighbours found, distances are squared, indices
// refer to the original cloud, k is clamped to the number of valid points.  Header-only on
// top of the C-ABI (include/pcc_nn.h); the per-point methods are batches of one for
// compatibility, callers that care about speed use the *Batch methods.
#pragma once
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "pcc_nn.h"
#include "pcc/point_types.hpp"

namespace pcc {

struct Error : std::runtime_error {
    int status;
    Error(int s, const char* what) : std::runtime_error(std::string("libpcc_nn: ") + what), status(s) {}
};
inline void check(int status) {
    if (status != PCC_OK) throw Error(status, pcc_last_error());
}

// Device new search objects of THIS thread are created on (default 0).  The reference's call sites construct their
// trees without saying where (src/comparator.cpp:564, src/segmentation.cpp:120); a host that spreads work over
// the GPUs of a node gives each worker thread its device once (pcc/multi_device.hpp) instead of threading a
// device argument through every object.
inline int& threadDevice() {
    static thread_local int device = 0;
    return device;
}

namespace search {

template <class PointT>
class KdTree {
public:
    typedef typename PointCloud<PointT>::ConstPtr CloudConstPtr;
    typedef std::shared_ptr<KdTree<PointT>> Ptr;

    // pcl::KdTreeFLANN(bool sorted = true); the reference writes `= new KdTreeFLANN<T>(false)`,
    // a pointer-to-bool conversion that yields sorted == TRUE
    // device < 0: the calling thread's default device (threadDevice())
    explicit KdTree(bool sorted = true, int device = -1, int engine = PCC_ENGINE_AUTO)
        : sorted_(sorted), device_(device < 0 ? threadDevice() : device), engine_(engine) {}
    int device() const { return device_; }
    ~KdTree() { if (index_) pcc_index_destroy(index_); }
    KdTree(const KdTree&) = delete;
    KdTree& operator=(const KdTree&) = delete;

    // PCL prints "Cannot create a KDTree with an empty input cloud" and returns; so do we.
    // A tree object can be pointed at a new cloud any number of times (PCL: cleanup() + rebuild);
    // the device handle, its stream and its buffers are kept and only the index is rebuilt, so a
    // long-lived tree (see matchRIFTFeaturesKnn) costs no allocation per call.
    typedef std::shared_ptr<const std::vector<int>> IndicesConstPtr;

    // pcl::search::Search<PointT>::setInputCloud(cloud, indices): with a non-empty index list
    // the tree holds only cloud[indices[j]]; results still name points of `cloud` (PCL maps them back through
    // index_mapping_ = indices), and nearestKSearch(int index, ...) takes its query from cloud[indices[index]].
    void setInputCloud(const CloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) {
        input_ = cloud;
        indices_ = indices;
        valid_ = false;
        subset_.clear();
        if (!cloud || cloud->empty()) return;
        const PointT* pts = cloud->points.data();
        size_t n = cloud->size();
        if (indices_ && !indices_->empty()) {
            subset_.reserve(indices_->size());
            for (int j : *indices_) subset_.push_back(cloud->points.at((size_t)j));
            pts = subset_.data();
            n = subset_.size();
        }
        if (!index_) {
            int st = pcc_index_create(pts, n, sizeof(PointT), 3, PCC_MEM_HOST, device_, engine_, &index_);
            if (st == PCC_ERR_EMPTY) { index_ = nullptr; return; }
            check(st);
            if (ties_ != PCC_TIES_LOWEST_INDEX) check(pcc_index_set_tie_order(index_, ties_));
            for (const auto& o : options_) check(pcc_index_set_option(index_, o.first, o.second));
            valid_ = true;
            return;
        }
        check(pcc_index_set_input(index_, pts, n, sizeof(PointT), 3, PCC_MEM_HOST));
        size_t n_valid = 0;
        check(pcc_index_size(index_, &n_valid));  // waits for the build; 0 == PCL's "empty input cloud"
        valid_ = n_valid > 0;
    }
    // PCC_TIES_FLANN: k = 1 results name, among equally near points, the one PCL's FLANN tree reaches first
    // (default: the lowest index); kept across setInputCloud calls
    void setTieOrder(int ties) {
        ties_ = ties;
        if (index_) check(pcc_index_set_tie_order(index_, ties_));
    }
    // a per-handle implementation choice (enum pcc_option, include/pcc_nn.h); none changes a result bit.  E.g.
    // setOption(PCC_OPT_KNN_CACHE_K, 100) before NormalEstimation(50) + RegionGrowing(100) on this tree: one search, not two
    void setOption(int option, double value) {  // (kept and applied to the handle once a cloud has created it)
        for (auto& o : options_) if (o.first == option) { o.second = value; option = -1; break; }
        if (option >= 0) options_.emplace_back(option, value);
        if (index_) for (const auto& o : options_) check(pcc_index_set_option(index_, o.first, o.second));
    }
    CloudConstPtr getInputCloud() const { return input_; }
    IndicesConstPtr getIndices() const { return indices_; }
    bool getSortedResults() const { return sorted_; }
    pcc_index* handle() const { return valid_ ? index_ : nullptr; }
    // the device index holds every point of `cloud` (not a subset selected by an index list)
    bool covers(const CloudConstPtr& cloud) const { return valid_ && input_ == cloud && subset_.empty(); }

    int nearestKSearch(const PointT& p, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const {
        if (!handle() || !isFinite(p)) { k_indices.clear(); k_sqr_distances.clear(); return 0; }  // PCL asserts here
        size_t total = 0;
        check(pcc_index_size(index_, &total));
        if ((size_t)k > total) k = (int)total;
        k_indices.resize(k);
        k_sqr_distances.resize(k);
        if (k == 0) return 0;
        if (k == 1) check(pcc_nn1(index_, &p, 1, sizeof(PointT), PCC_MEM_HOST, k_indices.data(), k_sqr_distances.data()));
        else check(pcc_knn(index_, &p, 1, sizeof(PointT), PCC_MEM_HOST, k, k_indices.data(), k_sqr_distances.data()));
        remap(k_indices);
        return k;
    }
    int nearestKSearch(int index, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const {
        const size_t at = subset_.empty() ? (size_t)index : (size_t)indices_->at((size_t)index);
        return nearestKSearch(input_->points.at(at), k, k_indices, k_sqr_distances);
    }
    int radiusSearch(const PointT& p, double radius, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances,
                     unsigned int max_nn = 0) const {
        k_indices.clear();
        k_sqr_distances.clear();
        if (!handle() || !isFinite(p)) return 0;
        int32_t cnt = 0;
        // (max_nn != 0: FLANN keeps the max_nn nearest within the radius, ascending -- the _max entry points)
        check(pcc_radius_count_max(index_, &p, 1, sizeof(PointT), PCC_MEM_HOST, radius, max_nn, &cnt));
        if (cnt == 0) return 0;
        int64_t offs[2] = {0, cnt};
        k_indices.resize(cnt);
        k_sqr_distances.resize(cnt);
        check(pcc_radius_fill_max(index_, &p, 1, sizeof(PointT), PCC_MEM_HOST, radius, sorted_ ? 1 : 0, max_nn, offs,
                                  k_indices.data(), k_sqr_distances.data()));
        remap(k_indices);
        return cnt;
    }

    // ---- batch forms: what a GPU wants (one call for the whole query cloud) ----
    void nearestKSearchBatch(const PointCloud<PointT>& queries, std::vector<int>& idx, std::vector<float>& d2) const {
        idx.assign(queries.size(), -1);
        d2.assign(queries.size(), 0.f);
        if (!handle() || queries.empty()) return;
        check(pcc_nn1(index_, queries.points.data(), queries.size(), sizeof(PointT), PCC_MEM_HOST, idx.data(), d2.data()));
        remap(idx);
    }
    void nearestKSearchBatch(const PointCloud<PointT>& queries, int k, std::vector<int>& idx, std::vector<float>& d2) const {
        idx.assign(queries.size() * (size_t)k, -1);
        d2.assign(queries.size() * (size_t)k, 0.f);
        if (!handle() || queries.empty()) return;
        check(pcc_knn(index_, queries.points.data(), queries.size(), sizeof(PointT), PCC_MEM_HOST, k, idx.data(), d2.data()));
        remap(idx);
    }

private:
    // positions in the indexed subset -> indices of the input cloud (identity without an index list)
    void remap(std::vector<int>& idx) const {
        if (subset_.empty()) return;
        for (int& v : idx)
            if (v >= 0) v = (*indices_)[(size_t)v];
    }
    IndicesConstPtr indices_;
    std::vector<PointT> subset_;  // cloud[indices]: what the device index was built from
    bool sorted_;
    int ties_ = PCC_TIES_LOWEST_INDEX;
    int device_, engine_;
    CloudConstPtr input_;
    pcc_index* index_ = nullptr;
    std::vector<std::pair<int, double>> options_;
    bool valid_ = false;  // index_ holds at least one finite point of input_
};

}  // namespace search

template <class PointT>
using KdTreeFLANN = search::KdTree<PointT>;

}  // namespace pcc
