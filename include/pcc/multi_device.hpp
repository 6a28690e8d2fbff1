// multi_device.hpp -- the host side of SURVEY.md 8e in C++: one index per GPU, queries sharded, no collective in
// the search.  The reference is single-threaded, single-GPU code (src/comparator.cpp:571-577 walks its queries one
// by one against one tree); on a node with several MI355X the same loop is spread like this:
//
//   pcc::ShardedKdTree<PointT> tree({0, 1, 2, 3});      // or pcc::allDevices()
//   tree.setInputCloud(cloud);                          // ONE host upload; the other devices get the packed cloud
//                                                       // device-to-device, all copies at once (pcc_index_clone_to_devices)
//   tree.nearestKSearchBatch(queries, idx, d2);         // contiguous query shards, one worker thread per device
//
// and independent clouds (the CLI has two: src/comparator.cpp:1191-1197, 1520-1549) run as REPLICAS, one device
// each: pcc::onDevices(2, devices, [&](int k) { segment(cloud[k]); }).  Header-only over include/pcc_nn.h.
#pragma once
#include <exception>
#include <thread>
#include <vector>
#include "pcc/search.hpp"

namespace pcc {

inline int deviceCount() {
    int n = 0;
    return pcc_device_count(&n) == PCC_OK ? n : 0;
}
inline std::vector<int> allDevices() {
    std::vector<int> d(deviceCount() > 0 ? deviceCount() : 0);
    for (size_t k = 0; k < d.size(); ++k) d[k] = (int)k;
    return d;
}
// contiguous shard r of n items split over `parts` (the first n % parts shards hold one item more); the same
// split pointcloudcomparator_amd/sharding.py and bench.py use
inline void shardRange(size_t n, size_t r, size_t parts, size_t& start, size_t& count) {
    const size_t base = n / parts, rem = n % parts;
    start = r * base + (r < rem ? r : rem);
    count = base + (r < rem ? 1 : 0);
}

// fn(k) for k = 0 .. jobs-1, job k on its own thread whose default device (threadDevice()) is devices[k % size]:
// every pcc:: object the job constructs lives on that device.  Exceptions are carried back to the caller.
template <class F>
inline void onDevices(int jobs, const std::vector<int>& devices, F fn) {
    std::vector<std::exception_ptr> err((size_t)jobs);
    std::vector<std::thread> th;
    for (int k = 0; k < jobs; ++k)
        th.emplace_back([&, k]() {
            threadDevice() = devices.empty() ? 0 : devices[(size_t)k % devices.size()];
            try { fn(k); } catch (...) { err[(size_t)k] = std::current_exception(); }
        });
    for (std::thread& t : th) t.join();
    for (std::exception_ptr& e : err)
        if (e) std::rethrow_exception(e);
}

// One search index per device over the same cloud; k = 1 / k-NN batches are sharded over them.
template <class PointT>
class ShardedKdTree {
public:
    typedef typename PointCloud<PointT>::ConstPtr CloudConstPtr;
    explicit ShardedKdTree(const std::vector<int>& devices = std::vector<int>(1, 0), int engine = PCC_ENGINE_AUTO)
        : devices_(devices.empty() ? std::vector<int>(1, 0) : devices), engine_(engine) {}
    ~ShardedKdTree() { release(); }
    ShardedKdTree(const ShardedKdTree&) = delete;
    ShardedKdTree& operator=(const ShardedKdTree&) = delete;

    void setInputCloud(const CloudConstPtr& cloud) {
        release();
        input_ = cloud;
        if (!cloud || cloud->empty()) return;
        pcc_index* first = nullptr;
        int st = pcc_index_create(cloud->points.data(), cloud->size(), sizeof(PointT), 3, PCC_MEM_HOST, devices_[0], engine_, &first);
        if (st == PCC_ERR_EMPTY) return;  // PCL: "Cannot create a KDTree with an empty input cloud"
        check(st);
        handles_.push_back(first);
        if (devices_.size() > 1) {  // the other devices: every peer copy in flight at once, builds overlapped, one join
            std::vector<pcc_index*> more(devices_.size() - 1, nullptr);
            check(pcc_index_clone_to_devices(first, devices_.data() + 1, (int)more.size(), more.data()));
            handles_.insert(handles_.end(), more.begin(), more.end());
        }
    }
    // Shard k's part of a batch whose queries already live in the HBM of device k (produced there, or uploaded once):
    // nothing crosses PCIe.  Asynchronous on the shard's stream; syncShards() waits for all of them.
    void nearestKSearchShardDevice(size_t shard, const void* queries_dev, size_t nq, size_t stride_bytes, int* idx_dev, float* d2_dev) const {
        check(pcc_nn1(handles_.at(shard), queries_dev, nq, stride_bytes, PCC_MEM_DEVICE, idx_dev, d2_dev));
    }
    void syncShards() const {
        for (pcc_index* h : handles_) check(pcc_index_sync(h));
    }
    size_t shards() const { return handles_.size(); }
    pcc_index* handle(size_t k) const { return handles_.at(k); }

    // idx / d2 of the nearest reference of every query (k = 1), in query order
    void nearestKSearchBatch(const PointCloud<PointT>& queries, std::vector<int>& idx, std::vector<float>& d2) const {
        idx.assign(queries.size(), -1);
        d2.assign(queries.size(), 0.f);
        if (handles_.empty() || queries.empty()) return;
        run(queries.size(), [&](size_t k, size_t s, size_t c) {
            check(pcc_nn1(handles_[k], queries.points.data() + s, c, sizeof(PointT), PCC_MEM_HOST, idx.data() + s, d2.data() + s));
        });
    }
    void nearestKSearchBatch(const PointCloud<PointT>& queries, int k_nn, std::vector<int>& idx, std::vector<float>& d2) const {
        idx.assign(queries.size() * (size_t)k_nn, -1);
        d2.assign(queries.size() * (size_t)k_nn, 0.f);
        if (handles_.empty() || queries.empty()) return;
        run(queries.size(), [&](size_t k, size_t s, size_t c) {
            check(pcc_knn(handles_[k], queries.points.data() + s, c, sizeof(PointT), PCC_MEM_HOST, k_nn,
                          idx.data() + s * (size_t)k_nn, d2.data() + s * (size_t)k_nn));
        });
    }

private:
    template <class F>
    void run(size_t n, F shard_fn) const {
        const size_t parts = handles_.size();
        std::vector<std::exception_ptr> err(parts);
        std::vector<std::thread> th;
        for (size_t k = 0; k < parts; ++k) {
            size_t s, c;
            shardRange(n, k, parts, s, c);
            if (c == 0) continue;
            th.emplace_back([&, k, s, c]() {
                try { shard_fn(k, s, c); } catch (...) { err[k] = std::current_exception(); }
            });
        }
        for (std::thread& t : th) t.join();
        for (std::exception_ptr& e : err)
            if (e) std::rethrow_exception(e);
    }
    void release() {
        for (pcc_index* h : handles_) pcc_index_destroy(h);
        handles_.clear();
    }
    std::vector<int> devices_;
    int engine_;
    CloudConstPtr input_;
    std::vector<pcc_index*> handles_;
};

}  // namespace pcc
