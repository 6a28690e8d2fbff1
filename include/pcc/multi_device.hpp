// multi_device.hpp -- the host side of SURVEY.md 8e in C++: one index per GPU, queries sharded, no collective in
// the search.  The reference is single-threaded, single-GPU code (src/comparator.cpp:571-577 walks its queries one
// by one against one tree); on a node with several MI355X the same loop is spread like this:
//
//   pcc::ShardedKdTree<PointT> tree({0, 1, 2, 3});      // or pcc::allDevices()
//   tree.setInputCloud(cloud);                          // ONE host upload; the other devices get the packed cloud
//                                                       // device-to-device, all copies at once (pcc_index_clone_to_devices)
//   tree.nearestKSearchBatch(queries, idx, d2);         // contiguous query shards, one worker thread per device
//   tree.icpAlign(source, 20, T, fitness, it, conv);    // ICP with the SOURCE sharded: 17 sums all-reduced per pass (RCCL)
//   tree.sor(50, 1.5, mean, inlier, thr, kept);         // the -n noise pass with the cloud's points sharded
//
// With more than one device the cloud travels by ONE RCCL broadcast over xGMI (pcc_index_create_broadcast: one rank per
// device, a worker thread each) and ICP / SOR use RCCL all-reduces on the handles' streams; should RCCL be missing the
// tree falls back to peer copies (pcc_index_clone_to_devices) and offers the searches only.
//
// and independent clouds (the CLI has two: src/comparator.cpp:1191-1197, 1520-1549) run as REPLICAS, one device
// each: pcc::onDevices(2, devices, [&](int k) { segment(cloud[k]); }).  Header-only over include/pcc_nn.h.
#pragma once
#include <array>
#include <cstdint>
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
        : devices_(devices.empty() ? std::vector<int>(1, 0) : devices), engine_(engine), use_rccl_(devices_.size() > 1) {}
    // RCCL communicators also for a single device (tests on a one-GPU box; otherwise they are made from two devices on)
    void setUseRccl(bool on) { use_rccl_ = on; }
    bool hasCollectives() const { return !comms_.empty(); }
    ~ShardedKdTree() { release(); releaseComms(); }
    ShardedKdTree(const ShardedKdTree&) = delete;
    ShardedKdTree& operator=(const ShardedKdTree&) = delete;

    void setInputCloud(const CloudConstPtr& cloud) {
        release();
        input_ = cloud;
        if (!cloud || cloud->empty()) return;
        if (use_rccl_ && comms_.empty()) {  // one rank per device, made once
            comms_.assign(devices_.size(), nullptr);
            if (pcc_comm_create_local(devices_.data(), (int)devices_.size(), comms_.data()) != PCC_OK) comms_.clear();
        }
        if (!comms_.empty()) {
            // rank 0 indexes the host cloud; the packed copy goes to every other rank in one broadcast (one thread per rank:
            // a collective returns when all ranks have joined), each rank builds its own index
            handles_.assign(devices_.size(), nullptr);
            std::vector<int> st(devices_.size(), PCC_OK);
            std::vector<std::string> msg(devices_.size());
            std::vector<std::thread> th;
            for (size_t k = 0; k < devices_.size(); ++k)
                th.emplace_back([&, k]() {
                    st[k] = pcc_index_create_broadcast(comms_[k], 0, k == 0 ? cloud->points.data() : nullptr, k == 0 ? cloud->size() : 0,
                                                       sizeof(PointT), PCC_MEM_HOST, engine_, &handles_[k], nullptr);
                    if (st[k] != PCC_OK) msg[k] = pcc_last_error();
                });
            for (std::thread& t : th) t.join();
            for (size_t k = 0; k < st.size(); ++k)
                if (st[k] != PCC_OK) {
                    release();
                    if (st[k] == PCC_ERR_EMPTY) return;  // PCL: "Cannot create a KDTree with an empty input cloud"
                    throw std::runtime_error("libpcc_nn: " + msg[k]);
                }
            return;
        }
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

    // pcl::IterativeClosestPoint::align against this tree's cloud with the SOURCE sharded over the devices (reference
    // src/comparator.cpp:1089-1099): every pass all-reduces the 17 sums over RCCL, all ranks apply the same transform.
    // T row-major; fitness = getFitnessScore() over the whole source.  Needs the communicators (hasCollectives()).
    void icpAlign(const PointCloud<PointT>& source, int max_iterations, std::array<float, 16>& T, double& fitness, int& iterations,
                  bool& converged) const {
        if (comms_.empty() || handles_.empty()) throw std::runtime_error("ShardedKdTree::icpAlign needs RCCL communicators and an indexed cloud");
        if (source.size() < handles_.size()) throw std::runtime_error("ShardedKdTree::icpAlign: fewer source points than devices");
        std::vector<std::array<float, 16>> Ts(handles_.size());
        std::vector<double> fit(handles_.size(), 0.0);
        std::vector<int> its(handles_.size(), 0), conv(handles_.size(), 0);
        run(source.size(), [&](size_t k, size_t s, size_t c) {
            check(pcc_icp_align_sharded(handles_[k], comms_[k], source.points.data() + s, c, sizeof(PointT), PCC_MEM_HOST, max_iterations, 0,
                                        Ts[k].data(), &fit[k], &its[k], &conv[k]));
        });
        T = Ts[0];  // (the same on every rank: same sums, same arithmetic)
        fitness = fit[0];
        iterations = its[0];
        converged = conv[0] != 0;
    }
    // pcl::StatisticalOutlierRemoval over the indexed cloud (src/comparator.cpp:1523-1541) with its points sharded over the
    // devices: mean distances and inlier flags in cloud order, PCL's threshold over the whole cloud
    void sor(int mean_k, double stddev_mult, std::vector<float>& mean_dist, std::vector<uint8_t>& inlier, double& threshold, size_t& kept) const {
        if (comms_.empty() || handles_.empty()) throw std::runtime_error("ShardedKdTree::sor needs RCCL communicators and an indexed cloud");
        const size_t n = input_->size();
        mean_dist.assign(n, 0.f);
        inlier.assign(n, 0);
        std::vector<double> thr(handles_.size(), 0.0);
        std::vector<size_t> kp(handles_.size(), 0);
        run(n, [&](size_t k, size_t s, size_t c) {
            check(pcc_sor_sharded(handles_[k], comms_[k], s, c, mean_k, stddev_mult, PCC_MEM_HOST, mean_dist.data() + s, inlier.data() + s, &thr[k], &kp[k]));
        }, /*every_rank=*/true);
        threshold = thr[0];
        kept = kp[0];
    }

private:
    // shard_fn(k, start, count) on one thread per device; every_rank: also for empty shards (collectives need all ranks)
    template <class F>
    void run(size_t n, F shard_fn, bool every_rank = false) const {
        const size_t parts = handles_.size();
        std::vector<std::exception_ptr> err(parts);
        std::vector<std::thread> th;
        for (size_t k = 0; k < parts; ++k) {
            size_t s, c;
            shardRange(n, k, parts, s, c);
            if (c == 0 && !every_rank) continue;
            th.emplace_back([&, k, s, c]() {
                try { shard_fn(k, s, c); } catch (...) { err[k] = std::current_exception(); }
            });
        }
        for (std::thread& t : th) t.join();
        for (std::exception_ptr& e : err)
            if (e) std::rethrow_exception(e);
    }
    void release() {
        for (pcc_index* h : handles_)
            if (h) pcc_index_destroy(h);
        handles_.clear();
    }
    void releaseComms() {
        for (pcc_comm* c : comms_) pcc_comm_destroy(c);
        comms_.clear();
    }
    std::vector<int> devices_;
    int engine_;
    bool use_rccl_;
    std::vector<pcc_comm*> comms_;
    CloudConstPtr input_;
    std::vector<pcc_index*> handles_;
};

// performICP (src/comparator.cpp:1089-1110) over several devices: the target indexed on each (one broadcast), the source
// sharded, same printed lines as the single-device form in comparator_nn.hpp
inline bool performICP(const PointCloud<PointXYZRGB>::Ptr& point_cloud1, const PointCloud<PointXYZRGB>::Ptr& point_cloud2,
                       const std::vector<int>& devices) {
    ShardedKdTree<PointXYZRGB> tree(devices);
    tree.setInputCloud(point_cloud2);
    if (!tree.hasCollectives() || tree.shards() == 0) return performICP(point_cloud1, point_cloud2);  // (no RCCL: one device)
    std::array<float, 16> T;
    double fitness = 0;
    int iterations = 0;
    bool converged = false;
    tree.icpAlign(*point_cloud1, 20, T, fitness, iterations, converged);
    std::cout << "has converged:" << converged << " ICP fitness score: " << fitness << std::endl;
    if (converged)
        std::cout << "ICP has converged; starting comparison of point clouds" << std::endl;
    else
        std::cout << "ICP has not converged; point clouds too much different to perform a specific comparison" << std::endl;
    return converged;
}

}  // namespace pcc
