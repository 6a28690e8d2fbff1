// flann_order.hpp -- which of several EQUALLY NEAR references pcl::KdTreeFLANN would have returned.
//
// libpcc_nn resolves exact-distance ties to the lowest original index.  FLANN's KDTreeSingleIndex returns the
// tied point its tree walk reaches FIRST (KNNSimpleResultSet::addPoint rejects `dist >= worst`), which depends on
// the tree it built.  matchRIFTFeaturesKnn hands those indices to its caller (reference src/comparator.cpp:576-580),
// so PCC_TIES_FLANN mode reproduces them: the GPU search stays as it is, queries for which a second reference
// shares the minimum distance are flagged (k_tie_flags), and only those are walked through this host-side
// restatement of FLANN 1.8's single kd-tree -- build (divideTree / middleSplit / planeSplit, leaf size 15, data
// reordered into leaf order) and exact k = 1 search (computeInitialDistances, searchLevel: near child first, far
// child when its bound does not exceed the current worst distance) -- as SURVEY.md 9.1-9.2 records them.
// Distances are FLANN's L2_Simple<float>: ((dx*dx) + dy*dy) + dz*dz, every operation rounded to float; this file is
// compiled with -ffp-contract=off like the kernels.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace pcc {

class FlannOrder {
public:
    // packed[i] = (x, y, z, w) with w's sign bit set for non-finite points (they are not indexed, PCL's
    // convertCloudToArray skips them); indices reported are positions in `packed`
    void build(const float* packed_xyzw, size_t n);
    bool empty() const { return n_ == 0; }
    // index FLANN's findNeighbors(k = 1) returns for q, and its squared distance; -1 for an empty tree
    int32_t nearest(const float q[3], float* d2) const;

private:
    struct Interval { float low, high; };
    struct Node {
        int32_t child1 = -1, child2 = -1;  // inner node: indices into nodes_
        int32_t left = 0, right = 0;       // leaf: [left, right) of the reordered data
        int32_t divfeat = 0;
        float divlow = 0.f, divhigh = 0.f;
    };
    int32_t divide(int32_t left, int32_t right, Interval bbox[3]);
    void minmax(const int32_t* ind, int32_t count, int dim, float& mn, float& mx) const;
    void planeSplit(int32_t* ind, int32_t count, int cutfeat, float cutval, int32_t& lim1, int32_t& lim2) const;
    void middleSplit(int32_t* ind, int32_t count, int32_t& index, int& cutfeat, float& cutval, const Interval bbox[3]) const;
    void searchLevel(const float q[3], int32_t node, float mindistsq, float dists[3], float& worst, int32_t& best) const;

    size_t n_ = 0;
    std::vector<float> pts_;       // dense valid points (PCL order), 3 floats each
    std::vector<int32_t> map_;     // dense -> position in the packed cloud (index_mapping_)
    std::vector<int32_t> vind_;    // FLANN vind_
    std::vector<float> data_;      // points in leaf order (reorder = true)
    std::vector<Node> nodes_;
    int32_t root_ = -1;
    Interval root_bbox_[3];
};

}  // namespace pcc
