// flann_order.hip -- host-side restatement of FLANN 1.8 KDTreeSingleIndex (build + exact k = 1 search) used to
// resolve exact-distance ties the way pcl::KdTreeFLANN does (see flann_order.hpp), and the kernels that flag the
// tied queries.  Host code here is compiled by hipcc with -ffp-contract=off: the distances must round like
// FLANN's L2_Simple<float>.
#include "flann_order.hpp"
#include "pcc_internal.hpp"
#include "grid_device.hpp"
#include <cfloat>
#include <cstring>

namespace pcc {

static constexpr int32_t FLANN_LEAF_MAX = 15;  // KDTreeSingleIndexParams(15), pcl::KdTreeFLANN::setInputCloud

static inline float l2_simple(const float* a, const float* b) {
    float r = 0.f;
    for (int i = 0; i < 3; ++i) { const float d = a[i] - b[i]; r += d * d; }
    return r;
}
static inline bool is_valid_w(float w) {
    uint32_t u;
    memcpy(&u, &w, 4);
    return (u >> 31) == 0;
}

void FlannOrder::minmax(const int32_t* ind, int32_t count, int dim, float& mn, float& mx) const {
    mn = mx = pts_[(size_t)ind[0] * 3 + dim];
    for (int32_t i = 1; i < count; ++i) {
        const float v = pts_[(size_t)ind[i] * 3 + dim];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
}

// indices rearranged into  < cutval | == cutval | > cutval ; lim1 / lim2 are the two boundaries
void FlannOrder::planeSplit(int32_t* ind, int32_t count, int cutfeat, float cutval, int32_t& lim1, int32_t& lim2) const {
    auto at = [&](int32_t k) { return pts_[(size_t)ind[k] * 3 + cutfeat]; };
    int32_t lo = 0, hi = count - 1;
    for (;;) {
        while (lo <= hi && at(lo) < cutval) ++lo;
        while (lo <= hi && at(hi) >= cutval) --hi;
        if (lo > hi) break;
        const int32_t t = ind[lo]; ind[lo] = ind[hi]; ind[hi] = t;
        ++lo; --hi;
    }
    lim1 = lo;
    hi = count - 1;
    for (;;) {
        while (lo <= hi && at(lo) <= cutval) ++lo;
        while (lo <= hi && at(hi) > cutval) --hi;
        if (lo > hi) break;
        const int32_t t = ind[lo]; ind[lo] = ind[hi]; ind[hi] = t;
        ++lo; --hi;
    }
    lim2 = lo;
}

// split dimension: the widest side of the (approximate) box, corrected by the exact spread of the points; the cut
// is the middle of the exact range; the split position is the middle of the run of points equal to the cut when
// that run straddles count / 2, else its nearer end
void FlannOrder::middleSplit(int32_t* ind, int32_t count, int32_t& index, int& cutfeat, float& cutval, const Interval bbox[3]) const {
    float max_span = bbox[0].high - bbox[0].low;
    cutfeat = 0;
    for (int i = 1; i < 3; ++i) {
        const float span = bbox[i].high - bbox[i].low;
        if (span > max_span) { max_span = span; cutfeat = i; }
    }
    float mn, mx;
    minmax(ind, count, cutfeat, mn, mx);
    cutval = (mn + mx) / 2;
    max_span = mx - mn;
    const int first = cutfeat;
    for (int i = 0; i < 3; ++i) {
        if (i == first) continue;
        if (bbox[i].high - bbox[i].low > max_span) {
            minmax(ind, count, i, mn, mx);
            if (mx - mn > max_span) { max_span = mx - mn; cutfeat = i; cutval = (mn + mx) / 2; }
        }
    }
    int32_t lim1, lim2;
    planeSplit(ind, count, cutfeat, cutval, lim1, lim2);
    if (lim1 > count / 2) index = lim1;
    else if (lim2 < count / 2) index = lim2;
    else index = count / 2;
}

int32_t FlannOrder::divide(int32_t left, int32_t right, Interval bbox[3]) {
    const int32_t me = (int32_t)nodes_.size();
    nodes_.push_back(Node());
    if (right - left <= FLANN_LEAF_MAX) {
        nodes_[me].left = left;
        nodes_[me].right = right;
        for (int d = 0; d < 3; ++d) bbox[d].low = bbox[d].high = pts_[(size_t)vind_[left] * 3 + d];
        for (int32_t k = left + 1; k < right; ++k)
            for (int d = 0; d < 3; ++d) {
                const float v = pts_[(size_t)vind_[k] * 3 + d];
                if (v < bbox[d].low) bbox[d].low = v;
                if (v > bbox[d].high) bbox[d].high = v;
            }
        return me;
    }
    int32_t idx;
    int cutfeat;
    float cutval;
    middleSplit(vind_.data() + left, right - left, idx, cutfeat, cutval, bbox);
    Interval lb[3], rb[3];
    memcpy(lb, bbox, sizeof(lb));
    memcpy(rb, bbox, sizeof(rb));
    lb[cutfeat].high = cutval;
    const int32_t c1 = divide(left, left + idx, lb);
    rb[cutfeat].low = cutval;
    const int32_t c2 = divide(left + idx, right, rb);
    Node& nd = nodes_[me];  // (after the recursion: the vector may have moved)
    nd.child1 = c1;
    nd.child2 = c2;
    nd.divfeat = cutfeat;
    nd.divlow = lb[cutfeat].high;   // the children have tightened their boxes
    nd.divhigh = rb[cutfeat].low;
    for (int d = 0; d < 3; ++d) {
        bbox[d].low = lb[d].low < rb[d].low ? lb[d].low : rb[d].low;
        bbox[d].high = lb[d].high > rb[d].high ? lb[d].high : rb[d].high;
    }
    return me;
}

void FlannOrder::build(const float* packed, size_t n) {
    n_ = 0;
    pts_.clear(); map_.clear(); vind_.clear(); data_.clear(); nodes_.clear();
    root_ = -1;
    for (size_t i = 0; i < n; ++i) {
        const float* p = packed + i * 4;
        if (!is_valid_w(p[3])) continue;  // convertCloudToArray: invalid points are skipped, order kept
        pts_.push_back(p[0]); pts_.push_back(p[1]); pts_.push_back(p[2]);
        map_.push_back((int32_t)i);
    }
    n_ = map_.size();
    if (n_ == 0) return;
    vind_.resize(n_);
    for (size_t i = 0; i < n_; ++i) vind_[i] = (int32_t)i;
    for (int d = 0; d < 3; ++d) minmax(vind_.data(), (int32_t)n_, d, root_bbox_[d].low, root_bbox_[d].high);
    Interval bb[3];
    memcpy(bb, root_bbox_, sizeof(bb));
    nodes_.reserve(n_ / 4 + 16);
    root_ = divide(0, (int32_t)n_, bb);
    data_.resize(n_ * 3);
    for (size_t i = 0; i < n_; ++i) memcpy(&data_[i * 3], &pts_[(size_t)vind_[i] * 3], 3 * sizeof(float));
}

void FlannOrder::searchLevel(const float q[3], int32_t ni, float mindistsq, float dists[3], float& worst, int32_t& best) const {
    const Node& nd = nodes_[ni];
    if (nd.child1 < 0) {
        for (int32_t i = nd.left; i < nd.right; ++i) {
            const float d = l2_simple(q, &data_[(size_t)i * 3]);
            if (d < worst) { worst = d; best = vind_[i]; }  // k = 1: an equal distance never displaces the earlier point
        }
        return;
    }
    const int f = nd.divfeat;
    const float val = q[f];
    const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
    int32_t near_child, far_child;
    float cut;
    if (diff1 + diff2 < 0) { near_child = nd.child1; far_child = nd.child2; cut = (val - nd.divhigh) * (val - nd.divhigh); }
    else { near_child = nd.child2; far_child = nd.child1; cut = (val - nd.divlow) * (val - nd.divlow); }
    searchLevel(q, near_child, mindistsq, dists, worst, best);
    const float saved = dists[f];
    mindistsq = mindistsq + cut - saved;
    dists[f] = cut;
    if (mindistsq <= worst) searchLevel(q, far_child, mindistsq, dists, worst, best);  // epsError = 1 (exact search)
    dists[f] = saved;
}

int32_t FlannOrder::nearest(const float q[3], float* d2) const {
    if (n_ == 0) { if (d2) *d2 = FLT_MAX; return -1; }
    float dists[3] = {0.f, 0.f, 0.f};
    float distsq = 0.f;
    for (int i = 0; i < 3; ++i) {  // computeInitialDistances: the part of the query outside the root box
        if (q[i] < root_bbox_[i].low) { dists[i] = (q[i] - root_bbox_[i].low) * (q[i] - root_bbox_[i].low); distsq += dists[i]; }
        if (q[i] > root_bbox_[i].high) { dists[i] = (q[i] - root_bbox_[i].high) * (q[i] - root_bbox_[i].high); distsq += dists[i]; }
    }
    float worst = FLT_MAX;
    int32_t best = -1;
    searchLevel(q, root_, distsq, dists, worst, best);
    if (d2) *d2 = worst;
    return best < 0 ? -1 : map_[(size_t)best];
}

// ---- which queries have a tie ------------------------------------------------------------------------------
// keys[i] = (d2 bits << 32 | index) of the nearest reference.  flags[i] = 1 when some OTHER reference has exactly
// the same d2 (or when that cannot be decided cheaply: ball wider than the cell walk allows).
__global__ void __launch_bounds__(256)
k_tie_flags_grid(const float4* __restrict__ cell_refs, const unsigned int* __restrict__ cell_start,
                 const GridDev* __restrict__ gd, const float4* __restrict__ q, const unsigned long long* __restrict__ keys,
                 unsigned int n, uint8_t* __restrict__ flags) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GridParams g = gd->g;
    const float4 qv = q[i];
    const unsigned long long key = keys[i];
    if (__float_as_int(qv.w) < 0 || key_none(key)) { flags[i] = 0; return; }
    const float bd = __uint_as_float((unsigned int)(key >> 32));
    const unsigned int bi = (unsigned int)key;
    const float rb = sqrtf(bd) * 1.00001f + gd->slack;
    int x0, x1, y0, y1, z0, z1;
    cell_range(qv.x, rb, g.org[0], g.inv_h, g.dim[0], x0, x1);
    cell_range(qv.y, rb, g.org[1], g.inv_h, g.dim[1], y0, y1);
    cell_range(qv.z, rb, g.org[2], g.inv_h, g.dim[2], z0, z1);
    if (!(rb < __builtin_inff()) || (long long)(x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1) > 4096) { flags[i] = 1; return; }
    bool tie = false;
    for (int z = z0; z <= z1 && !tie; ++z)
        for (int y = y0; y <= y1 && !tie; ++y) {
            const unsigned int row = ((unsigned int)z * g.dim[1] + y) * g.dim[0];
            const unsigned int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
            for (unsigned int p = s; p < e; ++p) {
                const float4 r = cell_refs[p];
                if (dist2(qv.x, qv.y, qv.z, r) == bd && (unsigned int)__float_as_int(r.w) != bi) { tie = true; break; }
            }
        }
    flags[i] = tie ? 1 : 0;
}

// exhaustive form (BRUTE engine: small clouds): one lane per query over all references
__global__ void __launch_bounds__(256)
k_tie_flags_brute(const float4* __restrict__ refs, unsigned int m, const float4* __restrict__ q,
                  const unsigned long long* __restrict__ keys, unsigned int n, uint8_t* __restrict__ flags) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 qv = q[i];
    const unsigned long long key = keys[i];
    if (__float_as_int(qv.w) < 0 || key_none(key)) { flags[i] = 0; return; }
    const float bd = __uint_as_float((unsigned int)(key >> 32));
    const unsigned int bi = (unsigned int)key;
    bool tie = false;
    for (unsigned int p = 0; p < m; ++p) {  // wave-uniform address: one broadcast load per reference
        const float4 r = refs[p];
        if (__float_as_int(r.w) >= 0 && p != bi && dist2(qv.x, qv.y, qv.z, r) == bd) tie = true;
    }
    flags[i] = tie ? 1 : 0;
}

int launch_tie_flags(pcc_index* ix, const float4* q, const unsigned long long* keys, size_t nq, uint8_t* flags) {
    if (nq == 0) return PCC_OK;
    const unsigned int n = (unsigned int)nq, blocks = (n + 255) / 256;
    if (ix->engine == PCC_ENGINE_GRID && ix->has_grid)
        hipLaunchKernelGGL(k_tie_flags_grid, dim3(blocks), dim3(256), 0, ix->stream, ix->cell_refs.as<float4>(),
                           ix->cell_start.as<unsigned int>(), ix->d_grid.as<GridDev>(), q, keys, n, flags);
    else
        hipLaunchKernelGGL(k_tie_flags_brute, dim3(blocks), dim3(256), 0, ix->stream, ix->refs.as<float4>(),
                           (unsigned int)ix->n_orig, q, keys, n, flags);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

}  // namespace pcc
