// flann_tree.hpp -- which of several EQUALLY NEAR references pcl::KdTreeFLANN would have returned.
//
// libpcc_nn resolves exact-distance ties to the lowest original index.  FLANN's KDTreeSingleIndex returns the tied
// point its tree walk reaches FIRST (KNNSimpleResultSet::addPoint rejects `dist >= worst`), which depends on the tree
// it built.  matchRIFTFeaturesKnn hands those indices to its caller (reference src/comparator.cpp:576-580), so
// PCC_TIES_FLANN mode reproduces them: the GPU search stays as it is, queries for which a second reference shares the
// minimum distance are flagged on the device, and only those are walked through a tree of FLANN's shape.
//
// This file is that tree: plain C++ without any HIP dependency (the ASan build compiles it with g++).
//   build   on the host, once per indexed cloud: FLANN 1.8 KDTreeSingleIndex::divideTree as SURVEY.md 9.2 records it
//           (leaf size 15, data reordered into leaf order), with the split rule selectable (below).  Iterative -- an
//           explicit stack of frames --, nodes in ONE flat array in depth-first order (a node's first child is the next
//           entry, only the second child's position is stored), the top levels forked over threads (sub-ranges of the
//           index array are disjoint; the arrangement inside a range depends only on that range).
//   walk    flann_walk(): exact k = 1 findNeighbors (computeInitialDistances, searchLevel: near child first, far child
//           when its bound does not exceed the worst distance, leaves scanned in stored order with a strict <) over the
//           flat arrays with an explicit stack.  The SAME function runs on the device (k_tie_walk, flann_order.hip) and
//           on the host (tests without a GPU, trees deeper than the device stack).
// Distances are FLANN's L2_Simple<float>: ((dx*dx) + dy*dy) + dz*dz, every operation rounded to float; every file
// that includes this one is compiled with -ffp-contract=off.
//
// UNVERIFIED: PCL / FLANN sources are not in this image.  The tree shape decides only which of several
// equally near points is named, never a distance; which rule FLANN 1.8.4's divideTree calls is a recollection:
//   rule 0  middleSplit_ as recalled from 1.8.4 (the default): cut dimension among the box sides within (1 - 1e-5) of
//           the widest, by the largest spread of the points -- where the selection loop hands `cutfeat`, not its loop
//           variable, to computeMinMax --, cut value = middle of the box side clamped into the points' range
//   rule 1  middleSplit (what SURVEY.md 9.2 restates): widest box side, corrected by the exact spread, cut = middle of
//           the exact range
//   rule 2  middleSplit_ with the loop variable in the selection loop (nanoflann's later form)
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <vector>
#include <thread>

#if defined(__HIPCC__)
#define PCC_FLANN_HD __host__ __device__
#else
#define PCC_FLANN_HD
#endif

namespace pcc {

// inner node: a = position of the second child (the first child is the next entry), b = cut dimension (0..2)
// leaf:       a = first position in leaf order, b = ~count (negative)
struct FlannNode {
    int32_t a, b;
    float divlow, divhigh;
};
struct FlannBox { float lo[3], hi[3]; };

constexpr int FLANN_LEAF_MAX = 15;  // KDTreeSingleIndexParams(15), pcl::KdTreeFLANN::setInputCloud

// Exact k = 1 search.  leaf_pts: 4 floats per point in leaf order, (x, y, z, bits(position in the indexed cloud)).
// STACK: deferred far children; a tree deeper than that cannot be walked here (returns -2: the caller takes another
// route).  Returns the position FLANN's findNeighbors(k = 1) reports (-1: empty tree) and its squared distance.
template <int STACK>
PCC_FLANN_HD inline int32_t flann_walk(const FlannNode* nodes, const float* leaf_pts, const FlannBox& root, size_t n_valid,
                                 float qx, float qy, float qz, float* d2_out) {
    const float FLT_MAX_ = 3.402823466e+38f;
    if (n_valid == 0) { *d2_out = FLT_MAX_; return -1; }
    // (scalars and selects instead of q[f] / dists[f]: an array indexed by the cut dimension lives in scratch memory on
    // the device, three dependent round trips per level)
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;  // FLANN's dists[]
    float mind = 0.f;
    // computeInitialDistances: the part of the query outside the root box
    if (qx < root.lo[0]) { d0 = (qx - root.lo[0]) * (qx - root.lo[0]); mind += d0; }
    if (qx > root.hi[0]) { d0 = (qx - root.hi[0]) * (qx - root.hi[0]); mind += d0; }
    if (qy < root.lo[1]) { d1 = (qy - root.lo[1]) * (qy - root.lo[1]); mind += d1; }
    if (qy > root.hi[1]) { d1 = (qy - root.hi[1]) * (qy - root.hi[1]); mind += d1; }
    if (qz < root.lo[2]) { d2 = (qz - root.lo[2]) * (qz - root.lo[2]); mind += d2; }
    if (qz > root.hi[2]) { d2 = (qz - root.hi[2]) * (qz - root.hi[2]); mind += d2; }
    struct Deferred { int32_t node; float mind, d0, d1, d2; };
    Deferred stack[STACK];
    int sp = 0;
    float worst = FLT_MAX_;
    int32_t best = -1;
    int32_t ni = 0;
    for (;;) {
        const FlannNode nd = nodes[ni];
        if (nd.b < 0) {  // leaf: stored order, an equal distance never displaces the earlier point
            const int32_t first = nd.a, cnt = ~nd.b;
            for (int32_t i = first; i < first + cnt; ++i) {
                const float* p = leaf_pts + (size_t)i * 4;
                const float dx = qx - p[0], dy = qy - p[1], dz = qz - p[2];
                float d = dx * dx;
                d = d + dy * dy;
                d = d + dz * dz;
                if (d < worst) {
                    worst = d;
                    int32_t w;
                    memcpy(&w, p + 3, 4);
                    best = w;
                }
            }
            // back to the most recent deferred far child whose bound still admits it (epsError = 1: exact search)
            bool found = false;
            while (sp > 0) {
                const Deferred e = stack[--sp];
                if (e.mind <= worst) {
                    ni = e.node; mind = e.mind; d0 = e.d0; d1 = e.d1; d2 = e.d2;
                    found = true;
                    break;
                }
            }
            if (!found) break;
            continue;
        }
        const int f = nd.b;
        const float val = f == 0 ? qx : (f == 1 ? qy : qz);
        const float cur = f == 0 ? d0 : (f == 1 ? d1 : d2);
        const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
        int32_t near_child, far_child;
        float cut;
        if (diff1 + diff2 < 0) { near_child = ni + 1; far_child = nd.a; cut = (val - nd.divhigh) * (val - nd.divhigh); }
        else { near_child = nd.a; far_child = ni + 1; cut = (val - nd.divlow) * (val - nd.divlow); }
        // searchLevel recurses into the near child with the distances unchanged, then (on the way back) into the far
        // child with dists[f] = cut and mindistsq + cut - dists[f]: the far visit is deferred with exactly that state,
        // and its bound is tested against the worst distance of THAT moment when it is taken up
        Deferred e;
        e.node = far_child;
        e.mind = mind + cut - cur;
        // (a bound above today's worst distance can never be admitted later -- the worst distance only shrinks --, so
        // such a child is not kept: after the first leaf most of them are dropped here)
        if (e.mind <= worst) {
            if (sp == STACK) { *d2_out = FLT_MAX_; return -2; }
            e.d0 = f == 0 ? cut : d0;
            e.d1 = f == 1 ? cut : d1;
            e.d2 = f == 2 ? cut : d2;
            stack[sp++] = e;
        }
        ni = near_child;
    }
    *d2_out = worst;
    return best;
}

// The same answer for a query whose minimum distance `bd` is KNOWN (the GPU search has it, exactly) and shared by two
// or more references: the first of those the walk above would meet.  With the worst distance known from the start the
// walk needs nothing beyond it: FLANN takes up a deferred far child when its bound does not exceed the worst distance of
// that moment, which is never below bd -- so every branch with bound <= bd is visited by FLANN whatever it found before,
// in the fixed near-first order, and a branch whose bound exceeds bd cannot hold a point at bd.  "Cannot" holds up to
// the rounding of the bound (a sum and a difference of at most three squares, all of magnitude <= bd): branches whose
// bound lies in a thin band above bd (2e-6 relative + 1.2e-37) are entered too, and if the first tied point found lies
// under such a branch -- FLANN may or may not have gone there, depending on what it had found by then -- the caller is
// told (`uncertain`) and runs the full walk.  Otherwise: ~depth node visits, a handful of deferred children, stop at the
// first hit.  Returns the position, or -1 if no point at bd was met (then also `uncertain`).
template <int STACK>
PCC_FLANN_HD inline int32_t flann_walk_tied(const FlannNode* nodes, const float* leaf_pts, const FlannBox& root, size_t n_valid,
                                            float qx, float qy, float qz, float bd, bool* uncertain) {
    *uncertain = true;
    if (n_valid == 0) return -1;
    const float thr = bd * 1.000002f + 1.2e-37f;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, mind = 0.f;
    if (qx < root.lo[0]) { d0 = (qx - root.lo[0]) * (qx - root.lo[0]); mind += d0; }
    if (qx > root.hi[0]) { d0 = (qx - root.hi[0]) * (qx - root.hi[0]); mind += d0; }
    if (qy < root.lo[1]) { d1 = (qy - root.lo[1]) * (qy - root.lo[1]); mind += d1; }
    if (qy > root.hi[1]) { d1 = (qy - root.hi[1]) * (qy - root.hi[1]); mind += d1; }
    if (qz < root.lo[2]) { d2 = (qz - root.lo[2]) * (qz - root.lo[2]); mind += d2; }
    if (qz > root.hi[2]) { d2 = (qz - root.hi[2]) * (qz - root.hi[2]); mind += d2; }
    if (!(mind <= thr)) return -1;  // (cannot be: the root bound is below every distance; the full walk sorts it out)
    struct Deferred { int32_t node; float mind, d0, d1, d2; int band; };
    Deferred stack[STACK];
    int sp = 0;
    int band = mind > bd ? 1 : 0;  // the current branch was admitted only thanks to the band
    int32_t ni = 0;
    for (;;) {
        const FlannNode nd = nodes[ni];
        if (nd.b < 0) {
            const int32_t first = nd.a, cnt = ~nd.b;
            for (int32_t i = first; i < first + cnt; ++i) {
                const float* p = leaf_pts + (size_t)i * 4;
                const float dx = qx - p[0], dy = qy - p[1], dz = qz - p[2];
                float d = dx * dx;
                d = d + dy * dy;
                d = d + dz * dz;
                if (d == bd) {
                    int32_t w;
                    memcpy(&w, p + 3, 4);
                    *uncertain = band != 0;
                    return w;
                }
            }
            if (sp == 0) return -1;
            const Deferred e = stack[--sp];
            ni = e.node; mind = e.mind; d0 = e.d0; d1 = e.d1; d2 = e.d2; band = e.band;
            continue;
        }
        const int f = nd.b;
        const float val = f == 0 ? qx : (f == 1 ? qy : qz);
        const float cur = f == 0 ? d0 : (f == 1 ? d1 : d2);
        const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
        int32_t near_child, far_child;
        float cut;
        if (diff1 + diff2 < 0) { near_child = ni + 1; far_child = nd.a; cut = (val - nd.divhigh) * (val - nd.divhigh); }
        else { near_child = nd.a; far_child = ni + 1; cut = (val - nd.divlow) * (val - nd.divlow); }
        const float fm = mind + cut - cur;
        if (fm <= thr) {
            if (sp == STACK) return -1;  // (deeper than the stack: the caller's full walk decides)
            Deferred e;
            e.node = far_child;
            e.mind = fm;
            e.d0 = f == 0 ? cut : d0;
            e.d1 = f == 1 ? cut : d1;
            e.d2 = f == 2 ? cut : d2;
            e.band = (band != 0 || fm > bd) ? 1 : 0;
            stack[sp++] = e;
        }
        ni = near_child;
    }
}

class FlannTree {
public:
    std::vector<FlannNode> nodes;  // depth-first order, root at 0
    std::vector<float> leaf_pts;   // 4 floats per valid point, leaf order, w = bits(position in the packed cloud)
    FlannBox root{};
    size_t n_valid = 0;
    int depth = 0;                 // longest root-to-leaf path in nodes (sizes the walk's stack)

    // packed[i] = (x, y, z, w) with w's sign bit set for non-finite points (PCL's convertCloudToArray skips them)
    void build(const float* packed_xyzw, size_t n, int rule, unsigned int threads) {
        nodes.clear(); leaf_pts.clear(); pts_.clear(); map_.clear(); vind_.clear();
        n_valid = 0; depth = 0; rule_ = rule;
        for (size_t i = 0; i < n; ++i) {
            const float* p = packed_xyzw + i * 4;
            uint32_t w;
            memcpy(&w, p + 3, 4);
            if (w >> 31) continue;
            pts_.push_back(p[0]); pts_.push_back(p[1]); pts_.push_back(p[2]);
            map_.push_back((int32_t)i);
        }
        n_valid = map_.size();
        if (n_valid == 0) return;
        vind_.resize(n_valid);
        for (size_t i = 0; i < n_valid; ++i) vind_[i] = (int32_t)i;
        for (int d = 0; d < 3; ++d) min_max(vind_.data(), (int32_t)n_valid, d, root.lo[d], root.hi[d]);
        int forks = 0;
        while ((1u << forks) < threads && forks < 6) ++forks;
        FlannBox out;
        int dep = 0;
        fork(0, (int32_t)n_valid, root, forks, nodes, out, dep);
        depth = dep;
        leaf_pts.resize(n_valid * 4);
        for (size_t i = 0; i < n_valid; ++i) {
            memcpy(&leaf_pts[i * 4], &pts_[(size_t)vind_[i] * 3], 3 * sizeof(float));
            memcpy(&leaf_pts[i * 4 + 3], &map_[(size_t)vind_[i]], 4);
        }
        std::vector<float>().swap(pts_);
        std::vector<int32_t>().swap(map_);
        std::vector<int32_t>().swap(vind_);
    }

    // first reference at distance bd in FLANN's visit order (bd must be the query's exact minimum distance); falls back
    // to the full walk when the short one cannot vouch for its answer
    int32_t nearest_tied(const float q[3], float bd, bool* used_full = nullptr) const {
        bool unc = true;
        int32_t r = flann_walk_tied<32>(nodes.data(), leaf_pts.data(), root, n_valid, q[0], q[1], q[2], bd, &unc);
        if (used_full) *used_full = unc;
        if (!unc) return r;
        float d2 = 0.f;
        return nearest(q, &d2);
    }
    // host walk (any depth: the stack grows to the tree's own depth in steps)
    int32_t nearest(const float q[3], float* d2) const {
        int32_t r = flann_walk<64>(nodes.data(), leaf_pts.data(), root, n_valid, q[0], q[1], q[2], d2);
        if (r == -2) r = flann_walk<1024>(nodes.data(), leaf_pts.data(), root, n_valid, q[0], q[1], q[2], d2);
        return r;
    }

private:
    std::vector<float> pts_;     // dense valid points (PCL order), 3 floats each
    std::vector<int32_t> map_;   // dense -> position in the packed cloud (index_mapping_)
    std::vector<int32_t> vind_;  // FLANN vind_
    int rule_ = 0;

    float coord(int32_t dense, int dim) const { return pts_[(size_t)dense * 3 + dim]; }
    void min_max(const int32_t* ind, int32_t count, int dim, float& mn, float& mx) const {
        mn = mx = coord(ind[0], dim);
        for (int32_t i = 1; i < count; ++i) {
            const float v = coord(ind[i], dim);
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    }
    // KDTreeSingleIndex::planeSplit: indices rearranged into  < cutval | == cutval | > cutval ; the two boundaries
    void plane_split(int32_t* ind, int32_t count, int dim, float cutval, int32_t& lim1, int32_t& lim2) const {
        int32_t lo = 0, hi = count - 1;
        for (int pass = 0; pass < 2; ++pass) {  // pass 0 moves the values below the cut to the front, pass 1 the equal ones
            for (;;) {
                if (pass == 0) {
                    while (lo <= hi && coord(ind[lo], dim) < cutval) ++lo;
                    while (lo <= hi && coord(ind[hi], dim) >= cutval) --hi;
                } else {
                    while (lo <= hi && coord(ind[lo], dim) <= cutval) ++lo;
                    while (lo <= hi && coord(ind[hi], dim) > cutval) --hi;
                }
                if (lo > hi) break;
                const int32_t t = ind[lo]; ind[lo] = ind[hi]; ind[hi] = t;
                ++lo; --hi;
            }
            if (pass == 0) { lim1 = lo; hi = count - 1; }
            else lim2 = lo;
        }
    }
    void choose_cut(const int32_t* ind, int32_t count, const FlannBox& box, int& dim, float& cutval) const {
        float span[3];
        for (int i = 0; i < 3; ++i) span[i] = box.hi[i] - box.lo[i];
        if (rule_ == 1) {  // middleSplit
            dim = 0;
            for (int i = 1; i < 3; ++i) if (span[i] > span[dim]) dim = i;
            float mn, mx;
            min_max(ind, count, dim, mn, mx);
            cutval = (mn + mx) / 2;
            float widest = mx - mn;
            const int first = dim;
            for (int i = 0; i < 3; ++i) {
                if (i == first || !(span[i] > widest)) continue;
                min_max(ind, count, i, mn, mx);
                if (mx - mn > widest) { widest = mx - mn; dim = i; cutval = (mn + mx) / 2; }
            }
            return;
        }
        // middleSplit_
        const float EPS = 0.00001f;
        float max_span = span[0];
        for (int i = 1; i < 3; ++i) if (span[i] > max_span) max_span = span[i];
        float max_spread = -1;
        dim = 0;
        for (int i = 0; i < 3; ++i) {
            if (span[i] > (float)((1 - EPS) * max_span)) {
                float mn, mx;
                min_max(ind, count, rule_ == 2 ? i : dim, mn, mx);  // (rule 0: 1.8.4 as recalled hands over cutfeat here)
                const float spread = mx - mn;
                if (spread > max_spread) { dim = i; max_spread = spread; }
            }
        }
        const float split_val = (box.lo[dim] + box.hi[dim]) / 2;
        float mn, mx;
        min_max(ind, count, dim, mn, mx);
        cutval = split_val < mn ? mn : (split_val > mx ? mx : split_val);
    }
    // where the range is cut: the boundary of the "< cut" part when more than half lies below, of the "<= cut" part when
    // less than half does, else the middle (inside the run of values equal to the cut)
    int32_t split_range(int32_t* ind, int32_t count, const FlannBox& box, int& dim, float& cutval) const {
        choose_cut(ind, count, box, dim, cutval);
        int32_t lim1, lim2;
        plane_split(ind, count, dim, cutval, lim1, lim2);
        return lim1 > count / 2 ? lim1 : (lim2 < count / 2 ? lim2 : count / 2);
    }
    void leaf_box(int32_t left, int32_t right, FlannBox& b) const {
        for (int d = 0; d < 3; ++d) b.lo[d] = b.hi[d] = coord(vind_[left], d);
        for (int32_t k = left + 1; k < right; ++k)
            for (int d = 0; d < 3; ++d) {
                const float v = coord(vind_[k], d);
                if (v < b.lo[d]) b.lo[d] = v;
                if (v > b.hi[d]) b.hi[d] = v;
            }
    }
    static void join(const FlannBox& l, const FlannBox& r, FlannBox& out) {
        for (int d = 0; d < 3; ++d) {
            out.lo[d] = l.lo[d] < r.lo[d] ? l.lo[d] : r.lo[d];
            out.hi[d] = l.hi[d] > r.hi[d] ? l.hi[d] : r.hi[d];
        }
    }

    // one subtree, iteratively: positions inside `out` are relative to its first entry
    struct Frame {
        int32_t left, right, node, cut_at, dim, stage, depth;
        float cutval;
        FlannBox in, lbox;
    };
    void subtree(int32_t left, int32_t right, const FlannBox& in, std::vector<FlannNode>& out, FlannBox& box_out, int& depth_out) {
        std::vector<Frame> st;
        Frame root_f{};
        root_f.left = left; root_f.right = right; root_f.in = in; root_f.stage = 0; root_f.depth = 1;
        st.push_back(root_f);
        FlannBox ret{};
        int deepest = 0;
        while (!st.empty()) {
            Frame f = st.back();
            if (f.stage == 0) {
                f.node = (int32_t)out.size();
                out.push_back(FlannNode{});
                if (f.depth > deepest) deepest = f.depth;
                const int32_t count = f.right - f.left;
                if (count <= FLANN_LEAF_MAX) {
                    out[f.node] = FlannNode{f.left, ~count, 0.f, 0.f};
                    leaf_box(f.left, f.right, ret);
                    st.pop_back();
                    continue;
                }
                f.cut_at = split_range(vind_.data() + f.left, count, f.in, f.dim, f.cutval);
                f.stage = 1;
                st.back() = f;
                Frame c{};
                c.left = f.left; c.right = f.left + f.cut_at; c.in = f.in; c.in.hi[f.dim] = f.cutval; c.depth = f.depth + 1;
                st.push_back(c);
            } else if (f.stage == 1) {  // the first child is done: `ret` is its tightened box
                f.lbox = ret;
                f.stage = 2;
                out[f.node].a = (int32_t)out.size();  // the second child comes next
                st.back() = f;
                Frame c{};
                c.left = f.left + f.cut_at; c.right = f.right; c.in = f.in; c.in.lo[f.dim] = f.cutval; c.depth = f.depth + 1;
                st.push_back(c);
            } else {  // both children done: the node's bounds are what the children tightened their boxes to
                FlannNode& nd = out[f.node];
                nd.b = f.dim;
                nd.divlow = f.lbox.hi[f.dim];
                nd.divhigh = ret.lo[f.dim];
                FlannBox u;
                join(f.lbox, ret, u);
                ret = u;
                st.pop_back();
            }
        }
        box_out = ret;
        depth_out = deepest;
    }
    // the top of the tree: fork the first child's range to another thread, `forks` levels deep
    void fork(int32_t left, int32_t right, const FlannBox& in, int forks, std::vector<FlannNode>& out, FlannBox& box_out, int& depth_out) {
        const int32_t count = right - left;
        if (forks <= 0 || count <= 4096) { subtree(left, right, in, out, box_out, depth_out); return; }
        int dim;
        float cutval;
        const int32_t cut_at = split_range(vind_.data() + left, count, in, dim, cutval);
        FlannBox lin = in, rin = in, lbox, rbox;
        lin.hi[dim] = cutval;
        rin.lo[dim] = cutval;
        std::vector<FlannNode> lnodes, rnodes;
        int ldep = 0, rdep = 0;
        std::thread other([&] { fork(left, left + cut_at, lin, forks - 1, lnodes, lbox, ldep); });
        fork(left + cut_at, right, rin, forks - 1, rnodes, rbox, rdep);
        other.join();
        const int32_t me = (int32_t)out.size();
        out.push_back(FlannNode{me + 1 + (int32_t)lnodes.size(), dim, lbox.hi[dim], rbox.lo[dim]});
        const int32_t lbase = me + 1, rbase = me + 1 + (int32_t)lnodes.size();
        for (FlannNode nd : lnodes) { if (nd.b >= 0) nd.a += lbase; out.push_back(nd); }
        for (FlannNode nd : rnodes) { if (nd.b >= 0) nd.a += rbase; out.push_back(nd); }
        join(lbox, rbox, box_out);
        depth_out = 1 + (ldep > rdep ? ldep : rdep);
    }
};

}  // namespace pcc
