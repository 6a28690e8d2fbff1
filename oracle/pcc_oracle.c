/*
 * pcc_oracle.c -- CPU restatement of the reference's NN path (see pcc_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY / PARITY UNPINNED (no reference fixtures exist; PCL
 * and FLANN are absent from /root/reference and from this image).  Each block
 * cites the SURVEY.md section 9 paragraph (the restated PCL 1.7 / FLANN 1.8.4
 * behaviour) and the reference call site it serves.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction: FLANN's
 * L2_Simple was compiled for x86-64 SSE2 with separately rounded mul/add).
 */
#include "pcc_oracle.h"
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define LEAF_MAX 15 /* KDTreeSingleIndexParams(15), SURVEY 9.1 */

static inline const float *pt_at(const void *base, size_t stride, size_t i) {
    return (const float *)((const char *)base + stride * i);
}
static inline int finite3(const float *p) {
    return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]);
}
/* FLANN L2_Simple<float>::operator(): result=0; result+=diff*diff per dim (9.2) */
static inline float l2_simple(const float *a, const float *b) {
    float r = 0.0f, d;
    d = a[0] - b[0]; r += d * d;
    d = a[1] - b[1]; r += d * d;
    d = a[2] - b[2]; r += d * d;
    return r;
}

/* ===================== exhaustive definitional oracle ===================== */

void orc_nn1_exhaustive(const void *ref, size_t m, size_t rstride,
                        const void *qry, size_t n, size_t qstride,
                        int32_t *idx, float *d2) {
    for (size_t i = 0; i < n; ++i) {
        const float *q = pt_at(qry, qstride, i);
        int32_t bi = -1;
        float bd = FLT_MAX; /* KNNSimpleResultSet: worst_distance_ starts at max() and dist >= worst is rejected, so a
                               squared distance that overflowed (or equals FLT_MAX) is no neighbour */
        if (finite3(q)) {
            for (size_t j = 0; j < m; ++j) {
                const float *r = pt_at(ref, rstride, j);
                if (!finite3(r)) continue;
                float d = l2_simple(q, r);
                if (d < bd) { bd = d; bi = (int32_t)j; } /* strict <: lowest index wins */
            }
        }
        idx[i] = bi;
        d2[i] = bi < 0 ? INFINITY : bd;
    }
}

typedef struct { float d; int32_t i; } di_t;
static int di_cmp(const void *a, const void *b) {
    const di_t *x = (const di_t *)a, *y = (const di_t *)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

int orc_knn_exhaustive(const void *ref, size_t m, size_t rstride,
                       const void *qry, size_t n, size_t qstride, int k,
                       int32_t *idx, float *d2) {
    di_t *all = (di_t *)malloc(sizeof(di_t) * (m ? m : 1));
    int found = 0;
    for (size_t i = 0; i < n; ++i) {
        const float *q = pt_at(qry, qstride, i);
        size_t c = 0;
        if (finite3(q)) {
            for (size_t j = 0; j < m; ++j) {
                const float *r = pt_at(ref, rstride, j);
                if (!finite3(r)) continue;
                all[c].d = l2_simple(q, r);
                all[c].i = (int32_t)j;
                if (all[c].d < FLT_MAX) ++c; /* dist >= worst_distance_ (FLT_MAX at the start) is never inserted */
            }
            qsort(all, c, sizeof(di_t), di_cmp);
        }
        for (int t = 0; t < k; ++t) {
            if ((size_t)t < c) { idx[i * k + t] = all[t].i; d2[i * k + t] = all[t].d; }
            else { idx[i * k + t] = -1; d2[i * k + t] = INFINITY; }
        }
        found = (int)(c < (size_t)k ? c : (size_t)k);
    }
    free(all);
    return found;
}

void orc_radius_count_exhaustive(const void *ref, size_t m, size_t rstride,
                                 const void *qry, size_t n, size_t qstride,
                                 float r2, int32_t *counts) {
    for (size_t i = 0; i < n; ++i) {
        const float *q = pt_at(qry, qstride, i);
        int32_t c = 0;
        if (finite3(q))
            for (size_t j = 0; j < m; ++j) {
                const float *r = pt_at(ref, rstride, j);
                if (finite3(r) && l2_simple(q, r) < r2) ++c; /* strict, 9.3 */
            }
        counts[i] = c;
    }
}

/* =============== FLANN KDTreeSingleIndex restatement (9.2) ================ */

typedef struct { float low, high; } interval_t;
typedef struct kdnode {
    int left, right;       /* leaf: [left,right) into the reordered data */
    int divfeat;           /* inner */
    float divlow, divhigh; /* inner */
    int child1, child2;    /* node indices, -1 for leaf */
} kdnode;

struct orc_kdtree {
    size_t n;        /* valid points */
    float *pts;      /* dense n*3, PCL convertCloudToArray order (9.1) */
    int32_t *map;    /* index_mapping_: dense -> original */
    int *vind;       /* FLANN vind_ */
    float *data;     /* reordered copy (reorder = true) */
    kdnode *nodes;
    size_t nnodes, capnodes;
    int root;
    interval_t root_bbox[3];
};

static int new_node(orc_kdtree *t) {
    if (t->nnodes == t->capnodes) {
        t->capnodes = t->capnodes ? t->capnodes * 2 : 1024;
        t->nodes = (kdnode *)realloc(t->nodes, t->capnodes * sizeof(kdnode));
    }
    kdnode *nd = &t->nodes[t->nnodes];
    nd->child1 = nd->child2 = -1;
    return (int)t->nnodes++;
}

static void compute_minmax(const orc_kdtree *t, const int *ind, int count, int dim,
                           float *mn, float *mx) {
    *mn = *mx = t->pts[(size_t)ind[0] * 3 + dim];
    for (int i = 1; i < count; ++i) {
        float v = t->pts[(size_t)ind[i] * 3 + dim];
        if (v < *mn) *mn = v;
        if (v > *mx) *mx = v;
    }
}

/* KDTreeSingleIndex::planeSplit */
static void plane_split(const orc_kdtree *t, int *ind, int count, int cutfeat,
                        float cutval, int *lim1, int *lim2) {
    int left = 0, right = count - 1;
    for (;;) {
        while (left <= right && t->pts[(size_t)ind[left] * 3 + cutfeat] < cutval) ++left;
        while (left <= right && t->pts[(size_t)ind[right] * 3 + cutfeat] >= cutval) --right;
        if (left > right) break;
        int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim1 = left;
    right = count - 1;
    for (;;) {
        while (left <= right && t->pts[(size_t)ind[left] * 3 + cutfeat] <= cutval) ++left;
        while (left <= right && t->pts[(size_t)ind[right] * 3 + cutfeat] > cutval) --right;
        if (left > right) break;
        int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim2 = left;
}

/* KDTreeSingleIndex::middleSplit */
static void middle_split(const orc_kdtree *t, int *ind, int count, int *index,
                         int *cutfeat, float *cutval, const interval_t *bbox) {
    float max_span = bbox[0].high - bbox[0].low;
    *cutfeat = 0;
    *cutval = (bbox[0].high + bbox[0].low) / 2;
    for (int i = 1; i < 3; ++i) {
        float span = bbox[i].high - bbox[i].low;
        if (span > max_span) {
            max_span = span;
            *cutfeat = i;
            *cutval = (bbox[i].high + bbox[i].low) / 2;
        }
    }
    float mn, mx;
    compute_minmax(t, ind, count, *cutfeat, &mn, &mx);
    *cutval = (mn + mx) / 2;
    max_span = mx - mn;
    int k = *cutfeat;
    for (int i = 0; i < 3; ++i) {
        if (i == k) continue;
        float span = bbox[i].high - bbox[i].low;
        if (span > max_span) {
            compute_minmax(t, ind, count, i, &mn, &mx);
            span = mx - mn;
            if (span > max_span) {
                max_span = span;
                *cutfeat = i;
                *cutval = (mn + mx) / 2;
            }
        }
    }
    int lim1, lim2;
    plane_split(t, ind, count, *cutfeat, *cutval, &lim1, &lim2);
    if (lim1 > count / 2) *index = lim1;
    else if (lim2 < count / 2) *index = lim2;
    else *index = count / 2;
}

/* KDTreeSingleIndex::middleSplit_ -- the rule FLANN 1.8.x's divideTree calls, restated from memory of
 * flann/algorithms/kdtree_single_index.h (1.8.4); like everything FLANN here it cannot be checked against the real
 * source in this image.  Differences from middleSplit: the cut dimension is chosen among the sides within
 * (1 - 1e-5) of the widest side of the BOX by the largest spread of the POINTS, and the cut value is the middle of the
 * BOX side, clamped into the points' range.  1.8.4 as recalled passes `cutfeat` (not the loop variable) to
 * computeMinMax inside the selection loop -- rule 0 keeps that; rule 2 is the form with the loop variable (nanoflann's
 * later correction). */
static void middle_split_(const orc_kdtree *t, int *ind, int count, int *index,
                          int *cutfeat, float *cutval, const interval_t *bbox, int fixed) {
    const float EPS = 0.00001f;
    float max_span = bbox[0].high - bbox[0].low;
    for (int i = 1; i < 3; ++i) {
        float span = bbox[i].high - bbox[i].low;
        if (span > max_span) max_span = span;
    }
    float max_spread = -1;
    *cutfeat = 0;
    for (int i = 0; i < 3; ++i) {
        float span = bbox[i].high - bbox[i].low;
        if (span > (float)((1 - EPS) * max_span)) {
            float mn, mx;
            compute_minmax(t, ind, count, fixed ? i : *cutfeat, &mn, &mx);
            float spread = mx - mn;
            if (spread > max_spread) {
                *cutfeat = i;
                max_spread = spread;
            }
        }
    }
    float split_val = (bbox[*cutfeat].low + bbox[*cutfeat].high) / 2;
    float mn, mx;
    compute_minmax(t, ind, count, *cutfeat, &mn, &mx);
    if (split_val < mn) *cutval = mn;
    else if (split_val > mx) *cutval = mx;
    else *cutval = split_val;
    int lim1, lim2;
    plane_split(t, ind, count, *cutfeat, *cutval, &lim1, &lim2);
    if (lim1 > count / 2) *index = lim1;
    else if (lim2 < count / 2) *index = lim2;
    else *index = count / 2;
}

/* which rule divide_tree uses: 0 middleSplit_ (default: what FLANN 1.8.x's divideTree is believed to call), 1 middleSplit
 * (SURVEY 9.2), 2 middleSplit_ with the loop variable in the selection loop.  The choice shapes the tree, hence which of
 * several EQUALLY near points a search names -- never a distance. */
static int g_split_rule = 0;
void orc_set_split_rule(int rule) { g_split_rule = rule; }
int orc_get_split_rule(void) { return g_split_rule; }

/* KDTreeSingleIndex::divideTree */
static int divide_tree(orc_kdtree *t, int left, int right, interval_t *bbox) {
    int ni = new_node(t);
    if (right - left <= LEAF_MAX) {
        t->nodes[ni].left = left;
        t->nodes[ni].right = right;
        for (int d = 0; d < 3; ++d)
            bbox[d].low = bbox[d].high = t->pts[(size_t)t->vind[left] * 3 + d];
        for (int k = left + 1; k < right; ++k)
            for (int d = 0; d < 3; ++d) {
                float v = t->pts[(size_t)t->vind[k] * 3 + d];
                if (v < bbox[d].low) bbox[d].low = v;
                if (v > bbox[d].high) bbox[d].high = v;
            }
    } else {
        int idx, cutfeat;
        float cutval;
        if (g_split_rule == 1) middle_split(t, t->vind + left, right - left, &idx, &cutfeat, &cutval, bbox);
        else middle_split_(t, t->vind + left, right - left, &idx, &cutfeat, &cutval, bbox, g_split_rule == 2);
        interval_t lb[3], rb[3];
        memcpy(lb, bbox, sizeof(lb));
        memcpy(rb, bbox, sizeof(rb));
        lb[cutfeat].high = cutval;
        int c1 = divide_tree(t, left, left + idx, lb);
        rb[cutfeat].low = cutval;
        int c2 = divide_tree(t, left + idx, right, rb);
        kdnode *nd = &t->nodes[ni]; /* re-fetch: realloc may have moved nodes */
        nd->divfeat = cutfeat;
        nd->child1 = c1;
        nd->child2 = c2;
        nd->divlow = lb[cutfeat].high;
        nd->divhigh = rb[cutfeat].low;
        for (int d = 0; d < 3; ++d) {
            bbox[d].low = lb[d].low < rb[d].low ? lb[d].low : rb[d].low;
            bbox[d].high = lb[d].high > rb[d].high ? lb[d].high : rb[d].high;
        }
    }
    return ni;
}

orc_kdtree *orc_kdtree_build(const void *pts, size_t m, size_t stride) {
    orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof(*t));
    t->pts = (float *)malloc(sizeof(float) * 3 * (m ? m : 1));
    t->map = (int32_t *)malloc(sizeof(int32_t) * (m ? m : 1));
    /* KdTreeFLANN::convertCloudToArray: skip invalid, keep order (9.1) */
    size_t c = 0;
    for (size_t i = 0; i < m; ++i) {
        const float *p = pt_at(pts, stride, i);
        if (!finite3(p)) continue;
        t->pts[c * 3 + 0] = p[0];
        t->pts[c * 3 + 1] = p[1];
        t->pts[c * 3 + 2] = p[2];
        t->map[c] = (int32_t)i;
        ++c;
    }
    t->n = c;
    if (c == 0) { orc_kdtree_free(t); return NULL; }
    t->vind = (int *)malloc(sizeof(int) * c);
    for (size_t i = 0; i < c; ++i) t->vind[i] = (int)i;
    for (int d = 0; d < 3; ++d) {
        float mn, mx;
        compute_minmax(t, t->vind, (int)c, d, &mn, &mx);
        t->root_bbox[d].low = mn;
        t->root_bbox[d].high = mx;
    }
    interval_t bb[3];
    memcpy(bb, t->root_bbox, sizeof(bb));
    t->root = divide_tree(t, 0, (int)c, bb);
    t->data = (float *)malloc(sizeof(float) * 3 * c);
    for (size_t i = 0; i < c; ++i)
        memcpy(t->data + i * 3, t->pts + (size_t)t->vind[i] * 3, 3 * sizeof(float));
    return t;
}

void orc_kdtree_free(orc_kdtree *t) {
    if (!t) return;
    free(t->pts); free(t->map); free(t->vind); free(t->data); free(t->nodes);
    free(t);
}
size_t orc_kdtree_size(const orc_kdtree *t) { return t ? t->n : 0; }

/* result sets ------------------------------------------------------------- */
typedef struct {
    int knn_cap, knn_cnt; /* KNNSimpleResultSet */
    float worst;
    di_t *items;
    int radius_mode;      /* RadiusResultSet */
    float radius;
    size_t rcnt, rcap;
} rset_t;

static inline void rset_add(rset_t *rs, float dist, int32_t index) {
    if (rs->radius_mode) {
        if (dist < rs->radius) { /* strict (9.3) */
            if (rs->rcnt == rs->rcap) {
                rs->rcap = rs->rcap ? rs->rcap * 2 : 64;
                rs->items = (di_t *)realloc(rs->items, rs->rcap * sizeof(di_t));
            }
            rs->items[rs->rcnt].d = dist;
            rs->items[rs->rcnt].i = index;
            rs->rcnt++;
        }
        return;
    }
    /* KNNSimpleResultSet::addPoint: ties keep the earlier-visited point */
    if (dist >= rs->worst) return;
    if (rs->knn_cnt < rs->knn_cap) ++rs->knn_cnt;
    int i;
    for (i = rs->knn_cnt - 1; i > 0; --i) {
        if (rs->items[i - 1].d > dist) rs->items[i] = rs->items[i - 1];
        else break;
    }
    rs->items[i].d = dist;
    rs->items[i].i = index;
    rs->worst = rs->items[rs->knn_cap - 1].d;
}

/* KDTreeSingleIndex::searchLevel, epsError = 1 (eps = 0, exact) */
static void search_level(const orc_kdtree *t, rset_t *rs, const float *q, int ni,
                         float mindistsq, float dists[3]) {
    const kdnode *nd = &t->nodes[ni];
    if (nd->child1 < 0) {
        float worst = rs->radius_mode ? rs->radius : rs->worst;
        for (int i = nd->left; i < nd->right; ++i) {
            float d = l2_simple(q, t->data + (size_t)i * 3);
            if (d < worst) rset_add(rs, d, t->vind[i]);
        }
        return;
    }
    int f = nd->divfeat;
    float val = q[f];
    float diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
    int best, other;
    float cut;
    if ((diff1 + diff2) < 0) {
        best = nd->child1; other = nd->child2;
        cut = (val - nd->divhigh) * (val - nd->divhigh);
    } else {
        best = nd->child2; other = nd->child1;
        cut = (val - nd->divlow) * (val - nd->divlow);
    }
    search_level(t, rs, q, best, mindistsq, dists);
    float dst = dists[f];
    mindistsq = mindistsq + cut - dst;
    dists[f] = cut;
    float worst = rs->radius_mode ? rs->radius : rs->worst;
    if (mindistsq <= worst) search_level(t, rs, q, other, mindistsq, dists);
    dists[f] = dst;
}

static void find_neighbors(const orc_kdtree *t, rset_t *rs, const float *q) {
    float dists[3] = {0, 0, 0};
    float distsq = 0;
    for (int d = 0; d < 3; ++d) { /* computeInitialDistances */
        if (q[d] < t->root_bbox[d].low) {
            dists[d] = (q[d] - t->root_bbox[d].low) * (q[d] - t->root_bbox[d].low);
            distsq += dists[d];
        }
        if (q[d] > t->root_bbox[d].high) {
            dists[d] = (q[d] - t->root_bbox[d].high) * (q[d] - t->root_bbox[d].high);
            distsq += dists[d];
        }
    }
    search_level(t, rs, q, t->root, distsq, dists);
}

int orc_kdtree_knn(const orc_kdtree *t, const float q[3], int k, int32_t *idx, float *d2) {
    if (!t || !finite3(q)) return 0; /* PCL asserts on invalid query (9.1) */
    if ((size_t)k > t->n) k = (int)t->n;
    if (k <= 0) return 0;
    di_t stack_items[64];
    rset_t rs;
    memset(&rs, 0, sizeof(rs));
    rs.knn_cap = k;
    rs.items = k <= 64 ? stack_items : (di_t *)malloc(sizeof(di_t) * k);
    rs.items[k - 1].d = FLT_MAX; /* KNNSimpleResultSet::clear */
    rs.worst = FLT_MAX;
    find_neighbors(t, &rs, q);
    for (int i = 0; i < rs.knn_cnt; ++i) {
        idx[i] = t->map[rs.items[i].i]; /* index_mapping_ (9.1) */
        d2[i] = rs.items[i].d;
    }
    int c = rs.knn_cnt;
    if (rs.items != stack_items) free(rs.items);
    return c;
}

int orc_kdtree_radius(const orc_kdtree *t, const float q[3], float r2, int sorted,
                      int32_t *idx, float *d2, int cap) {
    if (!t || !finite3(q)) return 0;
    rset_t rs;
    memset(&rs, 0, sizeof(rs));
    rs.radius_mode = 1;
    rs.radius = r2;
    find_neighbors(t, &rs, q);
    for (size_t i = 0; i < rs.rcnt; ++i) rs.items[i].i = t->map[rs.items[i].i];
    if (sorted && rs.rcnt) qsort(rs.items, rs.rcnt, sizeof(di_t), di_cmp); /* (dist, index); (an empty result has no array) */
    for (size_t i = 0; i < rs.rcnt && (int)i < cap; ++i) {
        idx[i] = rs.items[i].i;
        d2[i] = rs.items[i].d;
    }
    int c = (int)rs.rcnt;
    free(rs.items);
    return c;
}

void orc_kdtree_nn1_batch(const orc_kdtree *t, const void *qry, size_t n,
                          size_t qstride, int32_t *idx, float *d2) {
    for (size_t i = 0; i < n; ++i) {
        int32_t bi = -1;
        float bd = INFINITY;
        if (orc_kdtree_knn(t, pt_at(qry, qstride, i), 1, &bi, &bd) != 1) { bi = -1; bd = INFINITY; }
        idx[i] = bi;
        d2[i] = bd;
    }
}

typedef struct {
    const orc_kdtree *t; const void *q; size_t n, stride; int32_t *idx; float *d2;
} mt_arg;
static void *mt_run(void *p) {
    mt_arg *a = (mt_arg *)p;
    orc_kdtree_nn1_batch(a->t, a->q, a->n, a->stride, a->idx, a->d2);
    return NULL;
}
void orc_kdtree_nn1_batch_mt(const orc_kdtree *t, const void *qry, size_t n,
                             size_t qstride, int32_t *idx, float *d2, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    mt_arg args[256];
    size_t per = (n + nthreads - 1) / nthreads;
    int used = 0;
    for (int i = 0; i < nthreads; ++i) {
        size_t b = per * i;
        if (b >= n) break;
        size_t e = b + per < n ? b + per : n;
        args[i] = (mt_arg){t, (const char *)qry + qstride * b, e - b, qstride, idx + b, d2 + b};
        pthread_create(&th[i], NULL, mt_run, &args[i]);
        ++used;
    }
    for (int i = 0; i < used; ++i) pthread_join(th[i], NULL);
}

/* ============== matchRIFTFeaturesKnn (src/comparator.cpp:560-588) ========== */

int orc_match_rift_knn(const void *des1, size_t n1, const void *des2, size_t n2,
                       size_t stride, int32_t *out) {
    orc_kdtree *t = orc_kdtree_build(des1, n1, stride); /* :564-565 */
    int c = 0;
    out[c++] = 0; /* std::vector<int> correspondence(1), :568 */
    for (size_t i = 0; i < n2; ++i) {
        int32_t nb;
        float sd;
        int found = orc_kdtree_knn(t, pt_at(des2, stride, i), 1, &nb, &sd); /* :576 */
        if (found == 1 && sd < 0.05f) out[c++] = nb;                        /* :579-580 */
    }
    orc_kdtree_free(t);
    return c;
}

/* ============ keypoint snap loop (src/comparator.cpp:696-713) ============== */

void orc_first_within(const void *pts, size_t m, size_t stride, const void *qry, size_t n, size_t qstride,
                      double radius, int32_t *idx) {
    for (size_t i = 0; i < n; ++i) {
        const float *s = pt_at(qry, qstride, i);
        idx[i] = -1;
        if (!finite3(s)) continue;
        for (size_t j = 0; j < m; ++j) {
            const float *p = pt_at(pts, stride, j);
            /* pow(float - float, 2): the difference is a float, pow promotes it to double */
            float fx = s[0] - p[0], fy = s[1] - p[1], fz = s[2] - p[2];
            if (sqrt(pow(fx, 2) + pow(fy, 2) + pow(fz, 2)) < radius) { idx[i] = (int32_t)j; break; }
        }
    }
}

/* ======== NormalEstimation (src/segmentation.cpp:232-241) [PCL 1.7, recalled] ======= */

/* pcl::computeRoots2 */
static void roots2(float b, float c, float r[3]) {
    r[0] = 0.0f;
    float d = b * b - 4.0f * c;
    if (d < 0.0f) d = 0.0f;
    float sd = sqrtf(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}
/* pcl::computeRoots: eigenvalues of a symmetric 3x3 (row-major m[9]) in increasing order */
static void roots3(const float m[9], float r[3]) {
    float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    float c2 = m[0] + m[4] + m[8];
    if (fabsf(c0) < FLT_EPSILON) { roots2(c2, c1, r); return; }
    const float s_inv3 = (float)(1.0 / 3.0), s_sqrt3 = sqrtf(3.0f);
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    float rho = sqrtf(-a_over_3);
    float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
    float ct = cosf(theta), st = sinf(theta);
    r[0] = c2_over_3 + 2.0f * rho * ct;
    r[1] = c2_over_3 - rho * (ct + s_sqrt3 * st);
    r[2] = c2_over_3 - rho * (ct - s_sqrt3 * st);
    float t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    if (r[1] >= r[2]) {
        t = r[1]; r[1] = r[2]; r[2] = t;
        if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    }
    if (r[0] <= 0.0f) roots2(c2, c1, r);
}
static void cross3(const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
/* pcl::eigen33 (smallest eigenpair) + solvePlaneParameters */
static void plane_params(const float cov[9], float n[3], float *curvature) {
    float scale = 0.0f;
    for (int i = 0; i < 9; ++i) if (fabsf(cov[i]) > scale) scale = fabsf(cov[i]);
    if (scale <= FLT_MIN) scale = 1.0f;
    float sm[9], ev[3];
    for (int i = 0; i < 9; ++i) sm[i] = cov[i] / scale;
    roots3(sm, ev);
    float eigenvalue = ev[0] * scale;
    sm[0] -= ev[0]; sm[4] -= ev[0]; sm[8] -= ev[0];
    float v1[3], v2[3], v3[3];
    cross3(sm + 0, sm + 3, v1);
    cross3(sm + 0, sm + 6, v2);
    cross3(sm + 3, sm + 6, v3);
    float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
    float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
    const float *v; float l;
    if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
    else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
    else { v = v3; l = l3; }
    float s = sqrtf(l);
    n[0] = v[0] / s; n[1] = v[1] / s; n[2] = v[2] / s;
    float eig_sum = cov[0] + cov[4] + cov[8];
    *curvature = eig_sum != 0.0f ? fabsf(eigenvalue / eig_sum) : 0.0f;
}

static void normal_of(const void *pts, size_t stride, const int32_t *nbr, int cnt, const float *p, const float vp[3], float *o) {
    if (cnt < 3) { o[0] = o[1] = o[2] = o[3] = NAN; return; }
    /* computeMeanAndCovarianceMatrix: single pass, float accumulators */
    float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < cnt; ++j) {
        const float *q = pt_at(pts, stride, (size_t)nbr[j]);
        a[0] += q[0] * q[0]; a[1] += q[0] * q[1]; a[2] += q[0] * q[2];
        a[3] += q[1] * q[1]; a[4] += q[1] * q[2]; a[5] += q[2] * q[2];
        a[6] += q[0]; a[7] += q[1]; a[8] += q[2];
    }
    /* accu /= point_count: Eigen 3.2's operator/=(scalar) multiplies by Scalar(1)/other for floats */
    const float inv_cnt = 1.0f / (float)cnt;
    for (int i = 0; i < 9; ++i) a[i] *= inv_cnt;
    float cov[9];
    cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
    cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float n[3], curv;
    plane_params(cov, n, &curv);
    /* flipNormalTowardsViewpoint */
    float vx = vp[0] - p[0], vy = vp[1] - p[1], vz = vp[2] - p[2];
    float cos_theta = vx * n[0] + vy * n[1] + vz * n[2];
    if (cos_theta < 0) { n[0] *= -1; n[1] *= -1; n[2] *= -1; }
    o[0] = n[0]; o[1] = n[1]; o[2] = n[2]; o[3] = curv;
}

void orc_normals_from_neighbours(const void *pts, size_t n, size_t stride, const int32_t *nbr, int k,
                                 const float vp[3], float *out) {
    for (size_t i = 0; i < n; ++i) {
        const float *p = pt_at(pts, stride, i);
        int cnt = 0;
        while (cnt < k && nbr[i * k + cnt] >= 0) ++cnt;
        if (!finite3(p)) cnt = 0;
        normal_of(pts, stride, nbr + i * k, cnt, p, vp, out + i * 4);
    }
}

void orc_normals(const void *pts, size_t n, size_t stride, int k, const float vp[3], float *out) {
    orc_kdtree *t = orc_kdtree_build(pts, n, stride);
    int32_t *ni = (int32_t *)malloc(sizeof(int32_t) * k);
    float *nd = (float *)malloc(sizeof(float) * k);
    for (size_t i = 0; i < n; ++i) {
        const float *p = pt_at(pts, stride, i);
        int cnt = (t && finite3(p)) ? orc_kdtree_knn(t, p, k, ni, nd) : 0;
        normal_of(pts, stride, ni, cnt, p, vp, out + i * 4);
    }
    free(ni); free(nd);
    orc_kdtree_free(t);
}

/* NormalEstimation with setRadiusSearch(radius) (src/comparator.cpp:628-635, radius 0.03): the neighbours are the
 * sorted radiusSearch result (search::KdTree sorts by default), r2 = float(radius * radius) in double */
void orc_normals_radius(const void *pts, size_t n, size_t stride, double radius, const float vp[3], float *out) {
    orc_kdtree *t = orc_kdtree_build(pts, n, stride);
    const float r2 = (float)(radius * radius);
    int cap = 1024;
    int32_t *ni = (int32_t *)malloc(sizeof(int32_t) * cap);
    float *nd = (float *)malloc(sizeof(float) * cap);
    for (size_t i = 0; i < n; ++i) {
        const float *p = pt_at(pts, stride, i);
        int cnt = 0;
        if (t && finite3(p)) {
            cnt = orc_kdtree_radius(t, p, r2, 1, ni, nd, cap);
            if (cnt > cap) {  /* the count is returned even when the buffers were too small */
                cap = cnt + cnt / 2;
                ni = (int32_t *)realloc(ni, sizeof(int32_t) * cap);
                nd = (float *)realloc(nd, sizeof(float) * cap);
                cnt = orc_kdtree_radius(t, p, r2, 1, ni, nd, cap);
            }
        }
        normal_of(pts, stride, ni, cnt, p, vp, out + i * 4);
    }
    free(ni); free(nd);
    orc_kdtree_free(t);
}

/* ======== SACSegmentation, SACMODEL_PLANE + SAC_RANSAC (src/segmentation.cpp:79-117) ========
 * [recalled from PCL 1.7 sample_consensus/{ransac,sac_model,sac_model_plane}.hpp, segmentation/sac_segmentation.hpp,
 *  Boost.Random, Eigen 3.2 with SSE3+ (the PCL 1.7 Ubuntu packages are built with -msse4.2)]
 * 4-float Eigen reductions (dot, squaredNorm) are one packet product + predux = (p0 + p1) + (p2 + p3). */

typedef struct { uint32_t mt[624]; int idx; } mt19937_t;
static void mt_seed(mt19937_t *g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937_t *g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
void orc_mt19937_raw(uint32_t seed, uint32_t *out, size_t n) { /* for the known-answer test of the generator */
    mt19937_t g; mt_seed(&g, seed);
    for (size_t i = 0; i < n; ++i) out[i] = mt_next(&g);
}
/* boost::variate_generator<mt19937&, uniform_int<>(0, INT_MAX)>: bucket size 2 -> eng() / 2, never rejected */
static int sac_rnd(mt19937_t *g) { return (int)(mt_next(g) / 2u); }

static float dot4(const float a[4], const float b[4]) {
    return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
}
/* SampleConsensusModelPlane::isSampleGood / the collinearity check of computeModelCoefficients:
 * Array4f (p1-p0)/(p2-p0) on the first three lanes */
static int plane_sample_degenerate(const float *p0, const float *p1, const float *p2) {
    float d0 = (p1[0] - p0[0]) / (p2[0] - p0[0]);
    float d1 = (p1[1] - p0[1]) / (p2[1] - p0[1]);
    float d2 = (p1[2] - p0[2]) / (p2[2] - p0[2]);
    return (d0 == d1) && (d2 == d1);
}
static int plane_from_sample(const float *p0, const float *p1, const float *p2, float c[4]) {
    float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    float b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    if (plane_sample_degenerate(p0, p1, p2)) return 0;
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
    c[3] = 0.0f;
    /* VectorXf::normalize(): *this /= norm(), i.e. *= 1/norm in Eigen 3.2 */
    float nrm = sqrtf(dot4(c, c));
    float inv = 1.0f / nrm;
    c[0] *= inv; c[1] *= inv; c[2] *= inv; c[3] *= inv;
    float p[4] = {p0[0], p0[1], p0[2], 1.0f};
    c[3] = -1.0f * dot4(c, p);
    return 1;
}
static int plane_inlier(const float c[4], const float *q, double threshold) {
    float p[4] = {q[0], q[1], q[2], 1.0f};
    return fabs((double)dot4(c, p)) < threshold;
}

long orc_sac_plane(const void *pts, size_t n, size_t stride, int max_iterations, double threshold, double probability,
                   int optimize, int32_t *inliers, float coeff[4], int *iterations_out) {
    if (iterations_out) *iterations_out = 0;
    coeff[0] = coeff[1] = coeff[2] = coeff[3] = 0.0f;
    if (n < 3) return 0; /* getSamples: "Can not select 3 unique points out of N" -> no model */
    mt19937_t g; mt_seed(&g, 12345u);
    int32_t *shuffled = (int32_t *)malloc(sizeof(int32_t) * n);
    for (size_t i = 0; i < n; ++i) shuffled[i] = (int32_t)i;
    int iterations = 0, best_count = -INT32_MAX, have_model = 0;
    double k = 1.0;
    const double log_probability = log(1.0 - probability);
    const double one_over_indices = 1.0 / (double)n;
    unsigned skipped = 0;
    const unsigned max_skip = (unsigned)max_iterations * 10u;
    float best[4] = {0, 0, 0, 0};
    while (iterations < k && skipped < max_skip) {
        /* getSamples: up to 1000 draws until isSampleGood */
        int32_t smp[3]; int got = 0;
        for (int it = 0; it < 1000 && !got; ++it) {
            for (unsigned i = 0; i < 3; ++i) {
                size_t j = i + (size_t)sac_rnd(&g) % (n - i);
                int32_t t = shuffled[i]; shuffled[i] = shuffled[j]; shuffled[j] = t;
            }
            smp[0] = shuffled[0]; smp[1] = shuffled[1]; smp[2] = shuffled[2];
            got = !plane_sample_degenerate(pt_at(pts, stride, smp[0]), pt_at(pts, stride, smp[1]), pt_at(pts, stride, smp[2]));
        }
        if (!got) break; /* "No samples could be selected!" */
        float c[4];
        if (!plane_from_sample(pt_at(pts, stride, smp[0]), pt_at(pts, stride, smp[1]), pt_at(pts, stride, smp[2]), c)) {
            ++skipped;
            continue;
        }
        int cnt = 0;
        for (size_t i = 0; i < n; ++i) cnt += plane_inlier(c, pt_at(pts, stride, i), threshold);
        if (cnt > best_count) {
            best_count = cnt;
            have_model = 1;
            memcpy(best, c, sizeof(best));
            double w = (double)best_count * one_over_indices;
            double p_no_outliers = 1.0 - pow(w, 3.0);
            if (p_no_outliers < DBL_EPSILON) p_no_outliers = DBL_EPSILON;
            if (p_no_outliers > 1.0 - DBL_EPSILON) p_no_outliers = 1.0 - DBL_EPSILON;
            k = log_probability / log(p_no_outliers);
        }
        ++iterations;
        if (iterations > max_iterations) break;
    }
    free(shuffled);
    if (iterations_out) *iterations_out = iterations;
    if (!have_model) return 0;
    long m = 0;
    for (size_t i = 0; i < n; ++i) if (plane_inlier(best, pt_at(pts, stride, i), threshold)) inliers[m++] = (int32_t)i;
    memcpy(coeff, best, sizeof(best));
    if (optimize && m >= 4) {
        /* optimizeModelCoefficients: least-squares plane through the inliers (single-pass float sums, in order) */
        float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (long j = 0; j < m; ++j) {
            const float *q = pt_at(pts, stride, (size_t)inliers[j]);
            a[0] += q[0] * q[0]; a[1] += q[0] * q[1]; a[2] += q[0] * q[2];
            a[3] += q[1] * q[1]; a[4] += q[1] * q[2]; a[5] += q[2] * q[2];
            a[6] += q[0]; a[7] += q[1]; a[8] += q[2];
        }
        const float inv_cnt = 1.0f / (float)m;
        for (int i = 0; i < 9; ++i) a[i] *= inv_cnt;
        float cov[9];
        cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
        cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
        cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
        float nrm[3], curv;
        plane_params(cov, nrm, &curv);
        float o[4] = {nrm[0], nrm[1], nrm[2], 0.0f};
        float cen[4] = {a[6], a[7], a[8], 0.0f};
        o[3] = -1.0f * dot4(o, cen);
        memcpy(coeff, o, sizeof(o));
        m = 0; /* refine inliers: selectWithinDistance with the refined coefficients */
        for (size_t i = 0; i < n; ++i) if (plane_inlier(o, pt_at(pts, stride, i), threshold)) inliers[m++] = (int32_t)i;
    }
    return m;
}

/* ======== RegionGrowing (src/segmentation.cpp:259-271) [PCL 1.7, recalled] ========= */

typedef struct { float c; int32_t i; } resid_t;
static int resid_cmp(const void *a, const void *b) {
    const resid_t *x = (const resid_t *)a, *y = (const resid_t *)b;
    int nx = x->c != x->c, ny = y->c != y->c; /* NaN curvature (no normal): last */
    if (nx != ny) return nx - ny;
    if (x->c < y->c) return -1;
    if (x->c > y->c) return 1;
    return (x->i > y->i) - (x->i < y->i); /* std::sort leaves equal curvatures unspecified; index order here */
}

int orc_region_growing(size_t n, const float *normals, const int32_t *nbr, int k, float smoothness,
                       float curvature_threshold, int min_size, int max_size, int32_t *labels) {
    int32_t *seg = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) seg[i] = -1;
    resid_t *res = (resid_t *)malloc(sizeof(resid_t) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) { res[i].c = normals[i * 4 + 3]; res[i].i = (int32_t)i; }
    qsort(res, n, sizeof(resid_t), resid_cmp);
    int32_t *queue = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    int *seg_size = (int *)malloc(sizeof(int) * (n ? n : 1));
    const float cosine_threshold = cosf(smoothness);
    size_t segmented = 0, seed_counter = 0;
    int nseg = 0;
    int32_t seed = n ? res[0].i : 0;
    while (segmented < n) {
        /* growRegion */
        size_t qh = 0, qt = 0;
        queue[qt++] = seed;
        seg[seed] = nseg;
        int cnt = 1;
        while (qh < qt) {
            int32_t cur = queue[qh++];
            const float *nc = normals + (size_t)cur * 4;
            for (int j = 0; j < k; ++j) {
                int32_t idx = nbr[(size_t)cur * k + j];
                if (idx < 0) break;
                if (seg[idx] != -1) continue;
                /* validatePoint, smooth mode: angle between the CURRENT seed's normal and the neighbour's */
                const float *nn = normals + (size_t)idx * 4;
                float dot = fabsf(nn[0] * nc[0] + nn[1] * nc[1] + nn[2] * nc[2]);
                if (dot < cosine_threshold) continue;
                seg[idx] = nseg;
                ++cnt;
                if (!(nn[3] > curvature_threshold)) queue[qt++] = idx;
            }
        }
        seg_size[nseg++] = cnt;
        segmented += (size_t)cnt;
        for (size_t s = seed_counter + 1; s < n; ++s)
            if (seg[res[s].i] == -1) { seed = res[s].i; seed_counter = s; break; }
    }
    /* assembleRegions + size filter: kept clusters keep their creation order */
    int32_t *remap = (int32_t *)malloc(sizeof(int32_t) * (nseg ? nseg : 1));
    int kept = 0;
    for (int s = 0; s < nseg; ++s) remap[s] = (seg_size[s] >= min_size && seg_size[s] <= max_size) ? kept++ : -1;
    for (size_t i = 0; i < n; ++i) labels[i] = seg[i] >= 0 ? remap[seg[i]] : -1;
    free(remap); free(seg_size); free(queue); free(res); free(seg);
    return kept;
}

/* ============ VoxelGrid (src/segmentation.cpp:69-74, 224-229) ============== */

typedef struct { unsigned int idx; unsigned int pt; } vox_t;
static int vox_cmp(const void *a, const void *b) {
    const vox_t *x = (const vox_t *)a, *y = (const vox_t *)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return (x->pt > y->pt) - (x->pt < y->pt);
}

long orc_voxel_grid(const void *pts, size_t m, size_t stride, float leaf, int has_rgb, void *out, size_t out_stride) {
    float inv = 1.0f / leaf;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    size_t valid = 0;
    for (size_t i = 0; i < m; ++i) { /* getMinMax3D over the finite points */
        const float *p = pt_at(pts, stride, i);
        if (!finite3(p)) continue;
        for (int a = 0; a < 3; ++a) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
        ++valid;
    }
    if (!valid) return 0;
    int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1,
            dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) return -1;
    int min_b[3], div_b[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = (int)floor(mn[a] * inv);
        div_b[a] = (int)floor(mx[a] * inv) - min_b[a] + 1;
    }
    vox_t *v = (vox_t *)malloc(sizeof(vox_t) * valid);
    size_t c = 0;
    for (size_t i = 0; i < m; ++i) {
        const float *p = pt_at(pts, stride, i);
        if (!finite3(p)) continue;
        int ijk0 = (int)(floor(p[0] * inv) - (float)min_b[0]);
        int ijk1 = (int)(floor(p[1] * inv) - (float)min_b[1]);
        int ijk2 = (int)(floor(p[2] * inv) - (float)min_b[2]);
        v[c].idx = (unsigned int)(ijk0 + ijk1 * div_b[0] + ijk2 * div_b[0] * div_b[1]);
        v[c].pt = (unsigned int)i;
        ++c;
    }
    qsort(v, c, sizeof(vox_t), vox_cmp);
    long nv = 0;
    for (size_t s = 0; s < c;) {
        size_t e = s;
        float cx = 0, cy = 0, cz = 0, cr = 0, cg = 0, cb = 0;
        while (e < c && v[e].idx == v[s].idx) {
            const float *p = pt_at(pts, stride, v[e].pt);
            cx += p[0]; cy += p[1]; cz += p[2];
            if (has_rgb) {
                uint32_t w;
                memcpy(&w, (const char *)p + 16, 4);
                cr += (float)((w >> 16) & 0xff); cg += (float)((w >> 8) & 0xff); cb += (float)(w & 0xff);
            }
            ++e;
        }
        float n = (float)(e - s);
        float *o = (float *)((char *)out + (size_t)nv * out_stride);
        o[0] = cx / n; o[1] = cy / n; o[2] = cz / n;
        if (out_stride >= 16) o[3] = 1.0f;
        if (has_rgb) {
            uint32_t w = ((uint32_t)(int)(cr / n) << 16) | ((uint32_t)(int)(cg / n) << 8) | (uint32_t)(int)(cb / n);
            memcpy((char *)o + 16, &w, 4);
        }
        ++nv;
        s = e;
    }
    free(v);
    return nv;
}

/* ======= EuclideanClusterExtraction (src/segmentation.cpp:125-131, 9.4) ==== */

typedef struct { int32_t first; int32_t size; int32_t id; } clus_t;
static int clus_cmp(const void *a, const void *b) {
    const clus_t *x = (const clus_t *)a, *y = (const clus_t *)b;
    if (x->size != y->size) return y->size - x->size; /* largest first */
    return (x->first > y->first) - (x->first < y->first); /* PCL leaves ties unspecified */
}

int orc_euclidean_clusters(const void *pts, size_t m, size_t stride, float tolerance,
                           uint32_t min_size, uint32_t max_size, int32_t *labels,
                           int32_t *cluster_sizes, int max_clusters) {
    for (size_t i = 0; i < m; ++i) labels[i] = -1;
    orc_kdtree *t = orc_kdtree_build(pts, m, stride);
    if (!t) return 0;
    /* radiusSearch(pt, double(tol)): r2 = float(double(tol)*double(tol)) (9.3) */
    float r2 = (float)((double)tolerance * (double)tolerance);
    uint8_t *processed = (uint8_t *)calloc(m, 1);
    int32_t *queue = (int32_t *)malloc(sizeof(int32_t) * m);
    int32_t *tmp_label = (int32_t *)malloc(sizeof(int32_t) * m);
    int ncap = 1024, nn;
    int32_t *nidx = (int32_t *)malloc(sizeof(int32_t) * ncap);
    float *nd2 = (float *)malloc(sizeof(float) * ncap);
    clus_t *cl = NULL;
    size_t ncl = 0, capcl = 0;
    for (size_t i = 0; i < m; ++i) tmp_label[i] = -1;
    for (size_t i = 0; i < m; ++i) {
        if (processed[i]) continue;
        if (!finite3(pt_at(pts, stride, i))) continue; /* PCL asserts; callers strip NaNs first */
        size_t qn = 0, sq = 0;
        queue[qn++] = (int32_t)i;
        processed[i] = 1;
        while (sq < qn) {
            const float *p = pt_at(pts, stride, (size_t)queue[sq]);
            nn = orc_kdtree_radius(t, p, r2, 1, nidx, nd2, ncap);
            if (nn > ncap) {
                ncap = nn * 2;
                nidx = (int32_t *)realloc(nidx, sizeof(int32_t) * ncap);
                nd2 = (float *)realloc(nd2, sizeof(float) * ncap);
                nn = orc_kdtree_radius(t, p, r2, 1, nidx, nd2, ncap);
            }
            /* nn_start_idx = 1: the first sorted result is assumed to be the point itself */
            for (int j = 1; j < nn; ++j) {
                if (nidx[j] < 0 || processed[nidx[j]]) continue;
                queue[qn++] = nidx[j];
                processed[nidx[j]] = 1;
            }
            ++sq;
        }
        if (qn >= min_size && qn <= max_size) {
            if (ncl == capcl) {
                capcl = capcl ? capcl * 2 : 256;
                cl = (clus_t *)realloc(cl, capcl * sizeof(clus_t));
            }
            cl[ncl].first = (int32_t)i; /* lowest member index: i is the seed */
            cl[ncl].size = (int32_t)qn;
            cl[ncl].id = (int32_t)ncl;
            for (size_t k = 0; k < qn; ++k) tmp_label[queue[k]] = (int32_t)ncl;
            ++ncl;
        }
    }
    /* std::sort(clusters.rbegin(), clusters.rend(), comparePointClusters) */
    if (ncl) qsort(cl, ncl, sizeof(clus_t), clus_cmp);
    int32_t *remap = (int32_t *)malloc(sizeof(int32_t) * (ncl ? ncl : 1));
    for (size_t r = 0; r < ncl; ++r) {
        remap[cl[r].id] = (int32_t)r;
        if ((int)r < max_clusters && cluster_sizes) cluster_sizes[r] = cl[r].size;
    }
    for (size_t i = 0; i < m; ++i) labels[i] = tmp_label[i] >= 0 ? remap[tmp_label[i]] : -1;
    free(remap); free(cl); free(nidx); free(nd2); free(tmp_label); free(queue); free(processed);
    orc_kdtree_free(t);
    return (int)ncl;
}

/* ======== StatisticalOutlierRemoval (src/comparator.cpp:1523-1541, 9.6) ==== */

size_t orc_sor(const void *pts, size_t n, size_t stride, int mean_k, double stddev_mult,
               float *mean_dist, uint8_t *inlier, double *thresh) {
    orc_kdtree *t = orc_kdtree_build(pts, n, stride);
    int k = mean_k + 1;
    int32_t *ni = (int32_t *)malloc(sizeof(int32_t) * k);
    float *nd = (float *)malloc(sizeof(float) * k);
    size_t valid = 0;
    for (size_t i = 0; i < n; ++i) {
        const float *p = pt_at(pts, stride, i);
        mean_dist[i] = 0.0f;
        if (!t || !finite3(p)) continue;
        int found = orc_kdtree_knn(t, p, k, ni, nd);
        if (found != k) continue; /* "no neighbours found": distance stays 0 */
        double s = 0.0;
        for (int j = 1; j < k; ++j) s += sqrt((double)nd[j]); /* k = 0 is the point itself */
        mean_dist[i] = (float)(s / mean_k);
        ++valid;
    }
    double sum = 0, sq = 0;
    /* PCL: sq_sum += distances[i] * distances[i] -- a float product (rounded to float), widened for the sum */
    for (size_t i = 0; i < n; ++i) { const float f = mean_dist[i]; sum += f; sq += (double)(f * f); }
    double mean = sum / (double)valid;
    double var = (sq - sum * sum / (double)valid) / ((double)valid - 1);
    double thr = mean + stddev_mult * sqrt(var);
    if (thresh) *thresh = thr;
    size_t kept = 0;
    for (size_t i = 0; i < n; ++i) {
        inlier[i] = !(mean_dist[i] > thr);
        kept += inlier[i];
    }
    free(ni); free(nd);
    orc_kdtree_free(t);
    return kept;
}

/* ================= ICP (src/comparator.cpp:1089-1110, 9.5) ================= */

void orc_transform(const float T[16], const void *src, size_t n, size_t sstride, float *dst) {
    for (size_t i = 0; i < n; ++i) {
        const float *p = pt_at(src, sstride, i);
        float x = p[0], y = p[1], z = p[2];
        /* pcl::transformPointCloud: ((m0*x + m1*y) + m2*z) + m3, unfused floats */
        dst[i * 3 + 0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
        dst[i * 3 + 1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
        dst[i * 3 + 2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
    }
}

void orc_icp_step_sums(const orc_kdtree *tree, const void *tgt, size_t tstride,
                       const void *src, size_t n, size_t sstride, int32_t *idx, float *d2,
                       double sums[17]) {
    for (int i = 0; i < 17; ++i) sums[i] = 0;
    orc_kdtree_nn1_batch(tree, src, n, sstride, idx, d2);
    for (size_t i = 0; i < n; ++i) {
        if (idx[i] < 0) continue;
        const float *p = pt_at(src, sstride, i);
        const float *q = pt_at(tgt, tstride, (size_t)idx[i]);
        for (int a = 0; a < 3; ++a) { sums[a] += p[a]; sums[3 + a] += q[a]; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) sums[6 + r * 3 + c] += (double)q[r] * (double)p[c];
        sums[15] += d2[i];
        sums[16] += 1.0;
    }
}

/* 3x3 one-sided Jacobi SVD in double: A = U diag(s) V^T */
static void svd3(const double A[9], double U[9], double s[3], double V[9]) {
    double B[9];
    memcpy(B, A, sizeof(B));
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, c = 0;
                for (int r = 0; r < 3; ++r) {
                    a += B[r * 3 + p] * B[r * 3 + p];
                    b += B[r * 3 + q] * B[r * 3 + q];
                    c += B[r * 3 + p] * B[r * 3 + q];
                }
                off += fabs(c);
                if (fabs(c) < 1e-300 || fabs(c) <= 1e-17 * sqrt(a * b)) continue;
                double zeta = (b - a) / (2.0 * c);
                double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int r = 0; r < 3; ++r) {
                    double bp = B[r * 3 + p], bq = B[r * 3 + q];
                    B[r * 3 + p] = cs * bp - sn * bq;
                    B[r * 3 + q] = sn * bp + cs * bq;
                    double vp = V[r * 3 + p], vq = V[r * 3 + q];
                    V[r * 3 + p] = cs * vp - sn * vq;
                    V[r * 3 + q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-300) break;
    }
    for (int c = 0; c < 3; ++c) {
        double nrm = 0;
        for (int r = 0; r < 3; ++r) nrm += B[r * 3 + c] * B[r * 3 + c];
        s[c] = sqrt(nrm);
    }
    /* sort descending */
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (s[order[j]] > s[order[i]]) { int tmp = order[i]; order[i] = order[j]; order[j] = tmp; }
    double Us[9], Vs[9], ss[3];
    for (int c = 0; c < 3; ++c) {
        int o = order[c];
        ss[c] = s[o];
        for (int r = 0; r < 3; ++r) {
            Us[r * 3 + c] = s[o] > 1e-300 ? B[r * 3 + o] / s[o] : 0.0;
            Vs[r * 3 + c] = V[r * 3 + o];
        }
    }
    /* complete U for rank-deficient cases: third column = cross of first two */
    if (ss[2] <= 1e-300 * (ss[0] > 0 ? ss[0] : 1)) {
        Us[2] = Us[3] * Us[7] - Us[6] * Us[4];
        Us[5] = Us[6] * Us[1] - Us[0] * Us[7];
        Us[8] = Us[0] * Us[4] - Us[3] * Us[1];
    }
    memcpy(U, Us, sizeof(Us)); memcpy(V, Vs, sizeof(Vs)); memcpy(s, ss, sizeof(ss));
}
static double det3(const double M[9]) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* rotation and translation from the covariance sigma = (1/n) sum (q - qm)(p - pm)^T and the two means */
static void umeyama_core(const double S[9], const double pm[3], const double qm[3], float T[16]);

int orc_umeyama_from_sums(const double sums[17], float T[16]) {
    double n = sums[16];
    if (n < 3) return -1; /* min_number_correspondences_ = 3 (9.5) */
    double pm[3], qm[3], S[9];
    for (int a = 0; a < 3; ++a) { pm[a] = sums[a] / n; qm[a] = sums[3 + a] / n; }
    /* sigma = (1/n) sum (q - qm)(p - pm)^T = (1/n) sum q p^T - qm pm^T */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S[r * 3 + c] = sums[6 + r * 3 + c] / n - qm[r] * pm[c];
    umeyama_core(S, pm, qm, T);
    return 0;
}

static void umeyama_core(const double S[9], const double pm[3], const double qm[3], float T[16]) {
    double U[9], V[9], sv[3];
    svd3(S, U, sv, V);
    double d[3] = {1, 1, 1};
    if (det3(U) * det3(V) < 0) d[2] = -1; /* reflection fix (Umeyama eq. 39-43) */
    double R[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += U[r * 3 + k] * d[k] * V[c * 3 + k];
            R[r * 3 + c] = acc;
        }
    for (int r = 0; r < 3; ++r) {
        double tr = qm[r];
        for (int c = 0; c < 3; ++c) {
            T[r * 4 + c] = (float)R[r * 3 + c];
            tr -= R[r * 3 + c] * pm[c];
        }
        T[r * 4 + 3] = (float)tr;
    }
    T[12] = T[13] = T[14] = 0.0f;
    T[15] = 1.0f;
}

static void mat4_mul(const float A[16], const float B[16], float C[16]) {
    float R[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float acc = 0;
            for (int k = 0; k < 4; ++k) acc += A[r * 4 + k] * B[k * 4 + c];
            R[r * 4 + c] = acc;
        }
    memcpy(C, R, sizeof(R));
}

int orc_icp(const void *src, size_t n, size_t sstride, const void *tgt, size_t m,
            size_t tstride, int max_iter, int fixed, float T[16], double *fitness,
            int32_t *corr_idx, double *iter_mse) {
    orc_kdtree *tree = orc_kdtree_build(tgt, m, tstride);
    float *cur = (float *)malloc(sizeof(float) * 3 * (n ? n : 1));
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    float *d2 = (float *)malloc(sizeof(float) * (n ? n : 1));
    float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(T, I, sizeof(I));
    orc_transform(I, src, n, sstride, cur); /* input_transformed = *input_ */
    int it = 0;
    double prev_mse = DBL_MAX;
    while (tree && it < max_iter) {
        double sums[17];
        orc_icp_step_sums(tree, tgt, tstride, cur, n, 12, idx, d2, sums);
        float Ti[16];
        if (sums[16] < 3) break; /* min_number_correspondences_ */
        {
            /* Eigen::umeyama demeans the matched pairs before it forms the covariance (two passes); the one-pass
               sum q p^T - n qm pm^T of orc_umeyama_from_sums cancels for a small cloud far from the origin */
            double pm[3] = {0, 0, 0}, qm[3] = {0, 0, 0}, S[9] = {0};
            for (size_t i = 0; i < n; ++i) {
                if (idx[i] < 0) continue;
                const float *q = pt_at(tgt, tstride, (size_t)idx[i]);
                for (int a = 0; a < 3; ++a) { pm[a] += cur[3 * i + a]; qm[a] += q[a]; }
            }
            for (int a = 0; a < 3; ++a) { pm[a] /= sums[16]; qm[a] /= sums[16]; }
            for (size_t i = 0; i < n; ++i) {
                if (idx[i] < 0) continue;
                const float *q = pt_at(tgt, tstride, (size_t)idx[i]);
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) S[r * 3 + c] += ((double)q[r] - qm[r]) * ((double)cur[3 * i + c] - pm[c]);
            }
            for (int k = 0; k < 9; ++k) S[k] /= sums[16];
            umeyama_core(S, pm, qm, Ti);
        }
        orc_transform(Ti, cur, n, 12, cur);
        mat4_mul(Ti, T, T); /* final = T * final */
        double mse = sums[15] / sums[16];
        if (iter_mse) iter_mse[it] = mse;
        ++it;
        if (!fixed) { /* DefaultConvergenceCriteria, absolute MSE 1e-12 */
            if (fabs(mse - prev_mse) < 1e-12) break;
        }
        prev_mse = mse;
    }
    if (corr_idx) memcpy(corr_idx, idx, sizeof(int32_t) * n);
    /* getFitnessScore: transform input with the final matrix, one more NN pass */
    if (fitness) {
        orc_transform(T, src, n, sstride, cur);
        double score = 0;
        size_t nr = 0;
        if (tree) {
            orc_kdtree_nn1_batch(tree, cur, n, 12, idx, d2);
            for (size_t i = 0; i < n; ++i)
                if (idx[i] >= 0) { score += d2[i]; ++nr; }
        }
        *fitness = nr ? score / (double)nr : DBL_MAX;
    }
    free(cur); free(idx); free(d2);
    orc_kdtree_free(tree);
    return it;
}

/* ======== RegionGrowingRGB (src/segmentation.cpp:161-216: color_growing_segmentation) [PCL 1.7, recalled] =========
 * TEST INFRASTRUCTURE.  pcl::RegionGrowingRGB::extract as recalled from PCL 1.7's region_growing.hpp and
 * region_growing_rgb.hpp (SURVEY.md has no section for it; parity unpinned like everything PCL here):
 *   findPointNeighbours      nearestKSearch(i, region_neighbour_number_ = 100) per point (rows given: nbr / nbr_d2, or
 *                            taken from this file's kd-tree when nbr == NULL)
 *   applySmoothRegionGrowing seeds in index order (no normals: residual 0), growRegion walks the first neighbour_number_
 *                            (30) neighbours, validatePoint = squared colour distance to the CURRENT point <= p2p^2;
 *                            every joined point is a seed
 *   findSegmentNeighbours    per segment the 100 nearest other segments by min neighbour distance (max-heap of pairs,
 *                            handed over farthest first)
 *   applyRegionMerging       mean colours (float sums in index order / count, truncated to unsigned), homogeneous regions
 *                            (distance <= dist^2, colour difference < r2r^2), small regions folded into the region of their
 *                            nearest neighbouring segment; std::sort of equal distances is unspecified in PCL: stable here
 *   clusters outside [min, max] dropped.  labels[i] = index of the point's cluster in the returned order, or -1. */
typedef struct { float d; int32_t s; } rgb_pair_t;
static void rgb_stable_sort(rgb_pair_t *a, size_t n) { /* insertion sort: stable, lists are short */
    for (size_t i = 1; i < n; ++i) {
        rgb_pair_t v = a[i];
        size_t j = i;
        while (j > 0 && a[j - 1].d > v.d) { a[j] = a[j - 1]; --j; }
        a[j] = v;
    }
}
static int rgb_pair_greater(rgb_pair_t a, rgb_pair_t b) { return a.d > b.d || (a.d == b.d && a.s > b.s); }

int orc_region_growing_rgb(const void *pts, size_t n, size_t stride, const uint8_t *rgb /* n x 3: r g b */,
                           const int32_t *nbr_in, const float *nbr_d2_in, int k_rows, float dist_thr, float p2p_thr,
                           float r2r_thr, int min_size, int max_size, int nn, int region_nn, int32_t *labels) {
    for (size_t i = 0; i < n; ++i) labels[i] = -1;
    if (n == 0 || nn <= 0 || region_nn <= 0) return 0;
    const float dist2 = dist_thr * dist_thr, p2p2 = p2p_thr * p2p_thr, r2r2 = r2r_thr * r2r_thr;
    /* --- findPointNeighbours */
    int K = k_rows;
    int32_t *nbr = NULL;
    float *nd2 = NULL;
    if (nbr_in) {
        nbr = (int32_t *)nbr_in;
        nd2 = (float *)nbr_d2_in;
    } else {
        K = (size_t)region_nn < n ? region_nn : (int)n;
        nbr = (int32_t *)malloc(sizeof(int32_t) * n * (size_t)K);
        nd2 = (float *)malloc(sizeof(float) * n * (size_t)K);
        orc_kdtree *t = orc_kdtree_build(pts, n, stride);
        for (size_t i = 0; i < n; ++i) {
            const float *q = (const float *)((const char *)pts + i * stride);
            for (int j = 0; j < K; ++j) { nbr[i * K + j] = -1; nd2[i * K + j] = 0.f; }
            if (t) orc_kdtree_knn(t, q, K, nbr + i * (size_t)K, nd2 + i * (size_t)K);
        }
        orc_kdtree_free(t);
    }
    /* --- applySmoothRegionGrowingAlgorithm / growRegion */
    int32_t *seg = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *queue = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *seg_pts = (int32_t *)malloc(sizeof(int32_t) * n);
    for (size_t i = 0; i < n; ++i) seg[i] = -1;
    int ns = 0;
    for (size_t s0 = 0; s0 < n; ++s0) {
        if (seg[s0] != -1) continue;
        size_t head = 0, tail = 0;
        queue[tail++] = (int32_t)s0;
        seg[s0] = ns;
        int cnt = 1;
        while (head < tail) {
            const int32_t cur = queue[head++];
            for (int j = 0; j < nn && j < K; ++j) {
                const int32_t v = nbr[(size_t)cur * K + j];
                if (v < 0 || seg[v] != -1) continue;
                unsigned int diff = 0;
                for (int c = 0; c < 3; ++c) {
                    const unsigned int a = rgb[(size_t)cur * 3 + c], b = rgb[(size_t)v * 3 + c];
                    diff += (a - b) * (a - b); /* unsigned arithmetic, as PCL's std::vector<unsigned int> colours */
                }
                if ((float)diff > p2p2) continue;
                seg[v] = ns;
                ++cnt;
                queue[tail++] = v;
            }
        }
        seg_pts[ns++] = cnt;
    }
    /* --- findSegmentNeighbours / findRegionsKNN */
    int32_t *first = (int32_t *)calloc((size_t)ns + 1, sizeof(int32_t)); /* CSR of the segments' members, index order */
    for (size_t i = 0; i < n; ++i) first[seg[i] + 1]++;
    for (int s = 0; s < ns; ++s) first[s + 1] += first[s];
    int32_t *members = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)ns);
    for (int s = 0; s < ns; ++s) fill[s] = first[s];
    for (size_t i = 0; i < n; ++i) members[fill[seg[i]]++] = (int32_t)i;
    rgb_pair_t **snb = (rgb_pair_t **)calloc((size_t)ns, sizeof(rgb_pair_t *)); /* neighbour lists, farthest first */
    int *snb_n = (int *)calloc((size_t)ns, sizeof(int));
    float *dmin = (float *)malloc(sizeof(float) * (size_t)ns);
    rgb_pair_t *heap = (rgb_pair_t *)malloc(sizeof(rgb_pair_t) * ((size_t)region_nn + 2));
    for (int s = 0; s < ns; ++s) dmin[s] = FLT_MAX;
    for (int s = 0; s < ns; ++s) {
        for (int32_t m = first[s]; m < first[s + 1]; ++m) {
            const int32_t p = members[m];
            for (int j = 0; j < K; ++j) {
                const int32_t v = nbr[(size_t)p * K + j];
                if (v < 0) continue;
                const int32_t t = seg[v];
                if (t != s && dmin[t] > nd2[(size_t)p * K + j]) dmin[t] = nd2[(size_t)p * K + j];
            }
        }
        /* the region_nn smallest (distance, segment) pairs: a bounded max-heap kept as a sorted array (largest first) */
        int hn = 0;
        for (int t = 0; t < ns; ++t) {
            if (!(dmin[t] < FLT_MAX)) continue;
            rgb_pair_t e = {dmin[t], t};
            dmin[t] = FLT_MAX;
            int pos = hn;
            while (pos > 0 && rgb_pair_greater(e, heap[pos - 1])) { heap[pos] = heap[pos - 1]; --pos; }
            heap[pos] = e;
            ++hn;
            if (hn > region_nn) { memmove(heap, heap + 1, sizeof(rgb_pair_t) * (size_t)(hn - 1)); --hn; } /* pop the largest */
        }
        snb[s] = (rgb_pair_t *)malloc(sizeof(rgb_pair_t) * (size_t)(hn > 0 ? hn : 1));
        memcpy(snb[s], heap, sizeof(rgb_pair_t) * (size_t)hn);
        snb_n[s] = hn;
    }
    /* --- applyRegionMergingAlgorithm */
    /* segment colours: PCL sums the channels in std::vector<unsigned int> (exact), then
     * static_cast<unsigned int>(static_cast<float>(sum) / static_cast<float>(num_pts_in_segment_)) */
    float *col = (float *)calloc((size_t)ns * 3, sizeof(float));
    unsigned int *csum = (unsigned int *)calloc((size_t)ns * 3, sizeof(unsigned int));
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) csum[(size_t)seg[i] * 3 + c] += rgb[i * 3 + c];
    for (int s = 0; s < ns; ++s)
        for (int c = 0; c < 3; ++c) col[(size_t)s * 3 + c] = (float)(unsigned int)((float)csum[(size_t)s * 3 + c] / (float)seg_pts[s]);
    free(csum);
    int32_t *sreg = (int32_t *)malloc(sizeof(int32_t) * (size_t)ns);
    unsigned int *rpts = (unsigned int *)calloc((size_t)ns, sizeof(unsigned int));
    for (int s = 0; s < ns; ++s) sreg[s] = -1;
    int nr = 0;
    for (int s = 0; s < ns; ++s) {
        if (sreg[s] == -1) { sreg[s] = nr; rpts[nr] = (unsigned int)seg_pts[s]; ++nr; }
        const int cur = sreg[s];
        for (int j = 0; j < region_nn && j < snb_n[s]; ++j) {
            const int32_t t = snb[s][j].s;
            if (snb[s][j].d > dist2) continue;
            if (sreg[t] != -1) continue;
            float diff = 0.f;
            for (int c = 0; c < 3; ++c) {
                const float d = col[(size_t)s * 3 + c] - col[(size_t)t * 3 + c];
                diff += d * d;
            }
            if (diff < r2r2) { sreg[t] = cur; rpts[cur] += (unsigned int)seg_pts[t]; }
        }
    }
    /* region neighbour lists (findRegionNeighbours): dynamic arrays of pairs */
    rgb_pair_t **rnb = (rgb_pair_t **)calloc((size_t)nr, sizeof(rgb_pair_t *));
    size_t *rnb_n = (size_t *)calloc((size_t)nr, sizeof(size_t)), *rnb_cap = (size_t *)calloc((size_t)nr, sizeof(size_t));
#define RNB_PUSH(r, e) do { if (rnb_n[r] == rnb_cap[r]) { rnb_cap[r] = rnb_cap[r] ? rnb_cap[r] * 2 : 16; rnb[r] = (rgb_pair_t *)realloc(rnb[r], sizeof(rgb_pair_t) * rnb_cap[r]); } rnb[r][rnb_n[r]++] = (e); } while (0)
    for (int s = 0; s < ns; ++s) /* segments in index order = the order final_segments lists them per region */
        for (int j = 0; j < snb_n[s]; ++j) {
            if (snb[s][j].d == FLT_MAX) continue;
            if (sreg[snb[s][j].s] != sreg[s]) RNB_PUSH(sreg[s], snb[s][j]);
        }
    for (int r = 0; r < nr; ++r) rgb_stable_sort(rnb[r], rnb_n[r]);
    for (int r = 0; r < nr; ++r) {
        if (rpts[r] >= (unsigned int)min_size) continue;
        if (rnb_n[r] == 0 || rnb[r][0].d == FLT_MAX) continue;
        const int into = sreg[rnb[r][0].s];
        for (int s = 0; s < ns; ++s)
            if (sreg[s] == r) sreg[s] = into; /* (order inside a region's segment list does not matter below) */
        rpts[into] += rpts[r];
        rpts[r] = 0;
        for (size_t j = 0; j < rnb_n[into]; ++j)
            if (sreg[rnb[into][j].s] == into) { rnb[into][j].d = FLT_MAX; rnb[into][j].s = 0; }
        for (size_t j = 0; j < rnb_n[r]; ++j)
            if (sreg[rnb[r][j].s] != into) RNB_PUSH(into, rnb[r][j]);
        rnb_n[r] = 0;
        rgb_stable_sort(rnb[into], rnb_n[into]);
    }
#undef RNB_PUSH
    /* clusters: regions in index order, empty ones dropped, sizes outside [min, max] dropped */
    int32_t *rid = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nr > 0 ? nr : 1));
    int ncl = 0;
    for (int r = 0; r < nr; ++r) rid[r] = (rpts[r] > 0 && (int)rpts[r] >= min_size && (int)rpts[r] <= max_size) ? ncl++ : -1;
    for (size_t i = 0; i < n; ++i) labels[i] = rid[sreg[seg[i]]];
    for (int s = 0; s < ns; ++s) free(snb[s]);
    for (int r = 0; r < nr; ++r) free(rnb[r]);
    free(rid); free(rnb); free(rnb_n); free(rnb_cap); free(rpts); free(sreg); free(col); free(heap); free(dmin);
    free(snb_n); free(snb); free(fill); free(members); free(first); free(seg_pts); free(queue); free(seg);
    if (!nbr_in) { free(nbr); free(nd2); }
    return ncl;
}
