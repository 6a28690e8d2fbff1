/*
 * pcc_oracle.h -- CPU restatement of the reference's nearest-neighbour path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pointcloudcomparator_amd/ (the
 * product) may include, link or load this.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * PARITY UNPINNED: the reference (adr-arroyo/PointCloudComparator) ships no
 * tests, golden vectors or fixtures for this path, and the arithmetic lives in
 * PCL 1.7 / FLANN 1.8.4, which are not vendored under /root/reference and not
 * installed here.  This file restates FLANN's published single-kd-tree
 * algorithm (KDTreeSingleIndex, L2_Simple<float>, KNNSimpleResultSet,
 * RadiusResultSet) and the PCL wrappers around it as described in SURVEY.md
 * section 9, anchored on the reference's call sites:
 *   src/comparator.cpp:560-588   matchRIFTFeaturesKnn  (k = 1)
 *   src/comparator.cpp:1089-1110 performICP            (k = 1, <= 20 iterations + fitness)
 *   src/comparator.cpp:1520-1549 StatisticalOutlierRemoval (k = 51)
 *   src/segmentation.cpp:119-131 EuclideanClusterExtraction (radius 0.05)
 */
#ifndef PCC_ORACLE_H
#define PCC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- exhaustive definitional oracle (lowest-index tie-break) ------------- */
/* points are AoS with a byte stride; the first three floats are x,y,z.
 * Reference points with a non-finite coordinate are skipped (PCL
 * KdTreeFLANN::convertCloudToArray); indices are ORIGINAL indices.
 * d2 = ((dx*dx)+dy*dy)+dz*dz, each op rounded to fp32 (FLANN L2_Simple, 9.2).
 * A non-finite query yields idx=-1, d2=+inf. */
void orc_nn1_exhaustive(const void *ref, size_t m, size_t rstride,
                        const void *qry, size_t n, size_t qstride,
                        int32_t *idx, float *d2);
/* k nearest, ascending by (d2, idx); rows padded with idx=-1,d2=+inf when
 * fewer than k valid references exist.  Returns min(k, valid refs). */
int orc_knn_exhaustive(const void *ref, size_t m, size_t rstride,
                       const void *qry, size_t n, size_t qstride, int k,
                       int32_t *idx, float *d2);
/* strict d2 < r2 neighbour counts per query */
void orc_radius_count_exhaustive(const void *ref, size_t m, size_t rstride,
                                 const void *qry, size_t n, size_t qstride,
                                 float r2, int32_t *counts);

/* ---- FLANN KDTreeSingleIndex restatement ---------------- */
typedef struct orc_kdtree orc_kdtree;
/* Builds on the finite points of the cloud (leaf_max_size 15, middle split,
 * reorder).  Returns NULL when no valid point exists (PCL prints an error). */
/* split rule of the trees built from now on: 0 middleSplit_ (default), 1 middleSplit, 2 middleSplit_ with the loop
 * variable in its selection loop (pcc_oracle.c) */
void orc_set_split_rule(int rule);
int orc_get_split_rule(void);
orc_kdtree *orc_kdtree_build(const void *pts, size_t m, size_t stride);
void orc_kdtree_free(orc_kdtree *t);
size_t orc_kdtree_size(const orc_kdtree *t);
/* pcl::KdTreeFLANN::nearestKSearch for one query: returns number found
 * (k clamped to the number of valid points), ascending distance,
 * first-visited wins among exact ties (KNNSimpleResultSet). */
int orc_kdtree_knn(const orc_kdtree *t, const float q[3], int k,
                   int32_t *idx, float *d2);
/* pcl::KdTreeFLANN::radiusSearch(point, radius, ..., max_nn = 0): r2 is
 * float(radius*radius) evaluated in double by the caller; results sorted by
 * (d2, idx) when sorted != 0.  Returns the neighbour count; writes at most
 * cap entries. */
int orc_kdtree_radius(const orc_kdtree *t, const float q[3], float r2,
                      int sorted, int32_t *idx, float *d2, int cap);
/* batch helpers (the reference's per-query loops, src/comparator.cpp:571-577) */
void orc_kdtree_nn1_batch(const orc_kdtree *t, const void *qry, size_t n,
                          size_t qstride, int32_t *idx, float *d2);
/* same, spread over nthreads host threads (reported-only CPU baseline) */
void orc_kdtree_nn1_batch_mt(const orc_kdtree *t, const void *qry, size_t n,
                             size_t qstride, int32_t *idx, float *d2,
                             int nthreads);

/* ---- PCL algorithm restatements on top of the tree ----------------------- */
/* matchRIFTFeaturesKnn (src/comparator.cpp:560-588): tree on des1, one k=1
 * query per element of des2, keep when found==1 && d2 < 0.05f.  out[0] is the
 * dummy 0 the reference's vector starts with (:568).  Returns size() (= 1 +
 * matches); out needs n2+1 slots. */
int orc_match_rift_knn(const void *des1, size_t n1, const void *des2, size_t n2,
                       size_t stride, int32_t *out);

/* the keypoint snap loop of processRIFTwithSIFT (src/comparator.cpp:696-713): for each query the FIRST
 * j with sqrt(pow(qx-px,2)+pow(qy-py,2)+pow(qz-pz,2)) < radius (float differences, double math);
 * -1 when none. */
void orc_first_within(const void *pts, size_t m, size_t stride, const void *qry, size_t n, size_t qstride,
                      double radius, int32_t *idx);

/* pcl::NormalEstimation<PointT, pcl::Normal> with setKSearch(k) and the default viewpoint (0,0,0)
 * (src/segmentation.cpp:232-241, K = 50) [recalled from PCL 1.7 features/normal_3d.h, common/centroid.hpp,
 * common/eigen.hpp]: per point the k nearest neighbours (itself included), single-pass float
 * mean/covariance (accu[9] / n, cov = E[xx] - E[x]E[x]), smallest eigenpair by pcl::eigen33 (closed-form
 * roots), curvature = |lambda0 / trace|, normal flipped towards the viewpoint.  out: n x 4 floats
 * (nx, ny, nz, curvature); NaN for non-finite points or fewer than 3 neighbours. */
void orc_normals(const void *pts, size_t n, size_t stride, int k, const float vp[3], float *out);
/* the same with setRadiusSearch(radius): neighbours = sorted radiusSearch result (src/comparator.cpp:628-635) */
void orc_normals_radius(const void *pts, size_t n, size_t stride, double radius, const float vp[3], float *out);
/* the same arithmetic on given neighbour lists (n rows of k indices, -1 = unused) */
void orc_normals_from_neighbours(const void *pts, size_t n, size_t stride, const int32_t *nbr, int k,
                                 const float vp[3], float *out);

/* pcl::SACSegmentation<PointT>::segment with SACMODEL_PLANE, SAC_RANSAC, probability 0.99 default
 * (src/segmentation.cpp:79-99: optimize on, 100 iterations, threshold 0.02).  RANSAC with PCL's fixed
 * seed (mt19937(12345), boost uniform_int<>(0, INT_MAX) = eng()/2, partial Fisher-Yates over the persistent
 * shuffled index array), adaptive iteration bound k = log(1-p)/log(1-w^3), inliers |c . (x,y,z,1)| <
 * threshold in ascending index order; with optimize != 0 the least-squares refit (float mean/covariance,
 * pcl::eigen33) and a second inlier selection.  Returns the inlier count (0: no model). */
long orc_sac_plane(const void *pts, size_t n, size_t stride, int max_iterations, double threshold, double probability,
                   int optimize, int32_t *inliers, float coeff[4], int *iterations_out);
void orc_mt19937_raw(uint32_t seed, uint32_t *out, size_t n);

/* pcl::RegionGrowing<PointT, pcl::Normal>::extract (src/segmentation.cpp:259-271: min 50, max 1e6,
 * 100 neighbours, smoothness 3 deg, curvature threshold 1) [recalled from PCL 1.7
 * segmentation/impl/region_growing.hpp]: points sorted by curvature, regions grown through the
 * precomputed neighbour lists while |n_seed . n_nbr| >= cos(theta); a neighbour becomes a seed when its
 * curvature <= the threshold.  nbr: n rows of k indices (ascending distance, itself first).
 * labels[i] = index of the kept cluster (in PCL's output order) or -1.  Returns the cluster count. */
int orc_region_growing(size_t n, const float *normals /* n x 4 */, const int32_t *nbr, int k,
                       float smoothness, float curvature_threshold, int min_size, int max_size,
                       int32_t *labels);

/* pcl::VoxelGrid<PointXYZRGB>::applyFilter with setLeafSize(leaf, leaf, leaf), downsample_all_data
 * (src/segmentation.cpp:69-74, 224-229; pcl/filters/impl/voxel_grid.hpp [recalled]): voxel index
 * floor(p*inv) - min_b on the world-aligned lattice, points grouped by index (ascending), centroid =
 * float sum / float count; rgb (packed word at byte 16 when has_rgb) averaged per channel in float and
 * truncated.  PCL's std::sort leaves the order inside a voxel unspecified; this restatement adds in
 * ascending point index.  out has m*out_stride bytes; returns the number of voxels, or -1 when the
 * lattice would overflow int32 (PCL then returns the input unchanged). */
long orc_voxel_grid(const void *pts, size_t m, size_t stride, float leaf, int has_rgb, void *out, size_t out_stride);

/* pcl::extractEuclideanClusters + EuclideanClusterExtraction::extract
 * (src/segmentation.cpp:125-131).  labels[i] = cluster id in the
 * returned order (size-descending, ties by lowest member index; -1 = not in
 * any kept cluster).  cluster_sizes gets up to max_clusters sizes.
 * Returns the number of kept clusters. */
int orc_euclidean_clusters(const void *pts, size_t m, size_t stride,
                           float tolerance, uint32_t min_size, uint32_t max_size,
                           int32_t *labels, int32_t *cluster_sizes,
                           int max_clusters);

/* pcl::StatisticalOutlierRemoval::applyFilterIndices (src/comparator.cpp:
 * 1523-1541): mean_dist[i] (float), inlier[i] (0/1), *thresh.
 * Returns the number of inliers. */
size_t orc_sor(const void *pts, size_t n, size_t stride, int mean_k,
               double stddev_mult, float *mean_dist, uint8_t *inlier,
               double *thresh);

/* pcl::IterativeClosestPoint (src/comparator.cpp:1089-1110) with
 * fixed iteration count (no early exit when fixed != 0).  T is the final 4x4
 * row-major float transform source->target.  corr_idx (optional) receives
 * the correspondence indices of the LAST iteration.  iter_mse (optional, len
 * max_iter) receives the per-iteration mean squared correspondence distance.
 * Returns iterations done; *fitness = getFitnessScore(). */
int orc_icp(const void *src, size_t n, size_t sstride,
            const void *tgt, size_t m, size_t tstride,
            int max_iter, int fixed, float T[16], double *fitness,
            int32_t *corr_idx, double *iter_mse);

/* one ICP building block on identical inputs: NN of every (already
 * transformed) source point + the double-precision sums Umeyama needs.
 * sums[0..2]=sum p, [3..5]=sum q, [6..14]=sum q p^T (row-major 3x3:
 * q_r * p_c), [15]=sum d2, [16]=count. */
void orc_icp_step_sums(const orc_kdtree *tgt_tree, const void *tgt, size_t tstride,
                       const void *src, size_t n, size_t sstride,
                       int32_t *idx, float *d2, double sums[17]);
/* rigid transform (no scaling) from the sums above, Umeyama/Kabsch in double,
 * cast to float 4x4 row-major.  Returns 0 on success. */
int orc_umeyama_from_sums(const double sums[17], float T[16]);
/* pcl::transformPointCloud arithmetic: ((m0*x + m1*y) + m2*z) + m3, unfused */
void orc_transform(const float T[16], const void *src, size_t n, size_t sstride,
                   float *dst_xyz /* n*3 packed */);

#ifdef __cplusplus
}
#endif
#endif

/* pcl::RegionGrowingRGB::extract as color_growing_segmentation configures it (src/segmentation.cpp:161-216; recalled from
 * PCL 1.7, see pcc_oracle.c).  rgb: n x 3 bytes (r, g, b).  nbr / nbr_d2: rows of k_rows neighbours per point (ascending,
 * -1 padded) or NULL to search them with this file's kd-tree.  labels[i] = cluster of point i in PCL's cluster order, -1
 * when its cluster was dropped.  Returns the number of clusters. */
int orc_region_growing_rgb(const void *pts, size_t n, size_t stride, const uint8_t *rgb, const int32_t *nbr, const float *nbr_d2,
                           int k_rows, float dist_thr, float p2p_thr, float r2r_thr, int min_size, int max_size, int nn,
                           int region_nn, int32_t *labels);
