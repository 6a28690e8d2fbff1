/*
 * pcc_nn.h -- C-ABI of libpcc_nn: the MI355X (gfx950) nearest-neighbour /
 * radius-search engine that replaces the pcl::KdTreeFLANN path of
 * adr-arroyo/PointCloudComparator.
 *
 * The reference has no FFI layer: it calls PCL C++ classes one query at a time
 * (SURVEY.md 8b).  Each entry point below names the reference interface it
 * replaces (file:line relative to the reference tree).  Everything is plain C:
 * opaque handle, pointers, sizes, int status.  No allocation crosses the ABI;
 * outputs are caller-allocated.
 *
 * Memory spaces: every data pointer is tagged by a `mem` argument --
 * PCC_MEM_HOST (library copies H2D/D2H itself) or PCC_MEM_DEVICE (pointer is
 * HBM on the index's device; results are written to device memory on the
 * index's stream and the call returns without synchronising).
 *
 * Points are AoS with a byte stride; the first `dim` floats of each element
 * are the coordinates.  dim must be 3 -- that is what every hot call site of
 * the reference searches on (pcl::PointXYZRGB 32 B, pcl::PointXYZ 16 B, and
 * pcl::Histogram<32> 128 B through PCL's 3-float DefaultPointRepresentation,
 * SURVEY.md 3.2 / 9.1).
 *
 * Semantics shared by all searches:
 *   - reference points with a non-finite coordinate are skipped; returned
 *     indices are positions in the ORIGINAL cloud (PCL index_mapping_);
 *   - d2 = ((dx*dx)+dy*dy)+dz*dz in fp32, every op rounded separately
 *     (FLANN L2_Simple<float>); distances are SQUARED;
 *   - exact-distance ties resolve to the LOWEST original index;
 *   - a non-finite query never aborts: idx = -1, d2 = +inf, count = 0.
 *   - a squared distance that overflows float (>= FLT_MAX) is no neighbour, as in FLANN (its result sets start
 *     with worst_distance = FLT_MAX and reject dist >= worst): idx = -1, d2 = +inf.
 */
#ifndef PCC_NN_H
#define PCC_NN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PCC_VERSION 100

typedef struct pcc_index pcc_index;

enum pcc_status {
    PCC_OK = 0,
    PCC_ERR_INVALID = -1,     /* bad argument */
    PCC_ERR_EMPTY = -2,       /* no valid reference point ("Cannot create a KDTree with an empty input cloud") */
    PCC_ERR_DEVICE = -3,      /* HIP runtime failure (no GPU, launch error, ...) */
    PCC_ERR_NOMEM = -4,
    PCC_ERR_UNSUPPORTED = -5, /* dim != 3, k too large, ... */
    PCC_ERR_OVERFLOW = -6     /* caller-provided output capacity too small */
};
enum pcc_mem { PCC_MEM_HOST = 0, PCC_MEM_DEVICE = 1 };
/* search engine behind an index.  Both are exact and return identical bits.
 *   BRUTE: tiled exhaustive scan (query tile in registers, reference tile in LDS)
 *   GRID : cell-sorted references, exact ring search with conservative bounds,
 *          BRUTE fallback for queries the rings do not resolve
 *   AUTO : GRID when the cloud is large enough to amortise its build. */
enum pcc_engine { PCC_ENGINE_AUTO = 0, PCC_ENGINE_BRUTE = 1, PCC_ENGINE_GRID = 2 };
/* which of several references at EXACTLY the same distance a k = 1 search names.
 *   LOWEST_INDEX: the lowest original index (default; what every engine computes on the GPU)
 *   FLANN       : the one pcl::KdTreeFLANN returns -- the first its kd-tree walk reaches.  The GPU
 *                 result is kept for every query whose minimiser is unique; queries with a second reference at
 *                 the same distance are flagged on the GPU and only those are walked through a host-side
 *                 restatement of FLANN's KDTreeSingleIndex (built once per indexed cloud).  Applies to pcc_nn1
 *                 and pcc_match_knn (the call sites that hand indices on, src/comparator.cpp:576-580); distances
 *                 are the same bits either way. */
enum pcc_ties { PCC_TIES_LOWEST_INDEX = 0, PCC_TIES_FLANN = 1 };

#define PCC_KNN_MAX_K 65536 /* k <= 512: wave-cooperative selection; larger k works, one lane per query with O(k) insertion */

int pcc_version(void);
/* thread-local message for the last non-OK status returned on this thread */
const char *pcc_last_error(void);
int pcc_device_count(int *count);

/* ---- index lifetime -------------------------------------------------------
 * replaces: pcl::KdTreeFLANN<T> ctor + setInputCloud (src/comparator.cpp:564-565),
 *           pcl::search::KdTree<T>::setInputCloud (src/segmentation.cpp:120-122)
 *           and the trees ICP / SOR / EC build internally
 *           (src/comparator.cpp:1096,1527,1541; src/segmentation.cpp:131).
 * Copies (uploads) the cloud; the caller may free `pts` after the call. */
int pcc_index_create(const void *pts, size_t n, size_t stride_bytes, int dim,
                     int mem, int device, int engine, pcc_index **out);
int pcc_index_destroy(pcc_index *index);
/* KdTreeFLANN::setInputCloud on an existing object (cleanup + rebuild):
 * replaces the indexed cloud, reusing the handle's device allocations.  On error
 * the index is left empty (searches return PCC_ERR_EMPTY). */
int pcc_index_set_input(pcc_index *index, const void *pts, size_t n, size_t stride_bytes,
                        int dim, int mem);
/* One index per device from one upload (every GPU holds the full reference cloud, queries are
 * sharded): the packed cloud of `src` is copied device-to-device (hipMemcpyPeerAsync over xGMI) into a new
 * handle on `device` and indexed there -- rebuilding the grid (~0.1 ms per million points) is cheaper than
 * shipping it.  Same cloud, same indices, same engine; `src` and `device` may be the same device (a second
 * handle with its own stream).  The reference is single-GPU code; this is what its one tree per search site
 * (src/comparator.cpp:564-565) becomes when the query loop is spread over several GPUs of a node. */
int pcc_index_clone_to_device(pcc_index *src, int device, pcc_index **out);
/* The same for `count` devices at once (an entry of devices[] may repeat, or name src's own device): every peer copy
 * is issued before any of them is waited for -- each goes over its own xGMI link --, the builds follow on the clones'
 * own streams, one join at the end.  out[count] receives the handles; on failure none is left behind.  Options, tie
 * order and the requested engine of `src` carry over.  (The reference cloud is "broadcast once" this way in the C++
 * host path; the Python bench uses torch.distributed.broadcast = RCCL for the same step.) */
int pcc_index_clone_to_devices(pcc_index *src, const int *devices, int count, pcc_index **out);
/* number of valid (finite) reference points == PCL total_nr_points_ */
int pcc_index_size(const pcc_index *index, size_t *n_valid);
/* run this index's work on a caller-owned hipStream_t (NULL = library stream) */
int pcc_index_set_stream(pcc_index *index, void *hip_stream);
int pcc_index_sync(pcc_index *index);
/* Device-memory calls (PCC_MEM_DEVICE) are enqueued on the index's stream and return at once.  A caller
 * that produces its inputs or consumes the outputs on ANOTHER stream of the same device orders the two
 * without blocking the host (hipEventRecord + hipStreamWaitEvent):
 *   pcc_index_wait_stream: work submitted to the index after this call starts only when everything
 *     submitted so far to `producer_stream` (a hipStream_t; NULL = the legacy default stream) has finished;
 *   pcc_stream_wait_index: work submitted to `consumer_stream` after this call waits for everything the
 *     index has been asked to do so far.
 * (The reference is single-threaded host code, src/comparator.cpp:571-577: this is what "the call has
 * returned, the vectors are filled" becomes for asynchronous device buffers.) */
int pcc_index_wait_stream(pcc_index *index, void *producer_stream);
int pcc_stream_wait_index(pcc_index *index, void *consumer_stream);
/* engine actually in use (PCC_ENGINE_BRUTE or PCC_ENGINE_GRID) */
int pcc_index_engine(const pcc_index *index, int *engine);
/* force the engine for subsequent searches on this index */
int pcc_index_set_engine(pcc_index *index, int engine);

int pcc_index_set_tie_order(pcc_index *index, int ties);

/* Tuning knobs of one handle.  None of them changes a result bit of a SEARCH (every mode is exact; PCC_OPT_ICP_SORTED is the
 * one knob that shows in a result -- the order a double sum is added up in, see there); they select between
 * implementations that the tests compare with each other and that measurements are taken with.  The PCC_*
 * environment variables of the same names give the DEFAULTS a new handle starts with; the library does not read
 * the environment anywhere else.  Options that shape the index (GRID_PPC, GRID_TRIM, SORT_MP_MIN)
 * take effect at the next pcc_index_set_input.  (The reference has no such surface -- PCL's KdTreeFLANN exposes
 * only setEpsilon / setSortedResults, src/comparator.cpp:564 uses neither.) */
enum pcc_option {
    PCC_OPT_GRID_PPC = 1,        /* mean references per cell the grid aims for (default 0.75) */
    PCC_OPT_GRID_TRIM = 2,       /* k of the trimmed bounding box the grid is laid over (default 3; 0 = plain box) */
    PCC_OPT_FAR_MODE = 3,        /* queries the cell walk leaves: -1 auto, 0 exhaustive kernel, 1 seed scan + ball walk */
    PCC_OPT_ICP_WARM = 4,        /* pcc_icp_align: passes start from the previous pass's neighbours (default 1) */
    PCC_OPT_ICP_DEVICE_LOOP = 5, /* pcc_icp_align: loop resident on the device (default 1; 0 = host-driven, same bits) */
    PCC_OPT_EC_CELLS = 6,        /* clustering over the clique-cell grid: 3 = union-find over the CELLS -- the runs of touching cells along a
                                    row joined by plain stores, one union per pair of neighbouring runs, the rest settled against flat
                                    roots through the caches (default); 4 = the same with one union per occupied cell and face (round 5);
                                    1 = one parent word per point, lanes over a cell's neighbour cells; 2 = the same with lanes over
                                    points; 0 = per-point ball scan on the search grid */
    PCC_OPT_SORT_MP_MIN = 7,     /* reference clouds from this size take the three-level cell sort */
    PCC_OPT_SORT_MP_MIN_Q = 8,   /* the same for query clouds */
    PCC_OPT_NN1_KERNEL = 9,      /* pruned k = 1 kernel: 0 one lane per query; 1 rows drained with lanes over candidates (default);
                                    2 / 3 the same with the open lanes always listed for a second kernel / always finished in place */
    PCC_OPT_FLANN_SPLIT = 10,    /* PCC_TIES_FLANN: split rule replayed, 0 = middleSplit_ (FLANN 1.8.x divideTree), 1 = middleSplit */
    PCC_OPT_NN1_DENSE_MIN = 11,  /* flat k = 1 kernel: a wave whose queries' own cells hold at least this many references on average
                                    takes its first bound from the own cell instead of the own row (default 4) */
    PCC_OPT_KNN_KERNEL = 12,     /* k-NN, k <= 512: 1 = selection by distance buckets, the merge network only for the queries it
                                    hands back (default); 0 = the merge network for every query */
    PCC_OPT_KNN_CACHE_K = 13,    /* 0 (default): off.  K > 0: the self k-NN rows behind pcc_normals / pcc_region_growing are searched
                                    with at least K neighbours and KEPT on the device until the next pcc_index_set_input; a later call
                                    that needs no more than the kept rows hold takes their prefix instead of searching again (the
                                    reference's default segmentation: normals with 50, then region growing with 100 neighbours of the
                                    same cloud -- one search instead of two).  Costs n x K x 8 bytes of device memory */
    PCC_OPT_NN1_OPEN_FLAT = 14,  /* flat k = 1 kernel, listed open lanes: 1 = their rows drained with lanes over candidates (default),
                                    0 = one lane per listed query (round 3) */
    PCC_OPT_SORT_STAGE1 = 15,    /* three-level cell sort, level 1: 1 = reference points leave in bucket-sorted LDS tiles, whole runs
                                    stored (default); 0 = one store per point; 2 = tiles for the queries' 8-byte pairs too */
    PCC_OPT_ICP_SORTED = 16,     /* pcc_icp_align: 1 = the source cloud is brought into the target grid's cell order once and every pass
                                    reads and writes it front to back (default); 0 = caller's order, gathered / scattered in every pass.
                                    (The one option whose setting shows in a result: the 17 double sums of a pass are added up in the
                                    working order, so T and fitness can differ between 0 and 1 in their last bits; every form of the
                                    loop -- device, host, sharded -- agrees to the bit under either.) */
    PCC_OPT_OVERLAP_PREP = 17,   /* pcc_nn1 called directly after pcc_index_set_input / pcc_index_create (the reference's pattern,
                                    src/comparator.cpp:564-577: setInputCloud, then the query loop): 1 = the queries are packed and
                                    sorted on a second stream of the library WHILE the build's cell sort runs (default); 0 = one stream,
                                    one kernel after the other.  Same kernels, same results; only their placement in time differs.
                                    It acts from 2M queries on (below that it hides nothing); 2 = at every size (tests).  Only on the
                                    library's own stream: after pcc_index_set_stream the caller's stream is the one order there is. */
    PCC_OPT_GRID_AXES = 18,      /* which coordinate of the cloud the grid's three axes -- along a row of cells, over the rows of a
                                    layer, over the layers -- follow: -1 = chosen per index from the cloud's extents (default: second
                                    shortest, shortest, longest: the rows and layers next to a query's row are then as close to it in
                                    memory as the cloud allows); -2 = the same whatever GRID_AXES_MIN_POINTS says; 0 = x, y, z (rounds 1-5); 1 xzy, 2 yxz, 3 yzx, 4 zxy, 5 zyx.  Takes
                                    effect at the next pcc_index_set_input.  No result bit depends on it. */
    PCC_OPT_XCD_RUN = 19,        /* k = 1 search: consecutive workgroups (128 cell-sorted queries each) steered to the same XCD,
                                    i.e. the stretch of the grid one L2 works on at a time (default 256) */
    PCC_OPT_FUSE_PARAMS = 20,    /* bits: 1 = index build: the grid (cell edge, dimensions, axes) is derived by the last workgroup of the
                                    pack kernel to finish instead of a kernel of its own behind it; 2 = a pass of pcc_icp_align is solved
                                    by the last workgroup of its sums kernel.  Default 0: see DESIGN.md 4.3 (both were measured) */
    PCC_OPT_HOST_PIPE = 21,      /* PCC_MEM_HOST clouds and results of 8 MB and more in PAGEABLE memory: 1 = staged by the library
                                    through two pinned chunk buffers by a few host threads (PCC_HOST_THREADS, default half the
                                    cores, at most 8), the DMA of a chunk running while the next is gathered; only x, y, z cross
                                    the link when the stride is 24 bytes or more (default); 0 = one hipMemcpyAsync of the raw
                                    array.  Memory the caller has pinned (hipHostMalloc / hipHostRegister) is always copied directly.
                                    Also the SMALL host calls (clouds and results up to 1 MB): 1 = kernels read the cloud from, and write
                                    the results into, the handle's pinned buffers, and a k = 1 query call against up to 4096
                                    exhaustively searched points is one launch; 0 = copies and the separate launches */
    PCC_OPT_SCAN_CHAINED = 22,   /* exclusive scans inside the sorts: 1 = one launch, workgroups hand their totals forward through tagged
                                    64-bit atomics (default); 0 = two launches (totals, then apply) that wait for nothing -- for
                                    environments where workgroups are not dispatched in order (preemption, shared devices) */
    PCC_OPT_KNN_RUN = 23         /* k-NN selection (k <= 512): a wave takes this many consecutive queries of the cell-sorted order in a
                                    row; every query after the first of its run starts from a bound -- its predecessor's K-th distance
                                    plus their separation (the K-th neighbour distance is 1-Lipschitz) -- and skips the cube sizing,
                                    the bucket histogram and the compaction (default 16; 1 = every query on its own, round 5) */
};
int pcc_index_set_option(pcc_index *index, int option, double value);
int pcc_index_get_option(pcc_index *index, int option, double *value);

/* ---- k = 1 nearest neighbour ------------------------------------------------
 * replaces: N calls of KdTreeFLANN::nearestKSearch(pt, 1, idx, d2)
 *           (src/comparator.cpp:571-577; ICP determineCorrespondences and
 *           getFitnessScore reached from :1096,:1099).
 * idx[nq], d2[nq] live in the same memory space as the queries. */
int pcc_nn1(pcc_index *index, const void *queries, size_t nq, size_t stride_bytes,
            int mem, int32_t *idx, float *d2);

/* ---- k nearest neighbours ---------------------------------------------------
 * replaces: nearestKSearch(pt, k, ...) with k = mean_k+1 = 51 inside
 *           StatisticalOutlierRemoval (src/comparator.cpp:1523-1541).
 * Row i holds min(k, n_valid) results ascending by (d2, idx), padded with
 * idx=-1, d2=+inf.  1 <= k <= PCC_KNN_MAX_K. */
int pcc_knn(pcc_index *index, const void *queries, size_t nq, size_t stride_bytes,
            int mem, int k, int32_t *idx, float *d2);

/* ---- radius search ------------------------------------------------------------
 * replaces: KdTreeFLANN::radiusSearch(pt, radius, idx, d2, max_nn = 0) as used
 *           by pcl::extractEuclideanClusters (src/segmentation.cpp:125-131).
 * r2 = float(radius*radius) evaluated in double; the test is strict d2 < r2.
 * Two-pass CSR: count, caller prefix-sums into offsets[nq+1], fill.  With
 * sorted != 0 each row is ascending by (d2, idx) (PCL's sorted results). */
int pcc_radius_count(pcc_index *index, const void *queries, size_t nq,
                     size_t stride_bytes, int mem, double radius, int32_t *counts);
int pcc_radius_fill(pcc_index *index, const void *queries, size_t nq,
                    size_t stride_bytes, int mem, double radius, int sorted,
                    const int64_t *offsets, int32_t *idx, float *d2);
/* the same with radiusSearch's max_nn: 0, or anything from the number of FINITE indexed points on (PCL's
 * total_nr_points_), means "all" and is the pair above; a max_nn below that but beyond PCC_KNN_MAX_K is refused by both calls
 * (PCC_ERR_UNSUPPORTED); otherwise the count is min(count, max_nn) and the row holds the max_nn NEAREST neighbours within the radius,
 * ascending by (d2, idx) whatever `sorted` says (FLANN's KNNRadiusResultSet).  Served by the k-NN kernels with k = max_nn
 * cut at the radius: the reference's own call sites pass 0 (src/segmentation.cpp:125-131), this is for completeness. */
int pcc_radius_count_max(pcc_index *index, const void *queries, size_t nq, size_t stride_bytes, int mem, double radius,
                         unsigned int max_nn, int32_t *counts);
int pcc_radius_fill_max(pcc_index *index, const void *queries, size_t nq, size_t stride_bytes, int mem, double radius,
                        int sorted, unsigned int max_nn, const int64_t *offsets, int32_t *idx, float *d2);

/* ---- Euclidean clustering -----------------------------------------------------
 * replaces: pcl::EuclideanClusterExtraction::extract with
 *           setClusterTolerance/MinClusterSize/MaxClusterSize
 *           (src/segmentation.cpp:125-131): connected components of the graph
 *           d2(i,j) < float(double(float(tol))^2) over the indexed cloud,
 *           filtered to [min_size, max_size], ordered by size descending
 *           (ties: lowest member index first).
 * labels[n_original] (memory space `mem`): cluster id or -1.  sizes[] (host,
 * nullable) receives up to max_sizes cluster sizes.  *n_clusters (host). */
int pcc_euclidean_clusters(pcc_index *index, double tolerance, uint32_t min_size,
                           uint32_t max_size, int mem, int32_t *labels,
                           int32_t *n_clusters, int32_t *sizes, int max_sizes);

/* ---- statistical outlier removal ------------------------------------------------
 * replaces: pcl::StatisticalOutlierRemoval::filter, setMeanK / setStddevMulThresh
 *           (src/comparator.cpp:1523-1527, 1537-1541).  Self-query of the
 *           indexed cloud with k = mean_k+1; mean_dist[i] = float(sum_{j=1..k-1}
 *           sqrt(d2_j) / mean_k) (double sum); threshold = mean + mult*stddev
 *           (double, n-1); inlier[i] = mean_dist[i] <= threshold.
 * mean_dist / inlier have n_original entries in memory space `mem` (either may
 * be NULL); *threshold and *kept are host scalars. */
int pcc_sor(pcc_index *index, int mean_k, double stddev_mult, int mem,
            float *mean_dist, uint8_t *inlier, double *threshold, size_t *kept);
/* PCL adds the mean distances up in index order in double (sum, and sq_sum of float squares).  The library takes both sums,
 * the threshold and the mask on the device whenever no addition of that chain rounds (every order then gives PCL's bits);
 * otherwise it falls back to the in-order loop on the host.  *on_device = 1 when the last pcc_sor stayed on the device. */
int pcc_index_sor_on_device(const pcc_index *index, int *on_device);

/* ---- ICP building blocks ----------------------------------------------------------
 * replaces: pcl::IterativeClosestPoint::align / getFitnessScore
 *           (src/comparator.cpp:1091-1099).
 * pcc_icp_step: NN of every source point against the target index plus the
 *   double-precision sums Umeyama needs, fused in one pass.
 *   sums[0..2]=sum p, [3..5]=sum q, [6..14]=sum q p^T (row-major, q row, p col),
 *   [15]=sum d2, [16]=count.  idx/d2 may be NULL.  sums is a HOST array.
 * pcc_rigid_from_sums: the rigid transform (rotation + translation, no scaling; Horn's quaternion form of
 *   the Umeyama/Kabsch solution, solved in double, returned as a row-major float 4x4) those sums determine.
 *   Pure host arithmetic, needs no handle: a caller that shards the SOURCE cloud over several GPUs adds up the
 *   sums of all shards (one all-reduce of 17 doubles per iteration) and gets the same transform on every rank.
 *   Returns PCC_ERR_INVALID when fewer than 3 correspondences contributed.  The sums are about the ORIGIN: for a small
 *   cloud at large coordinates (geo-referenced scans) sum q p^T - n pm qm^T cancels -- shift both clouds by a common
 *   offset first.  pcc_icp_align does the equivalent itself (it sums about a point of the source cloud).
 * pcc_transform: dst = T * src with PCL's transformPointCloud rounding
 *   ((m0*x + m1*y) + m2*z) + m3; T row-major 4x4 (host); dst may alias src.
 * pcc_icp_align: the whole loop on the device (source stays resident):
 *   max_iter iterations (early exit on |mse-prev| < 1e-12 unless fixed != 0).  In both modes the loop lives on the
 *   device: every pass is search -> sums -> one-workgroup solve (transform, running product, convergence criteria) ->
 *   transform, enqueued in chunks without a host round trip (5 passes per host look with the criteria active, up to 32
 *   with a fixed count); the sums are taken about the first valid source point.  PCC_OPT_ICP_DEVICE_LOOP = 0 selects the
 *   host-driven loop (same bits).
 *   final transform T (host, row-major), *fitness = mean squared NN distance of
 *   the finally transformed source, *converged as PCL's hasConverged(). */
int pcc_rigid_from_sums(const double sums[17], float T[16]);
int pcc_icp_step(pcc_index *target, const void *src, size_t n, size_t stride_bytes,
                 int mem, int32_t *idx, float *d2, double sums[17]);
/* The same two with the sums taken about `center` (sum (p - c), sum (q - c), sum (q - c)(p - c)^T; NULL = origin):
 * what a multi-GPU ICP loop over a geo-referenced cloud should use -- every rank MUST pass the identical center bits
 * (any point of the cloud, e.g. its first, broadcast once), adds up the sums of all ranks and solves with that center. */
int pcc_rigid_from_sums_about(const double sums[17], const double center[3], float T[16]);
int pcc_icp_step_about(pcc_index *target, const void *src, size_t n, size_t stride_bytes, int mem,
                       const double center[3], int32_t *idx, float *d2, double sums[17]);
int pcc_transform(pcc_index *ctx, const float T[16], const void *src, size_t n,
                  size_t src_stride, void *dst, size_t dst_stride, int mem);
int pcc_icp_align(pcc_index *target, const void *src, size_t n, size_t stride_bytes,
                  int mem, int max_iter, int fixed, float T[16], double *fitness,
                  int *iterations, int *converged);

/* ---- descriptor correspondence ------------------------------------------------------
 * replaces: matchRIFTFeaturesKnn (src/comparator.cpp:560-588): index built on
 *   descriptors1, one k=1 query per element of descriptors2, a match is kept
 *   when d2 < threshold (0.05f in the reference).  out[0] = 0 is the dummy
 *   element the reference's vector starts with (:568); *out_size = 1 + matches.
 *   out is a HOST array of at least n2+1 ints. */
int pcc_match_knn(pcc_index *index_des1, const void *des2, size_t n2,
                  size_t stride_bytes, int mem, float threshold, int32_t *out,
                  int32_t *out_size);

/* ---- voxel-grid down-sampling -----------------------------------------------------------------------
 * replaces: pcl::VoxelGrid<PointXYZRGB> with setLeafSize(l, l, l) + filter, the first step of both
 *   segmentation paths (src/segmentation.cpp:69-74 and :224-229, l = 0.025f).  Every occupied voxel
 *   of the world-aligned leaf lattice (index floor(p/leaf) - min_b, as PCL computes it) is replaced
 *   by the centroid of its points; with has_rgb != 0 the packed rgb word at byte offset 16 is
 *   averaged per channel and truncated, as PCL does.  Output order: ascending voxel index.
 *   Non-finite points are ignored.  `ctx` is any index handle (supplies device, stream, scratch).
 * out: caller-allocated, capacity n elements of out_stride bytes (x, y, z at 0 [, rgb at 16]);
 * *out_n (host) = number of voxels written.  Centroid coordinates are accumulated in double and
 * rounded once (PCL: float sums in sort order), i.e. equal to PCL's within float rounding. */
int pcc_voxel_grid(pcc_index *ctx, const void *pts, size_t n, size_t stride_bytes, int mem,
                   float leaf, int has_rgb, void *out, size_t out_stride, size_t *out_n);

/* ---- RANSAC plane segmentation ---------------------------------------------------------------------
 * replaces: pcl::SACSegmentation<PointXYZRGB> with setModelType(SACMODEL_PLANE), setMethodType(SAC_RANSAC),
 *   setOptimizeCoefficients(true), setMaxIterations(100), setDistanceThreshold(0.02), segment(inliers,
 *   coefficients) -- the plane-removal loop in front of the Euclidean clustering
 *   (src/segmentation.cpp:79-99; the ExtractIndices steps :103-116 are index bookkeeping on the host).
 *   PCL's RANSAC is deterministic (fixed-seed mt19937, see sac.hip): the same 3-point samples are drawn,
 *   every candidate plane's support |a x + b y + c z + d| < threshold is counted on the GPU (32 candidates
 *   per pass over the cloud), PCL's best-model / adaptive iteration-count logic runs on the counts, and the
 *   inliers are selected on the GPU in ascending index order.  optimize != 0 adds PCL's least-squares
 *   refit over the inliers and a second selection with the refined plane.
 * inliers[] (memory space `mem`, capacity n); *n_inliers, coefficients[4] (a, b, c, d) and *iterations
 *   (RANSAC iterations PCL would have run; may be NULL) are host values.  n_inliers == 0: no model
 *   (PCL: "Could not estimate a planar model").  `ctx` is any index handle. */
int pcc_sac_plane(pcc_index *ctx, const void *pts, size_t n, size_t stride_bytes, int mem,
                  int max_iterations, double distance_threshold, double probability, int optimize,
                  int32_t *inliers, size_t *n_inliers, float coefficients[4], int *iterations);

/* ---- normals + region growing (default segmentation path) --------------------------------------
 * pcc_normals replaces: pcl::NormalEstimation<PointXYZRGB, pcl::Normal> with setSearchMethod(tree),
 *   setKSearch(k), compute (src/segmentation.cpp:232-241, k = 50; viewpoint left at (0,0,0)).
 *   Per point of the index: its k nearest neighbours (itself first), PCL's single-pass float
 *   mean/covariance in neighbour order, the smallest eigenpair in closed form (pcl::eigen33),
 *   curvature = |lambda_0 / trace|, normal flipped towards the viewpoint.
 * out[n][4] floats (memory space `mem`): nx, ny, nz, curvature.  NaN for non-finite points and when
 *   fewer than 3 neighbours exist.  viewpoint may be NULL (= origin).
 * pcc_region_growing replaces: pcl::RegionGrowing<PointXYZRGB, pcl::Normal> with setSearchMethod,
 *   setNumberOfNeighbours(k), setSmoothnessThreshold, setCurvatureThreshold, setMin/MaxClusterSize,
 *   setInputCloud / setInputNormals, extract (src/segmentation.cpp:259-271: 50, 1000000, k = 100,
 *   3/180*pi, 1.0).  The k-neighbour rows of every point are searched on the GPU in one batch
 *   (PCL: findPointNeighbours, one nearestKSearch per point).  The result is PCL's: seeds by
 *   ascending curvature, breadth first through the rows while |n_current . n_neighbour| >=
 *   cos(smoothness), a neighbour continues the walk when its curvature is <= curvature_threshold.
 *   The regions are computed on the GPU in an equivalent order-free form: the label of a point is the
 *   lowest-ranked (curvature, index) point that reaches it along valid edges, where a point above the
 *   curvature threshold passes labels on only if it is a seed itself (region.hip).
 * normals[n][4] as pcc_normals writes them (memory space `mem`); labels[n] (memory space `mem`):
 *   index of the kept cluster, in PCL's output order (creation order), or -1;
 *   *n_clusters (host) = clusters.size().  Equal curvatures are taken in index order (PCL:
 *   std::sort, unspecified). */
int pcc_normals(pcc_index *index, int k, const float viewpoint[3], int mem, float *out);
/* the same with setRadiusSearch(radius) instead of setKSearch (the normals of the RIFT pipeline,
 * src/comparator.cpp:628-635, radius 0.03): the neighbourhood is the sorted radiusSearch result
 * (d2 < float(radius^2), ascending (d2, index)); NaN where it holds fewer than 3 points. */
int pcc_normals_radius(pcc_index *index, double radius, const float viewpoint[3], int mem, float *out);
int pcc_region_growing(pcc_index *index, const float *normals, int mem, int k, float smoothness,
                       float curvature_threshold, uint32_t min_size, uint32_t max_size,
                       int32_t *labels, int32_t *n_clusters);

/* ---- first point within a radius ----------------------------------------------------------------
 * replaces: the O(S*N) linear scan in processRIFTwithSIFT (src/comparator.cpp:696-713) that snaps
 *   every SIFT keypoint to the FIRST cloud point j (lowest index) with
 *   sqrt(pow(kx - px, 2) + pow(ky - py, 2) + pow(kz - pz, 2)) < radius  -- differences in float,
 *   squares / sum / sqrt in double, strict compare against the double radius (0.05 there).
 * idx[nq] (memory space `mem`): that lowest index, or -1 when no point qualifies. */
int pcc_first_within(pcc_index *index, const void *queries, size_t nq, size_t stride_bytes,
                     int mem, double radius, int32_t *idx);

/* ---- multi-GPU: RCCL over xGMI ---------------------------------------------
 * The path shards by independent queries: every GPU holds the whole reference cloud and its own index (one pcc_index per
 * device), queries are split into contiguous shards and searched with the calls above -- no exchange inside a search.
 * Exchanged are: the reference cloud, once (broadcast); the 17 sums of every ICP pass when the SOURCE cloud is sharded
 * (all-reduce on the handle's stream, inside the device-resident loop); SOR's statistics when the cloud's points are
 * sharded.  Clustering does not shard (global union-find): replicas only.
 * replaces: the single pcl::KdTreeFLANN / pcl::IterativeClosestPoint / pcl::StatisticalOutlierRemoval object per call site
 *   (src/comparator.cpp:564-577, 1089-1110, 1523-1541) when a node's GPUs share one cloud pair.
 * A pcc_comm is one rank of an RCCL communicator: one per (process, GPU).  librccl.so.1 is loaded when the first one is
 * made (dlopen); libpcc_nn.so does not depend on it otherwise.
 * The calls marked "collective" must be made by EVERY rank of the communicator.  A failure on ONE rank -- a shard outside
 * the cloud, an empty shard, an allocation that is refused, a cloud the root cannot index -- does not leave the others
 * waiting: before every data collective the ranks agree on one status word (all-reduce MIN), and every rank returns the
 * failing rank's code (pcc_last_error on the others: "another rank of the communicator failed").  Only a rank that never
 * makes the call (or passes no communicator) can still stall its peers, as in any RCCL program. */
typedef struct pcc_comm pcc_comm;
#define PCC_COMM_ID_BYTES 128
/* one process per GPU: rank 0 makes the id (PCC_COMM_ID_BYTES bytes), hands it to the other ranks over whatever launched
 * them, every rank then calls pcc_comm_create_rank (collective: it returns when all `world` ranks have called) */
int pcc_comm_unique_id(void *id, size_t bytes);
int pcc_comm_create_rank(const void *id, size_t bytes, int world, int rank, int device, pcc_comm **out);
/* one process driving several GPUs: out[k] is rank k on devices[k] (distinct devices).  Collective calls on these handles
 * must be issued from one thread per device (every rank blocks until the others have joined) */
int pcc_comm_create_local(const int *devices, int count, pcc_comm **out);
int pcc_comm_destroy(pcc_comm *comm);
int pcc_comm_info(const pcc_comm *comm, int *rank, int *world, int *device);
/* the reference cloud of rank `root` indexed on EVERY rank: root indexes (points, n, stride, mem) as pcc_index_create does,
 * the packed cloud (16 B per point) is broadcast once over RCCL, every other rank builds its own index over the copy
 * (0.5 ms at 10M points: cheaper than shipping the index).  Collective; points / n are read on the root only;
 * *n_out (nullable): the number of points of the cloud, on every rank. */
int pcc_index_create_broadcast(pcc_comm *comm, int root, const void *points, size_t n, size_t stride_bytes, int mem,
                               int engine, pcc_index **out, size_t *n_out);
/* pcc_icp_align with the SOURCE cloud sharded over the ranks (each rank: its shard, its handle over the same target).  Per
 * pass the 17 double sums are all-reduced (136 bytes) before the solver runs, so every rank applies the same transform;
 * fitness is the mean squared distance over ALL shards.  With one rank: pcc_icp_align's result, bit for bit.  Collective. */
int pcc_icp_align_sharded(pcc_index *index, pcc_comm *comm, const void *source_shard, size_t n, size_t stride_bytes, int mem,
                          int max_iterations, int fixed_iterations, float T[16], double *fitness, int *iterations,
                          int *converged);
/* SOR over a shard [start, start + count) of the indexed cloud: the shard's mean distances (memory space `mem`, nullable)
 * and its share of PCL's statistics -- sums[0] = sum of the means, [1] = sum of their float squares, [2] / [3] = bit
 * patterns (as doubles) of the smallest positive term of either sum.  Combine over shards with (+, +, min, min), then
 * pcc_sor_threshold (host arithmetic, no handle): *exact = 0 means PCL's in-order additions would round, i.e. the combined
 * sums need not carry PCL's last bits (pcc_sor_sharded and pcc_sor then take the sums in index order). */
int pcc_sor_partial(pcc_index *index, size_t start, size_t count, int mean_k, int mem, float *mean_dist, double sums[4]);
int pcc_sor_threshold(const double sums[4], uint64_t n_valid, int mean_k, double stddev_mult, double *threshold, int *exact);
/* pcc_sor over the ranks of `comm`: this rank filters the points [start, start + count) (the ranks' shards tile the cloud);
 * the statistics are PCL's over the WHOLE cloud (one all-reduce of the sums, one of the smallest terms, one of the kept
 * count).  mean_dist / inlier: `count` entries in memory space `mem`; *threshold, *kept_total: same on every rank. */
int pcc_sor_sharded(pcc_index *index, pcc_comm *comm, size_t start, size_t count, int mean_k, double stddev_mult, int mem,
                    float *mean_dist, uint8_t *inlier, double *threshold, size_t *kept_total);

/* ---- instrumentation ------------------------------------------------------------------
 * counters of the last search on this index (host):
 *  stats[0] queries resolved by the GRID engine, [1] queries sent to the BRUTE
 *  fallback, [2] reference points valid, [3] grid cells, [4] distances evaluated by the pruned kernels (k = 1, k-NN,
 *  radius, clustering, tie flags) since the previous pcc_index_stats call, process-wide -- counted by the PROFILING build
 *  only (libpcc_nn_prof.so, `make prof`; pcc_counts_pairs() == 1), 0 in libpcc_nn.so, [5] queries flagged as tied and [6] indices changed by
 *  the FLANN walk (PCC_TIES_FLANN, last search), [7] queries the 3x3x3 cube of the pruned k = 1 kernel left open
 *  (last search that listed them: from 2M queries on, or PCC_OPT_NN1_KERNEL = 2). */
int pcc_index_stats(const pcc_index *index, uint64_t stats[8]);
/* 1 when this library was built with the pair counter (-DPCC_COUNT_PAIRS: the profiling build), else 0 */
int pcc_counts_pairs(void);
/* test hook: the nth device allocation the library makes from now on (process-wide, nth >= 1) fails as an exhausted
 * hipMalloc does -- PCC_ERR_NOMEM from whichever call needed it; 0 disarms.  The reference maps every failure to a return
 * code (src/comparator.cpp:1123,1134,1179); this is how the tests reach the library's allocation-failure paths, the
 * collective ones above all (a rank that cannot allocate must take its peers out of the call, not leave them waiting). */
int pcc_debug_fail_alloc(int nth);
/* HIP-event timing of the library's own kernels, recorded on the index's stream
 * (events of another stream would not see them).  After enabling, every
 * set_input / search records events into a 64-call ring without synchronising;
 * pcc_index_timing synchronises the stream and returns the AVERAGE milliseconds
 * over the calls recorded since enabling (ms[7] = number of main-kernel samples):
 *  ms[0] main search kernel (GRID ring search, or the exhaustive kernel under
 *        ENGINE_BRUTE), ms[1] exhaustive fallback pass of the GRID engine,
 *  ms[2] whole last search call (first to last kernel), ms[3] whole last
 *  set_input/build, ms[4] query sort (GRID), ms[5..6] reserved. */
/* on: 0 off; 1 only the main search kernel (two events per call -- what a timed region can
 * afford: every recorded event costs a few microseconds of stream time); 2 full breakdown */
int pcc_index_enable_timing(pcc_index *index, int on);
int pcc_index_timing(pcc_index *index, float ms[8]);

#ifdef __cplusplus
}
#endif
#endif
