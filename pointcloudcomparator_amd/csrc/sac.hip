// sac.hip -- RANSAC plane segmentation: pcl::SACSegmentation with SACMODEL_PLANE + SAC_RANSAC, the plane
// removal loop in front of the Euclidean clustering (reference src/segmentation.cpp:79-117:
// optimize on, 100 iterations, distance threshold 0.02).
//
// PCL's RANSAC draws one 3-point sample per iteration from a fixed-seed generator, counts the points
// within the threshold of its plane (one pass over the cloud per iteration -- the cost), keeps the best
// and shortens the iteration bound k = log(1 - p) / log(1 - w^3) as the best inlier ratio w grows.
// The sample sequence does not depend on the counts, only the stopping point does.  So:
//   host   : replays PCL's sampling exactly (mt19937(12345), eng()/2, partial Fisher-Yates over the
//            persistent shuffled index array, degenerate-sample retries) and forms the candidate planes
//   device : k_sac_count evaluates a batch of 32 candidate planes in ONE pass over the cloud
//            (per-lane counters, wave reduction, one atomic per wave and plane)
//   host   : walks the counts with PCL's best/k logic; further batches only if k has not been reached
//   device : inlier flags -> exclusive scan -> ascending index list (PCL's selectWithinDistance order)
//   host   : optimizeModelCoefficients -- PCL's single-pass float moment sums over the inliers IN ORDER
//            (a serial float chain, kept serial so the refit equals PCL's), pcl::eigen33, and a second
//            device selection with the refined plane.
// 4-float Eigen reductions are evaluated as (p0 + p1) + (p2 + p3) (packet product + hadd predux).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "pcc_internal.hpp"
#include "plane_fit.hpp"

namespace pcc {

namespace {

constexpr int SAC_BATCH = 32;

__device__ __forceinline__ bool plane_inlier_dev(const float4 c, const float4 p, double threshold) {
    const float d = (c.x * p.x + c.y * p.y) + (c.z * p.z + c.w * 1.0f);
    return (__float_as_int(p.w) >= 0) && (fabs((double)d) < threshold);
}

__global__ void __launch_bounds__(256)
k_sac_count(const float4* __restrict__ pts, unsigned int n, const float4* __restrict__ models, double threshold,
            unsigned int* __restrict__ counts) {
    __shared__ float4 sm[SAC_BATCH];
    if (threadIdx.x < SAC_BATCH) sm[threadIdx.x] = models[threadIdx.x];
    __syncthreads();
    unsigned int cnt[SAC_BATCH];
#pragma unroll
    for (int m = 0; m < SAC_BATCH; ++m) cnt[m] = 0;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
#pragma unroll
        for (int m = 0; m < SAC_BATCH; ++m) cnt[m] += plane_inlier_dev(sm[m], p, threshold) ? 1u : 0u;
    }
    // wave sums -> LDS -> ONE row of partial counts per workgroup (no same-address atomics: 16k waves adding
    // to 32 words serialised to ~1 ms); the host adds the rows
    __shared__ unsigned int red[4][SAC_BATCH];
#pragma unroll
    for (int m = 0; m < SAC_BATCH; ++m) {
        unsigned int v = cnt[m];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][m] = v;
    }
    __syncthreads();
    if (threadIdx.x < SAC_BATCH)
        counts[blockIdx.x * SAC_BATCH + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void __launch_bounds__(256)
k_sac_flag(const float4* __restrict__ pts, unsigned int n, float4 model, double threshold, unsigned int* __restrict__ flag) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x)
        flag[i] = (i < n && plane_inlier_dev(model, pts[i], threshold)) ? 1u : 0u;  // flag[n] = 0: scan[n] = total
}

__global__ void __launch_bounds__(256)
k_sac_scatter(const float4* __restrict__ pts, unsigned int n, float4 model, double threshold,
              const unsigned int* __restrict__ pos, int32_t* __restrict__ out) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (plane_inlier_dev(model, pts[i], threshold)) out[pos[i]] = (int32_t)i;
}

// out[j] = the three coordinates of point idx[j] as PCL sees them (non-finite points, zeroed and flagged by the staging,
// come back as NaN): the handful of sampled points of a batch, and the inliers of the refit in list order
__global__ void __launch_bounds__(256)
k_sac_gather(const float4* __restrict__ pts, const int32_t* __restrict__ idx, unsigned int m, float* __restrict__ out) {
    for (unsigned int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
        const float4 p = pts[idx[j]];
        const bool bad = __float_as_int(p.w) < 0;
        const float qnan = __int_as_float(0x7fc00000);
        out[3 * (size_t)j + 0] = bad ? qnan : p.x;
        out[3 * (size_t)j + 1] = bad ? qnan : p.y;
        out[3 * (size_t)j + 2] = bad ? qnan : p.z;
    }
}

// ---- host side: PCL's generator, sampling and plane arithmetic ------------------------------------------
struct Mt19937 {
    uint32_t mt[624];
    int idx;
    explicit Mt19937(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int i = 0; i < 624; ++i) {
                const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    // boost::uniform_int<>(0, INT_MAX) over a 32-bit engine: bucket size 2, no rejection
    int rnd() { return (int)(next() / 2u); }
};

inline float dot4(const float a[4], const float b[4]) { return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]); }

inline bool sample_degenerate(const float* p0, const float* p1, const float* p2) {
    const float d0 = (p1[0] - p0[0]) / (p2[0] - p0[0]);
    const float d1 = (p1[1] - p0[1]) / (p2[1] - p0[1]);
    const float d2 = (p1[2] - p0[2]) / (p2[2] - p0[2]);
    return (d0 == d1) && (d2 == d1);
}

inline bool plane_from_sample(const float* p0, const float* p1, const float* p2, float c[4]) {
    const float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    const float b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    if (sample_degenerate(p0, p1, p2)) return false;
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
    c[3] = 0.f;
    const float inv = 1.0f / std::sqrt(dot4(c, c));  // normalize(): Eigen 3.2 multiplies by the reciprocal
    c[0] *= inv; c[1] *= inv; c[2] *= inv; c[3] *= inv;
    const float p[4] = {p0[0], p0[1], p0[2], 1.0f};
    c[3] = -1.0f * dot4(c, p);
    return true;
}

inline int g1(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// ascending list of the points within the threshold of `model`; *m = their number
int select_inliers(pcc_index* ix, const float4* pts, unsigned int n, const float c[4], double threshold,
                   int32_t* out_dev, size_t* m) {
    hipStream_t s = ix->stream;
    PCC_TRY(ix->scratch_c.reserve(((size_t)n + 1) * 4));
    unsigned int* pos = ix->scratch_c.as<unsigned int>();
    const float4 model = make_float4(c[0], c[1], c[2], c[3]);
    hipLaunchKernelGGL(k_sac_flag, dim3(g1(n + 1)), dim3(256), 0, s, pts, n, model, threshold, pos);
    PCC_HIP(hipGetLastError());
    PCC_TRY(launch_exclusive_scan(ix, s, pos, (size_t)n + 1, ix->scratch_d));
    hipLaunchKernelGGL(k_sac_scatter, dim3(g1(n)), dim3(256), 0, s, pts, n, model, threshold, pos, out_dev);
    PCC_HIP(hipGetLastError());
    unsigned int* h = static_cast<unsigned int*>(ix->pinned);
    PCC_HIP(hipMemcpyAsync(h, pos + n, 4, hipMemcpyDeviceToHost, s));
    PCC_HIP(hipStreamSynchronize(s));
    *m = h[0];
    return PCC_OK;
}

}  // namespace

// PCL's persistent shuffled index array (`shuffled_indices_`, the identity at first), touched in three places per sample:
// kept as the identity plus the entries that differ -- initialising n integers cost more than everything else the host
// does for a million points
struct SparseShuffle {
    std::vector<std::pair<uint32_t, int32_t>> slots;  // open addressing, key + 1 (0 = empty)
    size_t used = 0;
    SparseShuffle() : slots(1024, {0u, 0}) {}
    int32_t get(uint32_t i) const {
        for (size_t h = (i * 2654435761u) & (slots.size() - 1);; h = (h + 1) & (slots.size() - 1)) {
            if (slots[h].first == 0u) return (int32_t)i;
            if (slots[h].first == i + 1u) return slots[h].second;
        }
    }
    void set(uint32_t i, int32_t v) {
        if (2 * (used + 1) > slots.size()) {
            std::vector<std::pair<uint32_t, int32_t>> old(slots.size() * 2, {0u, 0});
            old.swap(slots);
            used = 0;
            for (const auto& e : old)
                if (e.first) set(e.first - 1u, e.second);
        }
        for (size_t h = (i * 2654435761u) & (slots.size() - 1);; h = (h + 1) & (slots.size() - 1)) {
            if (slots[h].first == 0u) { slots[h] = {i + 1u, v}; ++used; return; }
            if (slots[h].first == i + 1u) { slots[h].second = v; return; }
        }
    }
    void swap(uint32_t a, uint32_t b) {
        if (a == b) return;
        const int32_t va = get(a), vb = get(b);
        set(a, vb);
        set(b, va);
    }
};

// pts_dev: the staged cloud (float4, w < 0 = non-finite).  host_base / host_stride: the caller's own array when it lives on
// the host (single points are read from it).  host_base == nullptr (the cloud is in device memory): the points the host
// needs -- the 96 sampled points of a batch, the inliers of the refit -- are gathered on the device and copied compactly;
// nothing of the size of the cloud crosses to the host.  The batch's samples are then drawn before their coordinates are
// known, which is PCL's sequence as long as no sample is degenerate (PCL redraws those at once, exact float equalities:
// duplicates, collinear lattice points); the first degenerate sample ends the attempt with PCC_ERR_RETRY_HOST and the caller
// repeats the call with a host copy of the cloud, as round 3 always did.
int sac_plane(pcc_index* ix, const float4* pts_dev, size_t n_, const char* host_base, size_t host_stride,
              int max_iterations, double threshold, double probability, int optimize, int32_t* inliers_dev,
              size_t* n_inliers, float coeff[4], int* iterations_out) {
    hipStream_t s = ix->stream;
    const unsigned int n = (unsigned int)n_;
    const bool gathered = host_base == nullptr;
    auto P = [&](int32_t i) { return reinterpret_cast<const float*>(host_base + (size_t)i * host_stride); };
    *n_inliers = 0;
    coeff[0] = coeff[1] = coeff[2] = coeff[3] = 0.f;
    if (iterations_out) *iterations_out = 0;
    if (n < 3) return PCC_OK;  // PCL: "Can not select 3 unique points" -> no model, no inliers

    PCC_TRY(ix->scratch_a.reserve(SAC_BATCH * sizeof(float4)));
    const int count_blocks = std::min(g1(n), 512);  // 2 workgroups per CU, grid-stride
    PCC_TRY(ix->scratch_b.reserve((size_t)count_blocks * SAC_BATCH * sizeof(unsigned int)));
    std::vector<unsigned int> part((size_t)count_blocks * SAC_BATCH);
    float4* d_models = ix->scratch_a.as<float4>();
    unsigned int* d_counts = ix->scratch_b.as<unsigned int>();
    // the sampled points of a batch: indices up, coordinates back (pinned staging in the handle's small block)
    int32_t* d_smp_idx = nullptr;
    float* d_smp_xyz = nullptr;
    if (gathered) {
        PCC_TRY(ix->scratch_e.reserve(SAC_BATCH * 3 * (sizeof(int32_t) + 3 * sizeof(float)) + 64));
        d_smp_idx = ix->scratch_e.as<int32_t>();
        d_smp_xyz = reinterpret_cast<float*>(d_smp_idx + SAC_BATCH * 3);
    }

    Mt19937 gen(12345u);
    SparseShuffle shuffled;
    int iterations = 0, best_count = -INT32_MAX;
    bool have_model = false, stop = false;
    double k = 1.0;
    const double log_probability = std::log(1.0 - probability);
    const double one_over_indices = 1.0 / (double)n;
    unsigned skipped = 0;
    const unsigned max_skip = (unsigned)max_iterations * 10u;
    float best[4] = {0, 0, 0, 0};
    const float qnan = std::nanf("");
    ev_mark(ix, EV_MAIN0);
    while (!stop && iterations < k && skipped < max_skip) {
        // the next batch of candidate planes, drawn exactly as PCL's loop would draw them
        float models[SAC_BATCH][4];
        int nm = 0;
        bool sampling_failed = false;
        unsigned skipped_batch = skipped;
        auto draw = [&](int32_t smp[3]) {
            for (unsigned i = 0; i < 3; ++i) shuffled.swap(i, i + (uint32_t)((size_t)gen.rnd() % (n - i)));
            smp[0] = shuffled.get(0); smp[1] = shuffled.get(1); smp[2] = shuffled.get(2);
        };
        if (gathered) {
            // one draw per candidate, all candidates of the batch at once; their coordinates in one small round trip
            int32_t smp[SAC_BATCH][3];
            float xyz[SAC_BATCH][3][3];
            int want = 0;
            while (want < SAC_BATCH && iterations + want <= max_iterations) { draw(smp[want]); ++want; }
            if (want > 0) {
                PCC_HIP(hipMemcpyAsync(d_smp_idx, smp, (size_t)want * 3 * sizeof(int32_t), hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_sac_gather, dim3(1), dim3(256), 0, s, pts_dev, d_smp_idx, (unsigned int)want * 3u, d_smp_xyz);
                PCC_HIP(hipGetLastError());
                PCC_HIP(hipMemcpyAsync(xyz, d_smp_xyz, (size_t)want * 9 * sizeof(float), hipMemcpyDeviceToHost, s));
                PCC_HIP(hipStreamSynchronize(s));
            }
            for (int c = 0; c < want; ++c) {
                if (!plane_from_sample(xyz[c][0], xyz[c][1], xyz[c][2], models[nm])) return PCC_ERR_RETRY_HOST;  // PCL would redraw here
                ++nm;
            }
        } else
        while (nm < SAC_BATCH && iterations + nm <= max_iterations && skipped_batch < max_skip) {
            int32_t smp[3];
            bool got = false;
            for (int it = 0; it < 1000 && !got; ++it) {  // max_sample_checks_
                draw(smp);
                got = !sample_degenerate(P(smp[0]), P(smp[1]), P(smp[2]));
            }
            if (!got) { sampling_failed = true; break; }
            if (!plane_from_sample(P(smp[0]), P(smp[1]), P(smp[2]), models[nm])) { ++skipped_batch; continue; }
            ++nm;
        }
        // NOTE: the draws above run ahead of PCL's loop by at most one batch; the generator state is only
        // consumed by draws, so running ahead never changes a plane PCL would have used.
        for (int m = nm; m < SAC_BATCH; ++m) models[m][0] = models[m][1] = models[m][2] = models[m][3] = qnan;
        unsigned int counts[SAC_BATCH] = {0};
        if (nm > 0) {
            PCC_HIP(hipMemcpyAsync(d_models, models, sizeof(models), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_sac_count, dim3(count_blocks), dim3(256), 0, s, pts_dev, n, d_models, threshold, d_counts);
            PCC_HIP(hipGetLastError());
            PCC_HIP(hipMemcpyAsync(part.data(), d_counts, part.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, s));
            PCC_HIP(hipStreamSynchronize(s));
            for (int b = 0; b < count_blocks; ++b)
                for (int m = 0; m < SAC_BATCH; ++m) counts[m] += part[(size_t)b * SAC_BATCH + m];
        }
        // PCL's loop over this batch.  Skips (degenerate samples) happened between the valid models in
        // generation order; they only matter through the max_skip bound, which is checked per batch.
        skipped = skipped_batch;
        for (int m = 0; m < nm; ++m) {
            if (!(iterations < k)) { stop = true; break; }
            const int cnt = (int)counts[m];
            if (cnt > best_count) {
                best_count = cnt;
                have_model = true;
                std::memcpy(best, models[m], sizeof(best));
                const double w = (double)best_count * one_over_indices;
                double p_no_outliers = 1.0 - std::pow(w, 3.0);
                p_no_outliers = std::fmax(DBL_EPSILON, p_no_outliers);
                p_no_outliers = std::fmin(1.0 - DBL_EPSILON, p_no_outliers);
                k = log_probability / std::log(p_no_outliers);
            }
            ++iterations;
            if (iterations > max_iterations) { stop = true; break; }
        }
        if (sampling_failed || nm == 0) break;
    }
    if (iterations_out) *iterations_out = iterations;
    if (!have_model) { ev_mark(ix, EV_MAIN1); return PCC_OK; }

    size_t m = 0;
    PCC_TRY(select_inliers(ix, pts_dev, n, best, threshold, inliers_dev, &m));
    std::memcpy(coeff, best, sizeof(best));
    if (optimize && m >= 4) {
        // optimizeModelCoefficients: computeMeanAndCovarianceMatrix over the inliers, in index order, in float -- a serial
        // float chain, PCL's bits only in PCL's order: it stays a host loop.  What it reads: the caller's array, or the
        // inliers' coordinates gathered on the device in list order (12 bytes per INLIER, not 16 per point of the cloud)
        float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto add = [&](const float* q) {
            a[0] += q[0] * q[0]; a[1] += q[0] * q[1]; a[2] += q[0] * q[2];
            a[3] += q[1] * q[1]; a[4] += q[1] * q[2]; a[5] += q[2] * q[2];
            a[6] += q[0]; a[7] += q[1]; a[8] += q[2];
        };
        if (gathered) {
            PCC_TRY(ix->scratch_f.reserve(m * 3 * sizeof(float)));
            PCC_TRY(ix->host_a.reserve(m * 3 * sizeof(float)));
            hipLaunchKernelGGL(k_sac_gather, dim3(g1(m)), dim3(256), 0, s, pts_dev, inliers_dev, (unsigned int)m, ix->scratch_f.as<float>());
            PCC_HIP(hipGetLastError());
            PCC_HIP(hipMemcpyAsync(ix->host_a.p, ix->scratch_f.p, m * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
            PCC_HIP(hipStreamSynchronize(s));
            const float* hq = ix->host_a.as<float>();
            for (size_t j = 0; j < m; ++j) add(hq + 3 * j);
        } else {
            std::vector<int32_t> hi(m);
            PCC_HIP(hipMemcpyAsync(hi.data(), inliers_dev, m * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            PCC_HIP(hipStreamSynchronize(s));
            for (size_t j = 0; j < m; ++j) add(P(hi[j]));
        }
        float cov[9], nrm[3], curv;
        covariance_from_sums(a, (unsigned int)m, cov);
        plane_from_covariance(cov, nrm, &curv);
        float o[4] = {nrm[0], nrm[1], nrm[2], 0.f};
        const float cen[4] = {a[6], a[7], a[8], 0.f};
        o[3] = -1.0f * dot4(o, cen);
        std::memcpy(coeff, o, sizeof(o));
        PCC_TRY(select_inliers(ix, pts_dev, n, o, threshold, inliers_dev, &m));
    }
    ev_mark(ix, EV_MAIN1);
    *n_inliers = m;
    return PCC_OK;
}

}  // namespace pcc
