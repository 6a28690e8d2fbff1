// icp.hip -- the reduction ICP needs after every correspondence pass (gfx950).
//
// Replaces the host loop of pcl::registration::TransformationEstimationSVD::
// estimateRigidTransformation (reached from the reference's icp.align(),
// src/comparator.cpp:1096): PCL copies the matched pairs into two 3xN float matrices and
// calls Eigen's umeyama(); all umeyama needs from them are sum p, sum q and sum q p^T.
// The sums are accumulated in double (PCL/Eigen use float; compared with a tolerance,
// SURVEY hard part 6) as one partial row per workgroup, reduced on the host in a fixed
// order -> bitwise reproducible run to run (no float atomics).
#include "pcc_internal.hpp"
#include "rigid_solve.hpp"

namespace pcc {

// The device-resident loop's solver and judge: one workgroup.  Lanes 0..16 add the per-workgroup partial sums in
// workgroup order (the order the host loop uses, so the sums have the same bits), lane 0 then runs Horn's closed form
// (rigid_solve.hpp), composes the running transform and applies the loop's criteria exactly as the host loop does
// (SURVEY 9.5): fewer than 3 correspondences -> stop, not converged; iteration cap -> stop, converged; |mse - previous|
// < 1e-12 (unless `fixed`) -> stop, converged.  Once stopped the state is frozen and every later pass that was already
// enqueued applies the identity.
// One body, two homes: k_icp_solve (a launch of its own: the sharded loop, whose sums pass through an all-reduce first) and the
// LAST workgroup of k_icp_sums to finish (round 6: a pass of the one-GPU loop is search -> sums, one launch less).  AGENT: the
// rows were written by other workgroups of the same launch and are read past the caches.
template <bool AGENT>
__device__ __forceinline__ void icp_solve_block(const double* __restrict__ partials, int n_blocks, IcpState* __restrict__ st, int max_iter,
                                                int fixed, const double* __restrict__ center_dev, unsigned int* __restrict__ zero_word,
                                                double* __restrict__ part /* LDS, n_blocks * 17 doubles */) {
    // (the counters of the search that follows -- fallback list, far list, the sharded open-lane counters -- as k_pack / k_transform
    // zero them: when the next pass's search applies the transform itself there is no k_transform in between)
    if (zero_word && threadIdx.x < 64) {
        if (threadIdx.x < 2) zero_word[threadIdx.x] = 0u;
        zero_word[PCC_OPEN_CTR0 - 32 + threadIdx.x * PCC_OPEN_CTR_STRIDE] = 0u;
    }
    __shared__ double sums[17];
    // all partial rows, staged with coalesced loads, 8 per thread in flight (the rows come from other XCDs' write-backs: read
    // one by one in a dependent loop they cost 120 us)
    const int total = n_blocks * 17;
    for (int base = 0; base < total; base += 8 * (int)blockDim.x) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + (int)threadIdx.x + u * (int)blockDim.x;
            v[u] = 0.0;
            if (i < total) {
                if (AGENT) v[u] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(partials) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                else v[u] = partials[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + (int)threadIdx.x + u * (int)blockDim.x;
            if (i < total) part[i] = v[u];
        }
    }
    __syncthreads();
    if (threadIdx.x < 17) {
        double a = 0;
        int b = 0;
        for (; b + 16 <= n_blocks; b += 16) {  // 16 LDS reads in flight, added in workgroup order
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = part[(b + u) * 17 + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += v[u];
        }
        for (; b < n_blocks; ++b) a += part[b * 17 + threadIdx.x];
        sums[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float Ti[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (!st->stopped) {
        double sm[17];
        for (int k = 0; k < 17; ++k) sm[k] = sums[k];
        float Tn[16];
        const double center[3] = {center_dev[0], center_dev[1], center_dev[2]};
        if (rigid_from_sums(sm, Tn, center) != 0) {
            st->stopped = 1;  // min_number_correspondences_: not converged
            st->converged = 0;
        } else {
            for (int k = 0; k < 16; ++k) Ti[k] = Tn[k];
            float Tt[16];
            for (int k = 0; k < 16; ++k) Tt[k] = st->T[k];
            mat4_mul_f(Ti, Tt, Tt);  // final = T_i * final
            for (int k = 0; k < 16; ++k) st->T[k] = Tt[k];
            const double mse = sm[15] / sm[16];
            const int it = st->it + 1;
            st->it = it;
            if (it >= max_iter || (!fixed && fabs(mse - st->prev_mse) < 1e-12)) {
                st->stopped = 1;  // DefaultConvergenceCriteria: the iteration cap counts as converged
                st->converged = 1;
            }
            st->prev_mse = mse;
        }
    }
    for (int k = 0; k < 16; ++k) st->Ti[k] = Ti[k];
}

__global__ void __launch_bounds__(1024)
k_icp_solve(const double* __restrict__ partials, int n_blocks, IcpState* __restrict__ st, int max_iter, int fixed,
            const double* __restrict__ center_dev, unsigned int* __restrict__ zero_word) {
    extern __shared__ double part[];
    icp_solve_block<false>(partials, n_blocks, st, max_iter, fixed, center_dev, zero_word, part);
}

__global__ void __launch_bounds__(256)
k_icp_sums(const float4* __restrict__ src, unsigned int n, const unsigned long long* __restrict__ keys,
           const float4* __restrict__ refs, double* __restrict__ partials, const unsigned int* __restrict__ mirror_dev,
           unsigned int* __restrict__ mirror_host, const double* __restrict__ center, IcpFuse fuse) {
    const double cx = center ? center[0] : 0.0, cy = center ? center[1] : 0.0, cz = center ? center[2] : 0.0;
    if (mirror_dev && blockIdx.x == 0 && threadIdx.x == 0) *mirror_host = *mirror_dev;  // fallback count for the far-query heuristic
    double acc[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) acc[k] = 0.0;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long key = keys[i];
        const float4 p = src[i];
        if (key_none(key) || __float_as_int(p.w) < 0) continue;  // no correspondence
        const float4 t = refs[(unsigned int)(key & 0xffffffffull)];
        // (about the caller's centre: see rigid_from_sums; exact in double, a float difference would not be)
        const double px = (double)p.x - cx, py = (double)p.y - cy, pz = (double)p.z - cz;
        const double qx = (double)t.x - cx, qy = (double)t.y - cy, qz = (double)t.z - cz;
        acc[0] += px; acc[1] += py; acc[2] += pz;
        acc[3] += qx; acc[4] += qy; acc[5] += qz;
        acc[6] += qx * px; acc[7] += qx * py; acc[8] += qx * pz;
        acc[9] += qy * px; acc[10] += qy * py; acc[11] += qy * pz;
        acc[12] += qz * px; acc[13] += qz * py; acc[14] += qz * pz;
        acc[15] += (double)__uint_as_float((unsigned int)(key >> 32));
        acc[16] += 1.0;
    }
    __shared__ double red[4][17];
#pragma unroll
    for (int k = 0; k < 17; ++k) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 17)
        partials[(size_t)blockIdx.x * 17 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    // the device-resident loop on one GPU: the last workgroup to have written its row solves the pass (IcpFuse; k_icp_solve's body)
    if (fuse.st) {
        extern __shared__ double part[];
        __shared__ unsigned int last;
        __syncthreads();  // (the row is written)
        if (threadIdx.x == 0) {
            __threadfence();
            last = atomicAdd(fuse.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (last) {  // (block-uniform)
            __threadfence();
            if (threadIdx.x == 0) *fuse.ticket = 0u;  // ready for the next pass (stream order)
            icp_solve_block<true>(partials, (int)gridDim.x, fuse.st, fuse.max_iter, fuse.fixed, center, fuse.zero_word, part);
        }
    }
}


__global__ void __launch_bounds__(64)
k_icp_rows_to_sums(const double* __restrict__ partials, int n_blocks, double* __restrict__ sums) {
    if (threadIdx.x < 17) {
        double a = 0;
        for (int b = 0; b < n_blocks; ++b) a += partials[b * 17 + threadIdx.x];  // workgroup order: k_icp_solve's, the host loop's
        sums[threadIdx.x] = a;
    }
}
int launch_icp_rows_to_sums(hipStream_t s, const double* partials, int n_blocks, double* sums17) {
    hipLaunchKernelGGL(k_icp_rows_to_sums, dim3(1), dim3(64), 0, s, partials, n_blocks, sums17);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

int launch_icp_solve(hipStream_t s, const double* partials, int n_blocks, IcpState* state, int max_iter, int fixed,
                     const double* center_dev, unsigned int* zero_word) {
    hipLaunchKernelGGL(k_icp_solve, dim3(1), dim3(1024), (size_t)n_blocks * 17 * sizeof(double), s, partials, n_blocks, state,
                       max_iter, fixed, center_dev, zero_word);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

// The point the ICP sums are taken about: the first valid source point (any point OF the cloud keeps sum q p^T - n pm qm^T
// within a small factor of the covariance it is meant to be; the centre of a bounding box does not when stray points
// stretch the box).  Zero for an all-invalid cloud.
__global__ void __launch_bounds__(64)
k_icp_center(const float4* __restrict__ src, unsigned int n, double* __restrict__ center) {
    const unsigned int lane = threadIdx.x;
    for (unsigned int base = 0; base < n; base += 64) {  // wave-uniform
        const unsigned int i = base + lane;
        const float4 v = i < n ? src[i] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        const unsigned long long ok = __ballot(__float_as_int(v.w) >= 0);
        if (ok) {
            if (lane == (unsigned int)__builtin_ctzll(ok)) { center[0] = v.x; center[1] = v.y; center[2] = v.z; }
            return;
        }
    }
    if (lane == 0) center[0] = center[1] = center[2] = 0.0;
}
int launch_icp_center(hipStream_t s, const float4* src, size_t n, double* center_dev) {
    hipLaunchKernelGGL(k_icp_center, dim3(1), dim3(64), 0, s, src, (unsigned int)n, center_dev);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

int launch_icp_sums(hipStream_t s, const float4* src, size_t n, const unsigned long long* keys,
                    const float4* refs, double* partials, int* n_blocks, const unsigned int* mirror_dev,
                    unsigned int* mirror_host, const double* center, const IcpFuse* fuse) {
    size_t b = (n + 256 * 8 - 1) / (256 * 8);
    if (b < 1) b = 1;
    if (b > ICP_MAX_BLOCKS) b = ICP_MAX_BLOCKS;
    *n_blocks = (int)b;
    IcpFuse f{};
    if (fuse) f = *fuse;
    hipLaunchKernelGGL(k_icp_sums, dim3((unsigned)b), dim3(256), f.st ? b * 17 * sizeof(double) : 0, s, src, (unsigned int)n, keys, refs, partials,
                       mirror_dev, mirror_host, center, f);
    PCC_HIP(hipGetLastError());
    return PCC_OK;
}

}  // namespace pcc
