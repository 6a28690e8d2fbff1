// host_pipe.hpp -- clouds and results between PAGEABLE host memory and the device, for api.hip (host code only).
//
// Every drop-in call site of the reference starts from host vectors (src/comparator.cpp:1119,1130 load the clouds to
// host memory; :571-577 hand std::vectors over).  A hipMemcpyAsync from pageable memory moves the RAW array: 10M XYZRGB
// points are 320 MB for 120 MB of coordinates, 6.2 ms per cloud at the ~52 GB/s the runtime's staging reaches (a C3 step from and
// to host memory: 14.4 ms for 1.3 ms of device work).  Here the staging is done by the library:
//   upload   : a few host threads gather the cloud chunk by chunk into two pinned buffers -- only the 12 bytes of x, y, z
//              of every point when the stride is 24 bytes or more (PointXYZRGB: 12 of 32) -- and each chunk's DMA runs while
//              the next one is being gathered; the pack kernel then reads a 12-byte-stride cloud.
//   download : the mirror image: DMA into the pinned buffers, the threads copy out to the caller's arrays.
// Memory the caller has pinned itself (hipHostMalloc / hipHostRegister) is recognised and copied directly.
// Small transfers (below PIPE_MIN_BYTES) keep the plain hipMemcpyAsync: the threads cost more than they save.
#pragma once
#include "pcc_internal.hpp"
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

namespace pcc {

constexpr size_t PIPE_MIN_BYTES = 8u << 20;   // transfers from here on are pipelined
constexpr size_t PIPE_CHUNK_BYTES = 8u << 20;  // pinned bytes per chunk (two buffers)
constexpr size_t SMALL_DIRECT_BYTES = 1u << 20;  // a host cloud up to here is read by the pack kernel straight from its pinned copy
constexpr size_t SMALL_RESULT_BYTES = 1u << 20;  // a host result array up to here is written by the unpack kernel straight into pinned memory

inline bool host_pointer_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

inline int pipe_threads() {
    unsigned int hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 4;
    unsigned int t = hw / 2;
    if (t < 2) t = 2;
    if (t > 8) t = 8;  // (4 threads already gather at the host's memory bandwidth, ~80 GB/s: tools/exp_host.py)
    if (const char* e = getenv("PCC_HOST_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) t = (unsigned int)v;
    }
    return (int)t;
}

// Fork-join helpers that live for ONE transfer: `work(chunk, tid, nthreads)` is run by every thread for chunk 0, 1, ...
// as the owner releases them; the owner is thread 0 and does its share too.
class ChunkCrew {
public:
    template <class F>
    ChunkCrew(int nthreads, F work) : n_(1) {
        // (a helper that cannot be started -- thread limits of the process -- is done without: the owner alone is a crew of one.  n_ is
        // final before any helper reads it: they wait for the first release, which run() makes after this constructor)
        int started = 1;
        for (int t = 1; t < nthreads; ++t) {
            try {
                helpers_.emplace_back([this, t, work]() {
                for (long next = 0;; ++next) {
                    long rel;
                    while ((rel = released_.load(std::memory_order_acquire)) <= next) {
                        if (rel < 0) return;
                        std::this_thread::yield();
                    }
                    if (rel < 0) return;
                    work((size_t)next, t, n_);
                    done_.fetch_add(1, std::memory_order_release);
                }
                });
                ++started;
            } catch (...) {
                break;
            }
        }
        n_ = started;
    }
    // run chunk `c` on every thread and wait for all of them (chunks are released in order 0, 1, 2, ...)
    template <class F>
    void run(size_t c, F work) {
        released_.store((long)c + 1, std::memory_order_release);
        work(c, 0, n_);
        const long want = (long)(c + 1) * (n_ - 1);
        while (done_.load(std::memory_order_acquire) < want) std::this_thread::yield();
    }
    ~ChunkCrew() {
        released_.store(-1, std::memory_order_release);
        for (auto& h : helpers_) h.join();
    }

private:
    int n_;
    std::atomic<long> released_{0}, done_{0};
    std::vector<std::thread> helpers_;
};

// chunk size of a transfer (PCC_PIPE_CHUNK_MB: measurements only).  C3 from host memory, 10M XYZRGB points, step ms at 8 / 16 / 32 / 64 MB:
// 9.4-10.0 / 9.6-9.8 / 10.6-11.1 / 11.6-12.0 -- larger chunks overlap less at both ends; 4 / 8 / 12 / 16 / 24 threads: 10.1 / 10.2 / 10.7 /
// 10.9 / 11.8: the gather is bound by the host's memory bandwidth (320 MB read in ~4 ms), not by the number of threads.
inline size_t pipe_chunk_bytes(size_t) {
    if (const char* e = getenv("PCC_PIPE_CHUNK_MB")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 256) return (size_t)v << 20;
    }
    return PIPE_CHUNK_BYTES;
}

struct HostPipe {
    HostBuf buf[2];
    hipEvent_t ev[2] = {nullptr, nullptr};
    int init(size_t chunk = PIPE_CHUNK_BYTES) {
        for (int b = 0; b < 2; ++b) {
            PCC_TRY(buf[b].reserve(chunk));
            if (!ev[b]) PCC_HIP(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
        }
        return PCC_OK;
    }
    void release() {
        for (int b = 0; b < 2; ++b) {
            buf[b].release();
            if (ev[b]) { (void)hipEventDestroy(ev[b]); ev[b] = nullptr; }
        }
    }

    // n points of `stride` bytes at `src` (pageable host memory) -> device `dst`, packed to `dst_stride` bytes per point
    // (dst_stride == stride: plain copy; dst_stride == 12: x, y, z only).  Enqueued on `s`; returns when the last chunk's DMA
    // has been enqueued and its source buffer is safe (the caller's memory is no longer read).
    // whoever used the two buffers last -- a small cloud's pack kernel reads its buffer in place (api.hip) -- has finished
    int buffers_free() {
        for (int b = 0; b < 2; ++b)
            if (ev[b]) PCC_HIP(hipEventSynchronize(ev[b]));
        return PCC_OK;
    }
    int upload(hipStream_t s, const char* src, size_t n, size_t stride, char* dst, size_t dst_stride) {
        const size_t CH = pipe_chunk_bytes(n * dst_stride);
        PCC_TRY(buffers_free());  // (before a reserve that may free them)
        PCC_TRY(init(CH));
        const size_t per_chunk = CH / dst_stride;
        const size_t nchunks = (n + per_chunk - 1) / per_chunk;
        const bool gather = dst_stride != stride;
        auto fill = [=](size_t c, int tid, int nt) {
            const size_t p0 = c * per_chunk, cnt = (p0 + per_chunk <= n ? per_chunk : n - p0);
            const size_t a = cnt * (size_t)tid / (size_t)nt, b = cnt * (size_t)(tid + 1) / (size_t)nt;
            char* out = buf[c & 1].as<char>();
            if (!gather) {
                // (the last point of the cloud may end before a full stride: only its first 12 bytes are the caller's)
                size_t bytes = (b - a) * stride;
                if (p0 + b == n && b > a) bytes = (b - a - 1) * stride + 12;
                memcpy(out + a * stride, src + (p0 + a) * stride, bytes);
            } else {
                const char* in = src + (p0 + a) * stride;
                float* o = reinterpret_cast<float*>(out + a * 12);
                for (size_t i = a; i < b; ++i, in += stride, o += 3) {
                    const float* f = reinterpret_cast<const float*>(in);
                    o[0] = f[0]; o[1] = f[1]; o[2] = f[2];
                }
            }
        };
        ChunkCrew crew(pipe_threads(), fill);
        for (size_t c = 0; c < nchunks; ++c) {
            const int b = (int)(c & 1);
            if (c >= 2) PCC_HIP(hipEventSynchronize(ev[b]));  // the DMA that last read this buffer has finished
            crew.run(c, fill);
            const size_t p0 = c * per_chunk, cnt = (p0 + per_chunk <= n ? per_chunk : n - p0);
            size_t bytes = cnt * dst_stride;
            if (!gather && p0 + cnt == n) bytes = (cnt - 1) * stride + 12;
            PCC_HIP(hipMemcpyAsync(dst + p0 * dst_stride, buf[b].p, bytes, hipMemcpyHostToDevice, s));
            PCC_HIP(hipEventRecord(ev[b], s));
        }
        // the buffers are reused by the next transfer: wait for the last two DMAs (the caller's memory was released earlier)
        for (int b = 0; b < 2 && b < (int)nchunks; ++b) PCC_HIP(hipEventSynchronize(ev[b]));
        return PCC_OK;
    }

    // bytes at device `src` -> pageable host `dst`; everything enqueued on `s` before is waited for (the first DMA is behind it)
    int download(hipStream_t s, const char* src, char* dst, size_t bytes) {
        const size_t CH = pipe_chunk_bytes(bytes);
        PCC_TRY(buffers_free());
        PCC_TRY(init(CH));
        const size_t nchunks = (bytes + CH - 1) / CH;
        auto drain = [=](size_t c, int tid, int nt) {
            const size_t o0 = c * CH, cnt = (o0 + CH <= bytes ? CH : bytes - o0);
            const size_t a = (cnt * (size_t)tid / (size_t)nt) & ~(size_t)63, b = tid + 1 == nt ? cnt : (cnt * (size_t)(tid + 1) / (size_t)nt) & ~(size_t)63;
            if (b > a) memcpy(dst + o0 + a, buf[c & 1].as<char>() + a, b - a);
        };
        ChunkCrew crew(pipe_threads(), drain);
        auto enqueue = [&](size_t c) -> int {
            const size_t o0 = c * CH, cnt = (o0 + CH <= bytes ? CH : bytes - o0);
            PCC_HIP(hipMemcpyAsync(buf[c & 1].p, src + o0, cnt, hipMemcpyDeviceToHost, s));
            PCC_HIP(hipEventRecord(ev[c & 1], s));
            return PCC_OK;
        };
        PCC_TRY(enqueue(0));
        for (size_t c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) PCC_TRY(enqueue(c + 1));  // (its buffer was drained at c - 1)
            PCC_HIP(hipEventSynchronize(ev[c & 1]));
            crew.run(c, drain);
        }
        return PCC_OK;
    }
};

}  // namespace pcc
