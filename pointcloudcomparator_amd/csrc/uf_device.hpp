// uf_device.hpp -- lock-free union-find on the device (agent-scope atomics).  Roots only ever point
// to smaller ids, so a component's root is its lowest member.  Used by the Euclidean clustering
// (cluster.hip) and by region growing (region.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace pcc {

__device__ __forceinline__ unsigned int uf_find(unsigned int* __restrict__ parent, unsigned int x) {
    // path halving; racing writers only ever replace a parent by one of its ancestors
    for (;;) {
        unsigned int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        unsigned int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) atomicMin(&parent[x], gp);  // monotone: never undoes a concurrent link
        x = p;
    }
}
__device__ __forceinline__ void uf_union(unsigned int* __restrict__ parent, unsigned int a, unsigned int b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { unsigned int t = a; a = b; b = t; }  // a > b: hang the larger root under the smaller
        unsigned int old = atomicMin(&parent[a], b);
        if (old == a) return;  // a was still a root: linked
        a = old;               // somebody re-parented a meanwhile: retry from there
    }
}

}  // namespace pcc
