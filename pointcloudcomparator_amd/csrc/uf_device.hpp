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
// the root of x when NO union runs any more (the flatten pass after the links, a kernel of its own): parents only ever
// move towards the root, so any value a cache still holds is an ancestor and the walk ends at the same root -- plain
// loads through the caches instead of agent-scope ones past them, no halving writes
__device__ __forceinline__ unsigned int uf_find_settled(const unsigned int* __restrict__ parent, unsigned int x) {
    for (;;) {
        const unsigned int p = parent[x];
        if (p == x) return x;
        x = p;
    }
}
// are a and b in one component (now)?  Both walks advance together -- their loads are in flight at the same time -- and
// stop as soon as they meet: a parent is an ancestor, and a common ancestor proves the link before either root is reached
// (already-linked points one step below their root, the usual case late in a pass, cost ONE round trip instead of six).
__device__ __forceinline__ bool uf_linked(unsigned int* __restrict__ parent, unsigned int a, unsigned int b) {
    for (;;) {
        if (a == b) return true;
        const unsigned int pa = __hip_atomic_load(&parent[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int pb = __hip_atomic_load(&parent[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pa == pb || pa == b || pb == a) return true;
        const bool ra = pa == a, rb = pb == b;
        if (ra && rb) return false;  // two different roots
        const unsigned int gpa = __hip_atomic_load(&parent[pa], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int gpb = __hip_atomic_load(&parent[pb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!ra && gpa != pa) atomicMin(&parent[a], gpa);  // path halving, as in uf_find
        if (!rb && gpb != pb) atomicMin(&parent[b], gpb);
        a = pa;
        b = pb;
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
