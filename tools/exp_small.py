"""dev helper (round 6): the small-call regime -- matchRIFTFeaturesKnn's pattern (reference src/comparator.cpp:560-588): a handle
re-pointed at a descriptor cloud of n records of 128 bytes in HOST memory, then one k = 1 query per record of a second cloud,
us per (set_input + match_knn) with PCC_OPT_HOST_PIPE on and off and both tie modes, the CPU oracle beside it.
usage: exp_small.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from pointcloudcomparator_amd import capi

sizes = [int(float(x)) for x in sys.argv[1:]] or [4, 100, 1000, 4000, 18381]
rng = np.random.default_rng(0x51FF)


def per_call(fn, budget_s=0.3, kmin=5, kmax=3000):
    fn()
    t0 = time.perf_counter(); fn(); one = time.perf_counter() - t0
    k = int(max(kmin, min(kmax, budget_s / max(one, 1e-7))))
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    return (time.perf_counter() - t0) / k * 1e6


ix = capi.Index(np.zeros((4, 32), np.float32), auto_sync=False)
for n in sizes:
    d1 = np.round(rng.random((n, 32), dtype=np.float32) * 16) / np.float32(64)
    d2 = np.ascontiguousarray(d1[rng.permutation(n)] + (rng.random((n, 32), dtype=np.float32) < 0.3) * np.float32(1.0 / 64), dtype=np.float32)
    want = oracle.match_rift_knn(d1, d2)
    row = []
    for pipe in (1, 0):
        ix.set_option(capi.OPT_HOST_PIPE, pipe)
        for mode in (capi.TIES_LOWEST_INDEX, capi.TIES_FLANN):
            ix.set_tie_order(mode)

            def call():
                ix.set_input(d1)
                return ix.match_knn(d2)
            got = call()
            ok = mode != capi.TIES_FLANN or (len(got) == len(want) and (got == want).all())
            row.append(f"pipe {pipe} ties {'flann ' if mode else 'lowest'} {per_call(call):8.1f} us{'' if ok else ' MISMATCH'}")
    # descriptors without exact ties (continuous values): PCC_TIES_FLANN then needs no replay -- the kernel counts no tied query
    c1 = rng.random((n, 32), dtype=np.float32)
    c2 = np.ascontiguousarray(c1[rng.permutation(n)] + rng.random((n, 32), dtype=np.float32) * np.float32(0.01), dtype=np.float32)
    ix.set_option(capi.OPT_HOST_PIPE, 1)
    ix.set_tie_order(capi.TIES_FLANN)

    def call_c():
        ix.set_input(c1)
        return ix.match_knn(c2)
    got = call_c()
    wantc = oracle.match_rift_knn(c1, c2)
    okc = len(got) == len(wantc) and (got == wantc).all()
    row.append(f"pipe 1 ties flann, tie-free data {per_call(call_c):8.1f} us{'' if okc else ' MISMATCH'}")
    cpu = per_call(lambda: oracle.match_rift_knn(d1, d2))
    print(f"n = {n:6d}: " + "  |  ".join(row) + f"  |  cpu oracle {cpu:8.1f} us", flush=True)
ix.close()
