"""dev helper (round 6): 1000 small calls (set_input + match_knn on 100 descriptors of 128 bytes in host memory, lowest-index ties, then
the same in FLANN's tie order on tie-free descriptors) for a kernel trace: what a call launches.
usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/exp_small_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pointcloudcomparator_amd import capi

rng = np.random.default_rng(5)
d1 = rng.random((100, 32), dtype=np.float32)
d2 = np.ascontiguousarray(d1[rng.permutation(100)] + rng.random((100, 32), dtype=np.float32) * np.float32(0.01), dtype=np.float32)
ix = capi.Index(d1, auto_sync=False)
for mode in (capi.TIES_LOWEST_INDEX, capi.TIES_FLANN):
    ix.set_tie_order(mode)
    for _ in range(1000):
        ix.set_input(d1)
        ix.match_knn(d2)
ix.close()
print("2 x 1000 calls done")
