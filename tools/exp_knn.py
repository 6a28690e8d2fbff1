"""dev helper: self k-NN (k = 51 by default) over the 1M-point corridor scene, a few calls (for rocprofv3 --pmc)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 51
a = synth.corridor_cloud(n, synth.SEED_A)
ta = torch.from_numpy(a).cuda()
ix = capi.Index(ta)
for _ in range(4):
    t0 = time.perf_counter(); ix.knn(ta, k); ix.sync(); dt = time.perf_counter() - t0
print(f"knn k={k} n={n}: {dt*1e3:.2f} ms")
