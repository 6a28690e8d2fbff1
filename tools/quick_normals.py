"""timing of the default segmentation path at scale: normals K=50 and region growing K=100"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pointcloudcomparator_amd import capi, synth
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pts = synth.corridor_cloud(n, synth.SEED_A)
d = torch.from_numpy(pts).cuda()
ix = capi.Index(d)
for k in (50,):
    ix.normals(k, device="cuda:0")
    t = time.perf_counter()
    for _ in range(3):
        nrm = ix.normals(k, device="cuda:0")
    dt = (time.perf_counter() - t) / 3
    print(f"normals n={n} k={k}: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} M pts/s", flush=True)
ix.enable_timing(2)
nrm = ix.normals(50, device="cuda:0")
print("timing", ix.timing())
ix.enable_timing(0)
t = time.perf_counter()
lab, ncl = ix.region_growing(nrm, k=100)
dt = time.perf_counter() - t
print(f"region growing n={n} k=100: {dt*1e3:.1f} ms, {ncl} clusters", flush=True)
t = time.perf_counter()
lab, ncl = ix.region_growing(nrm, k=30)
dt = time.perf_counter() - t
print(f"region growing n={n} k=30: {dt*1e3:.1f} ms, {ncl} clusters", flush=True)
