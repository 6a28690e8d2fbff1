"""dev helper: per-call breakdown of one ICP correspondence pass at C4 size"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth
n = 2000000
tgt = synth.corridor_cloud(n, synth.SEED_A)
src = synth.rigid_offset(synth.corridor_cloud(n, synth.SEED_B))
tt, ts = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
ix = capi.Index(tt)
idx = torch.empty(n, dtype=torch.int32, device='cuda'); d2 = torch.empty(n, dtype=torch.float32, device='cuda')
for name, q in (("offset source (iteration 1)", ts), ("aligned source (same cloud B)", torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B)).cuda())):
    for _ in range(2): ix.nn1(q, idx, d2)
    ix.sync(); ix.enable_timing(2)
    for _ in range(5): ix.nn1(q, idx, d2)
    t = ix.timing(); ix.enable_timing(0)
    print(f"{name}: main {t[0]*1e3:.0f} us  fallback {t[1]*1e3:.0f} us  sort {t[4]*1e3:.0f} us  call {t[2]*1e3:.0f} us  stats {ix.stats()[:2]}", flush=True)
t0 = time.perf_counter(); r = ix.icp_step(ts, want_corr=False); t1 = time.perf_counter()
t0 = time.perf_counter(); r = ix.icp_step(ts, want_corr=False); t1 = time.perf_counter()
print(f"icp_step wall {1e3*(t1-t0):.3f} ms")
