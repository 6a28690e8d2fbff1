import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth
for n in (2000, 5000, 10000, 20000, 50000, 100000, 200000):
    a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A)).cuda()
    b = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B)).cuda()
    idx = torch.empty(n, dtype=torch.int32, device='cuda'); d2 = torch.empty(n, dtype=torch.float32, device='cuda')
    res = []
    for eng in (capi.ENGINE_BRUTE, capi.ENGINE_GRID):
        ix = capi.Index(a, engine=eng)
        def step():
            ix.set_input(a); ix.nn1(b, idx, d2)
        for _ in range(5): step()
        ix.sync(); t0 = time.perf_counter()
        for _ in range(50): step()
        ix.sync(); dt = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for _ in range(50): ix.nn1(b, idx, d2)
        ix.sync(); dq = (time.perf_counter() - t0) / 50
        res.append((dt, dq)); ix.close()
    print(f"n={n:7d}: brute step {res[0][0]*1e6:8.1f} us query {res[0][1]*1e6:8.1f} us | grid step {res[1][0]*1e6:8.1f} us query {res[1][1]*1e6:8.1f} us", flush=True)
