"""dev helper: k_grid_nn1 time by query layer (background vs objects) against the full C2 reference cloud."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth
m = 1000000
a = torch.from_numpy(synth.corridor_cloud(m, synth.SEED_A)).cuda()
ix = capi.Index(a, engine=capi.ENGINE_GRID)
for layer in ("both", "background", "objects"):
    for n in (1000000, 250000):
        b = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B, layer=layer)).cuda()
        idx = torch.empty(n, dtype=torch.int32, device='cuda'); d2 = torch.empty(n, dtype=torch.float32, device='cuda')
        for _ in range(3): ix.nn1(b, idx, d2)
        ix.sync(); ix.enable_timing(2)
        for _ in range(10): ix.nn1(b, idx, d2)
        t = ix.timing(); ix.enable_timing(0)
        print(f"{layer:10s} n={n:8d}: main kernel {t[0]*1e3:7.1f} us  sort {t[4]*1e3:6.1f} us  call {t[2]*1e3:7.1f} us  stats {ix.stats()[:2]}", flush=True)
