"""dev helper: k_grid_nn1 time by scene layer (uniform background / dense balls / both) -- the
density-landscape experiment behind the adaptive grid.  Not part of the judged bench.
usage: exp_nn1.py [n] [layers]   (env PCC_GRID_PPC etc. are read by the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
layers = sys.argv[2].split(",") if len(sys.argv) > 2 else ["both", "background", "objects"]
for layer in layers:
    a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A, layer=layer)).cuda()
    b = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B, layer=layer)).cuda()
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    d2 = torch.empty(n, dtype=torch.float32, device="cuda")
    ix = capi.Index(a, engine=capi.ENGINE_GRID)
    for _ in range(3):
        ix.nn1(b, idx, d2)
    ix.enable_timing(2)
    for _ in range(10):
        ix.set_input(a)
        ix.nn1(b, idx, d2)
    tm = ix.timing()
    st = ix.stats()
    print(f"{layer:10s} n={n} ppc={os.environ.get('PCC_GRID_PPC', 'default')} main {tm[0]*1e3:8.1f} us  fallback {tm[1]*1e3:7.1f} us  "
          f"call {tm[2]*1e3:8.1f} us  build {tm[3]*1e3:7.1f} us  qsort {tm[4]*1e3:7.1f} us  cells {st[3]}  fb {st[1]}", flush=True)
    ix.close()
