"""dev helper: k-NN call time at 1M x 1M (self query), a few K; PCC_LIB selects the library.  usage: exp_knn.py [n] [K ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
Ks = [int(x) for x in sys.argv[2:]] or [51, 100]
scene = os.environ.get("SCENE", "corridor")
a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A) if scene == "corridor" else synth.room_cloud(n, synth.SEED_A)).cuda()
ix = capi.Index(a, auto_sync=False)
for K in Ks:
    for _ in range(2):
        ix.knn(a, K)
    ix.enable_timing(2)
    ix.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        ix.knn(a, K)
    ix.sync()
    dt = (time.perf_counter() - t0) / 5
    tm = ix.timing()
    ix.enable_timing(0)
    print(f"{scene} n={n} K={K}: call {dt * 1e3:.3f} ms, search kernels {tm[0] * 1e3:.1f} us, query sort {tm[4] * 1e3:.1f} us  lib={os.environ.get('PCC_LIB', 'default')[-20:]}", flush=True)
