"""dev helper: SOR (k = 51) and clustering on the room scene against the grid density.  usage: exp_room_ops.py n ppc..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudcomparator_amd import capi, synth
n = int(float(sys.argv[1]))
a = torch.from_numpy(synth.room_cloud(n, synth.SEED_A)).cuda()
ix = capi.Index(a, engine=capi.ENGINE_GRID)
for ppc in [float(x) for x in sys.argv[2:]]:
    ix.set_option(capi.OPT_GRID_PPC, ppc)
    ix.set_input(a)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ix.sor(50, 1.5)
        best = min(best, time.perf_counter() - t0)
    idx, d2 = ix.knn(a[:200000], 51); ix.sync()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, d2 = ix.knn(a[:200000], 51); ix.sync()
    t1 = time.perf_counter() - t0
    print(f"room n={n} ppc={ppc} sor {best*1e3:.2f} ms  knn51(200k) {t1*1e3:.2f} ms cells {ix.stats()[3]}", flush=True)
