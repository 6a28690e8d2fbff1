"""dev helper: time the NN engines on the GPU box (not part of the judged bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts), sorted(ts)[len(ts)//2]

m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else m
engines = sys.argv[3].split(',') if len(sys.argv) > 3 else ['grid', 'brute']
a = torch.from_numpy(synth.corridor_cloud(m, synth.SEED_A)).cuda()
b = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B)).cuda()
idx = torch.empty(n, dtype=torch.int32, device='cuda'); d2 = torch.empty(n, dtype=torch.float32, device='cuda')
res = {}
for e in engines:
    eng = capi.ENGINE_GRID if e == 'grid' else capi.ENGINE_BRUTE
    tb = timeit(lambda: capi.Index(a, engine=eng).close(), 3)
    ix = capi.Index(a, engine=eng)
    def run():
        ix.nn1(b, idx, d2); ix.sync()
    reps = 5 if e == 'grid' else 2
    tq = timeit(run, reps)
    print(f"{e}: m={m} n={n} build {tb[0]*1e3:.3f} ms  query min {tq[0]*1e3:.3f} ms median {tq[1]*1e3:.3f} ms  "
          f"-> {n/tq[0]/1e6:.2f} Mq/s  pairs/s {n*m/tq[0]:.3e} stats {ix.stats()}", flush=True)
    res[e] = (idx.cpu().numpy().copy(), d2.cpu().numpy().copy())
    ix.close()
if len(res) == 2:
    (i0, d0), (i1, d1) = res.values()
    print("engines agree:", (i0 == i1).all(), (d0.view(np.uint32) == d1.view(np.uint32)).all())
