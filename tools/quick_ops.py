"""dev helper: time the secondary ops (SOR k=51, clustering, ICP) at benchmark scale on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), r

what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("sor", "all"):
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
    a = synth.corridor_cloud(n, synth.SEED_A)
    ix = capi.Index(a)
    dt, r = t(lambda: ix.sor(50, 1.5), 2)
    print(f"SOR n={n} k=51: {dt*1e3:.2f} ms  kept={r[3]}  thr={r[2]:.6f}", flush=True)
    ta = torch.from_numpy(a).cuda()
    dt, r = t(lambda: (ix.knn(ta, 51), ix.sync()), 2)
    print(f"kNN k=51 device n={n}: {dt*1e3:.2f} ms -> {n/dt/1e6:.1f} Mq/s", flush=True)
    ix.close()
if what in ("ec", "all"):
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5000000
    a = synth.corridor_cloud(n, synth.SEED_A, layer="objects")
    ix = capi.Index(a)
    dt, r = t(lambda: ix.euclidean_clusters(0.05, 100, 250000), 2)
    print(f"EC n={n}: {dt*1e3:.2f} ms  clusters={r[1]} sizes[:4]={r[2][:4]} min={r[2].min() if r[1] else 0}", flush=True)
    ix.close()
if what in ("icp", "all"):
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2000000
    tgt = synth.corridor_cloud(n, synth.SEED_A)
    src = synth.rigid_offset(synth.corridor_cloud(n, synth.SEED_B))
    tt, ts = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
    ix = capi.Index(tt)
    dt, r = t(lambda: ix.icp_align(ts, max_iter=50, fixed=True), 2)
    print(f"ICP n={n} 50 it: {dt*1e3:.2f} ms  ({dt/51*1e3:.3f} ms/NN pass, {51*n/dt/1e9:.2f} Gq/s) fitness={r[1]:.3e} it={r[2]}", flush=True)
    print(r[0])
    ix.close()
