import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth
for n, layer in ((1000000, "objects"), (5000000, "objects"), (1000000, "both")):
    a = synth.corridor_cloud(n, synth.SEED_A, layer=layer)
    ta = torch.from_numpy(a).cuda()
    ix = capi.Index(ta)
    for srt in (False, True):
        ix.radius_search(ta, 0.05, sorted=srt); torch.cuda.synchronize()
        t0 = time.perf_counter(); offs, idx, d2 = ix.radius_search(ta, 0.05, sorted=srt); ix.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"radius_search n={n} layer={layer} sorted={srt}: {dt*1e3:.1f} ms  total neighbours {int(offs[-1])}  mean degree {int(offs[-1])/n:.1f}", flush=True)
    t0 = time.perf_counter(); c = ix.radius_count(ta, 0.05); ix.sync(); torch.cuda.synchronize(); print(f"  count only: {(time.perf_counter()-t0)*1e3:.1f} ms")
    ix.close()
