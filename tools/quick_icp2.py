import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from pointcloudcomparator_amd import capi, synth
n = 2000000
tgt = synth.corridor_cloud(n, synth.SEED_A)
src = synth.rigid_offset(synth.corridor_cloud(n, synth.SEED_B))
tt, cur = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
ix = capi.Index(tt)
idx = torch.empty(n, dtype=torch.int32, device='cuda'); d2 = torch.empty(n, dtype=torch.float32, device='cuda')
for it in range(8):
    ix.enable_timing(2)
    ix.nn1(cur, idx, d2)
    t = ix.timing(); st = ix.stats(); ix.enable_timing(0)
    _, _, sums = ix.icp_step(cur, want_corr=False)
    rc, T = oracle.umeyama_from_sums(sums)
    cur = ix.transform(T, cur)
    print(f"it {it}: main {t[0]*1e3:.0f} us fallback {t[1]*1e3:.0f} us call {t[2]*1e3:.0f} us stats {st[:2]} mse {sums[15]/sums[16]:.5f}", flush=True)
