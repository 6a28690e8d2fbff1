"""dev helper: the C4 configuration (2M x 2M, fixed ICP iterations) as one call, for rocprofv3 / PMC passes, and the
open-lane share of an aligned and a misaligned k = 1 pass.  usage: exp_icp.py [n] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
tgt = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A)).cuda()
src_h = synth.rigid_offset(synth.corridor_cloud(n, synth.SEED_B))
src = torch.from_numpy(src_h).cuda()
ali = torch.from_numpy(synth.rigid_offset(synth.corridor_cloud(n, synth.SEED_B), rot_deg=0.0, t=(0, 0, 0))).cuda()
ix = capi.Index(tgt, auto_sync=False)
for name, q in (("misaligned", src), ("aligned", ali)):
    ix.set_option(capi.OPT_NN1_KERNEL, 2)
    ix.nn1(q)
    st = ix.stats()
    print(f"{name}: open lanes {st[7]} of {n} = {st[7] / n:.3f}; fallback {st[1]}", flush=True)
ix.set_option(capi.OPT_NN1_KERNEL, 1)
ix.icp_align(src, max_iter=iters, fixed=True)
ix.enable_timing(2)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T, fit, it, conv = ix.icp_align(src, max_iter=iters, fixed=True)
    best = min(best, time.perf_counter() - t0)
tm = ix.timing()
print(f"icp {iters} iterations: {best * 1e3:.2f} ms; per pass nn {tm[0] * 1e3:.1f} us far {tm[1] * 1e3:.1f} us total {tm[2] * 1e3:.1f} us", flush=True)
