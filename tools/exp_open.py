"""dev helper: the listed form of the k = 1 kernel (PCC_OPT_NN1_KERNEL = 2) with the open lanes drained flat or walked by lane,
against the in-place form, step by step with progress lines.  usage: exp_open.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A)).cuda()
b = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B)).cuda()
print("clouds up", flush=True)
with capi.Index(a, engine=capi.ENGINE_GRID) as ix:
    ix.set_option(capi.OPT_NN1_KERNEL, 3)
    i0, d0 = ix.nn1(b)
    torch.cuda.synchronize()
    print("in-place form done", flush=True)
    for flat in (0, 1):
        ix.set_option(capi.OPT_NN1_KERNEL, 2)
        ix.set_option(capi.OPT_NN1_OPEN_FLAT, flat)
        i1, d1 = ix.nn1(b)
        torch.cuda.synchronize()
        print(f"listed form, open_flat={flat}: idx equal {bool((i0 == i1).all())}, d2 equal {bool((d0.view(torch.int32) == d1.view(torch.int32)).all())}, "
              f"open lanes {ix.stats()[7]}", flush=True)
        ix.enable_timing(2)
        for _ in range(5):
            ix.nn1(b)
        tm = ix.timing()
        ix.enable_timing(0)
        print(f"   main kernels {tm[0] * 1e3:.1f} us", flush=True)
