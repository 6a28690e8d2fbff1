"""dev helper: k_grid_nn1 (one lane per query) against k_grid_nn1_flat (PCC_OPT_NN1_KERNEL 1 = hybrid, 2 = every pass
flat), same index, results compared bit for bit.  usage: exp_flat.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

sizes = [int(float(x)) for x in sys.argv[1:]] or [1_000_000]
for n in sizes:
    for layer in os.environ.get("SCENES", "both,background,objects").split(","):
        a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A, layer=layer)).cuda()
        b = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B, layer=layer)).cuda()
        ix = capi.Index(a, engine=capi.ENGINE_GRID)
        ref = None
        for mode in [int(x) for x in os.environ.get("MODES", "0,1").split(",")]:
            ix.set_option(capi.OPT_NN1_KERNEL, mode)
            idx = torch.empty(n, dtype=torch.int32, device="cuda")
            d2 = torch.empty(n, dtype=torch.float32, device="cuda")
            for _ in range(3):
                ix.nn1(b, idx, d2)
            ix.enable_timing(2)
            for _ in range(10):
                ix.nn1(b, idx, d2)
            tm = ix.timing()
            ix.enable_timing(0)
            torch.cuda.synchronize()
            if ref is None:
                ref = (idx.clone(), d2.clone())
                same = "ref"
            else:
                same = f"idx_equal={bool((idx == ref[0]).all())} d2_equal={bool((d2.view(torch.int32) == ref[1].view(torch.int32)).all())}"
            print(f"n={n} {layer:10s} nn1_kernel={mode} main {tm[0]*1e3:8.1f} us  fallback {tm[1]*1e3:7.1f} us  call {tm[2]*1e3:8.1f} us  {same}", flush=True)
        ix.close()
