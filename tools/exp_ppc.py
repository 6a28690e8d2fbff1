"""dev helper: grid density sweep (PCC_OPT_GRID_PPC) for the two k = 1 kernel forms.  usage: exp_ppc.py n scene ppc..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

n = int(float(sys.argv[1]))
scene = sys.argv[2]
ppcs = [float(x) for x in sys.argv[3:]] or [0.5]
if scene == "room":
    a = torch.from_numpy(synth.room_cloud(n, synth.SEED_A)).cuda()
    b = torch.from_numpy(synth.room_cloud(n, synth.SEED_B)).cuda()
else:
    a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A, layer=scene)).cuda()
    b = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B, layer=scene)).cuda()
idx = torch.empty(n, dtype=torch.int32, device="cuda")
d2 = torch.empty(n, dtype=torch.float32, device="cuda")
ix = capi.Index(a, engine=capi.ENGINE_GRID)
for ppc in ppcs:
    ix.set_option(capi.OPT_GRID_PPC, ppc)
    for mode in (0, 1):
        ix.set_option(capi.OPT_NN1_KERNEL, mode)
        ix.set_input(a)
        for _ in range(3):
            ix.nn1(b, idx, d2)
        ix.enable_timing(2)
        for _ in range(8):
            ix.set_input(a)
            ix.nn1(b, idx, d2)
        tm = ix.timing()
        ix.enable_timing(0)
        st = ix.stats()
        print(f"n={n} {scene:10s} ppc={ppc:4.2f} kernel={mode} main {tm[0]*1e3:8.1f} us  call {tm[2]*1e3:8.1f} us  build {tm[3]*1e3:7.1f} us  qsort {tm[4]*1e3:7.1f}  cells {st[3]} fb {st[1]}", flush=True)
