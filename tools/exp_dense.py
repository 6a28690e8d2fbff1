"""dev helper: PCC_OPT_NN1_DENSE_MIN sweep.  usage: exp_dense.py n scene v..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudcomparator_amd import capi, synth
n = int(float(sys.argv[1])); scene = sys.argv[2]
a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A, layer=scene)).cuda()
b = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B, layer=scene)).cuda()
idx = torch.empty(n, dtype=torch.int32, device="cuda"); d2 = torch.empty(n, dtype=torch.float32, device="cuda")
ix = capi.Index(a, engine=capi.ENGINE_GRID)
for v in [float(x) for x in sys.argv[3:]]:
    ix.set_option(capi.OPT_NN1_DENSE_MIN, v)
    for _ in range(3): ix.nn1(b, idx, d2)
    ix.enable_timing(2)
    for _ in range(10): ix.nn1(b, idx, d2)
    tm = ix.timing(); ix.enable_timing(0)
    print(f"n={n} {scene} dense_min={v} main {tm[0]*1e3:8.1f} us", flush=True)
