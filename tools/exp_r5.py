"""dev helper (round 5): the k = 1 kernels at C3 / C2 under the options named in OPTS ("name=value,..." sets, ';' separated),
results compared bit for bit with the first set.  usage: exp_r5.py [n ...];  PCC_LIB selects an ablation build (times only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

sizes = [int(float(x)) for x in sys.argv[1:]] or [10_000_000]
sets = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in st.split(",") if kv) for st in os.environ.get("OPTS", "").split(";")]
def cloud(n, seed, chunk=4_000_000):
    parts = [synth.corridor_cloud(min(chunk, n - o), seed, start=o) for o in range(0, n, chunk)]
    return parts[0] if len(parts) == 1 else np.concatenate(parts)
for n in sizes:
    a = torch.from_numpy(cloud(n, synth.SEED_A)).cuda()
    b = torch.from_numpy(cloud(n, synth.SEED_B)).cuda()
    ref = None
    for st in sets:
        ix = capi.Index(a, engine=capi.ENGINE_GRID)
        for k, v in st.items():
            ix.set_option(getattr(capi, "OPT_" + k.upper()), v)
        ix.set_input(a)
        idx = torch.empty(n, dtype=torch.int32, device="cuda")
        d2 = torch.empty(n, dtype=torch.float32, device="cuda")
        for _ in range(3):
            ix.nn1(b, idx, d2)
        ix.enable_timing(2)
        for _ in range(10):
            ix.set_input(a)
            ix.nn1(b, idx, d2)
        tm = ix.timing()
        ix.enable_timing(0)
        torch.cuda.synchronize()
        stt = ix.stats()
        if ref is None:
            ref = (idx.clone(), d2.clone()); same = "ref"
        else:
            same = f"idx_equal={bool((idx == ref[0]).all())} d2_equal={bool((d2.view(torch.int32) == ref[1].view(torch.int32)).all())}"
        print(f"n={n} {st} main {tm[0]*1e3:8.1f} us fallback {tm[1]*1e3:6.1f} call {tm[2]*1e3:8.1f} build {tm[3]*1e3:7.1f} sort {tm[4]*1e3:7.1f} open={stt[7]} fb={stt[1]} {same}", flush=True)
        ix.close()
