"""dev helper (round 5): the bench step -- setInputCloud + k = 1 search, back to back on one handle, no host wait -- under the
option sets of OPTS ("name=value,..." sets, ';' separated; default: the query staging behind / beside the build), alternating
on one box; results compared bit for bit.
usage: exp_overlap.py [n ...]   (n references = n queries; optional NQ=<queries> for a shard of the queries)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

sizes = [int(float(x)) for x in sys.argv[1:]] or [10_000_000]
steps = int(os.environ.get("STEPS", "20"))
rounds = int(os.environ.get("ROUNDS", "3"))
sets = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in st.split(",") if kv)
        for st in os.environ.get("OPTS", "overlap_prep=0;overlap_prep=1").split(";")]
def cloud(n, seed, chunk=4_000_000):
    parts = [synth.corridor_cloud(min(chunk, n - o), seed, start=o) for o in range(0, n, chunk)]
    return parts[0] if len(parts) == 1 else np.concatenate(parts)
for n in sizes:
    nq = int(float(os.environ.get("NQ", n)))
    a = torch.from_numpy(cloud(n, synth.SEED_A)).cuda()
    b = torch.from_numpy(cloud(n, synth.SEED_B)[:nq].copy()).cuda()
    torch.cuda.synchronize()
    ix = capi.Index(a, engine=capi.ENGINE_GRID, auto_sync=False)
    idx = torch.empty(nq, dtype=torch.int32, device="cuda")
    d2 = torch.empty(nq, dtype=torch.float32, device="cuda")
    ref = None
    for r in range(rounds):
        for st in sets:
            for k, v in st.items():
                ix.set_option(getattr(capi, "OPT_" + k.upper()), v)
            for _ in range(3):
                ix.set_input(a)
                ix.nn1(b, idx, d2)
            ix.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                ix.set_input(a)
                ix.nn1(b, idx, d2)
            ix.sync()
            ms = (time.perf_counter() - t0) / steps * 1e3
            if ref is None:
                ref = (idx.clone(), d2.clone()); same = "ref"
            else:
                same = f"idx_equal={bool((idx == ref[0]).all())} d2_equal={bool((d2.view(torch.int32) == ref[1].view(torch.int32)).all())}"
            print(f"n={n} nq={nq} {st} step {ms:7.4f} ms  {same}", flush=True)
    ix.close()
