"""dev helper (round 6): k-NN self query on ONE handle under alternating option sets (same memory for every set), medians.
usage: exp_knn_ab.py <corridor|room> <n> <K> "grid_axes=0" "grid_axes=-2" ..."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

scene, n, K = sys.argv[1], int(float(sys.argv[2])), int(sys.argv[3])
sets = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in s.split(",") if kv) for s in sys.argv[4:]]
a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A) if scene == "corridor" else synth.room_cloud(n, synth.SEED_A)).cuda()
ix = capi.Index(a, auto_sync=False)
keys = sorted({k for st in sets for k in st})
defaults = {k: ix.get_option(getattr(capi, "OPT_" + k.upper())) for k in keys}
res = [[] for _ in sets]
for r in range(int(os.environ.get("ROUNDS", "5"))):
    for i, st in enumerate(sets):
        for k in keys:
            ix.set_option(getattr(capi, "OPT_" + k.upper()), st.get(k, defaults[k]))
        ix.set_input(a)
        for _ in range(2):
            ix.knn(a, K)
        ix.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            ix.knn(a, K)
        ix.sync()
        res[i].append((time.perf_counter() - t0) / 5 * 1e3)
for st, v in zip(sets, res):
    print(f"{scene} {n} K={K} {str(st):40s} call {statistics.median(v):7.4f} ms (min {min(v):7.4f})", flush=True)
