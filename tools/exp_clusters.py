"""dev helper: EuclideanClusterExtraction(0.05, 100, 250000) over the object layer (the C3 -e leg) and the room's furniture under
the forms of PCC_OPT_EC_CELLS; labels compared with the first form.  usage: exp_clusters.py [n] [forms, e.g. 3,1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
forms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3,1").split(",")]
scenes = [("objects", synth.corridor_cloud(n, synth.SEED_A, layer="objects"), 250000),
          ("furniture", synth.room_cloud(max(n // 4, 100_000), synth.SEED_A, part="furniture"), 2_000_000_000),
          ("both layers", synth.corridor_cloud(n, synth.SEED_A), 2_000_000_000)]
for name, pts, mx in scenes:
    obj = torch.from_numpy(pts).cuda()
    m = len(pts)
    ref = None
    for form in forms:
        ix = capi.Index(obj, auto_sync=False)
        ix.set_option(capi.OPT_EC_CELLS, form)
        labels = torch.empty(m, dtype=torch.int32, device="cuda")
        ix.euclidean_clusters(0.05, 100, mx, device_out=labels)
        ix.sync()
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lab, ncl, sizes = ix.euclidean_clusters(0.05, 100, mx, device_out=labels)
            ix.sync()
            best = min(best, time.perf_counter() - t0)
        lab = torch.as_tensor(lab)
        if ref is None:
            ref = lab.clone(); same = "ref"
        else:
            same = f"labels_equal={bool((lab == ref).all())}"
        print(f"{name:12s} n={m} form={form} {best * 1e3:7.3f} ms clusters={ncl} clustered={int(np.asarray(sizes).sum())} {same}", flush=True)
        ix.close()
