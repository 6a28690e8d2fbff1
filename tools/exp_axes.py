"""dev helper (round 6): same-box A/B of the grid's axis assignment (PCC_OPT_GRID_AXES) and the XCD run length
(PCC_OPT_XCD_RUN) over everything that walks cells -- C3 / C2 / room scan k = 1, C4 ICP, K = 51, radius fill, clustering.
Times are the library's own HIP-event brackets (pcc_index_enable_timing) or wall clock around synchronised calls.
usage: exp_axes.py [legs]   legs: c3,c2,room,c4,knn,radius,clusters (default all)
Not part of the judged bench; its output is kept as profiles/r06_exp_axis_order.txt."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

legs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c3", "c2", "room", "c4", "knn", "radius", "clusters"]
AXES = [int(x) for x in os.environ.get("AXES", "0,-1").split(",")]
RUNS = [int(x) for x in os.environ.get("XCD_RUNS", "32").split(",")]
NAMES = {-2: "by extent!", -1: "by extent", 0: "xyz (r5)", 1: "xzy", 2: "yxz", 3: "yzx", 4: "zxy", 5: "zyx"}


def cloud(kind, n, seed, rgb=False):
    chunk = 4_000_000
    gen = synth.room_cloud if kind == "room" else synth.corridor_cloud
    kw = {} if kind in ("room", "both") else {"layer": kind}
    parts = [gen(min(chunk, n - o), seed, start=o, **kw) for o in range(0, n, chunk)]
    pts = parts[0] if len(parts) == 1 else np.concatenate(parts)
    return torch.from_numpy(synth.with_rgb_stride(pts) if rgb else pts).cuda()


def wall(ix, fn, k=5):
    fn(); ix.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    ix.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


def nn1_leg(name, kind, n, rgb):
    a, b = cloud(kind, n, synth.SEED_A, rgb), cloud(kind, n, synth.SEED_B, rgb)
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    d2 = torch.empty(n, dtype=torch.float32, device="cuda")
    ref = None
    for axes in AXES:
        for run in RUNS:
            ix = capi.Index(a, engine=capi.ENGINE_GRID, auto_sync=False)
            ix.set_option(capi.OPT_GRID_AXES, axes)
            ix.set_option(capi.OPT_XCD_RUN, run)

            def step():
                ix.set_input(a)
                ix.nn1(b, idx, d2)
            for _ in range(3):
                step()
            ix.enable_timing(2)
            for _ in range(10):
                step()
            ix.sync()
            tm = ix.timing()
            ix.enable_timing(0)
            ms = wall(ix, step, 20)
            qms = wall(ix, lambda: ix.nn1(b, idx, d2), 20)
            got = (idx.clone(), d2.clone())
            if ref is None:
                ref = got
            same = bool((got[0] == ref[0]).all() and (got[1].view(torch.int32) == ref[1].view(torch.int32)).all())
            print(f"{name:8s} axes {NAMES[axes]:10s} xcd_run {run:4d}: step {ms:7.3f} ms  query-only {qms:7.3f} ms  search kernels {tm[0]*1e3:7.1f} us  "
                  f"far {tm[1]*1e3:6.1f} us  build {tm[3]*1e3:6.1f} us  qsort {tm[4]*1e3:6.1f} us  same bits {same}", flush=True)
            ix.close()


def other_legs():
    n = 1_000_000
    a = cloud("both", n, synth.SEED_A)
    room = cloud("room", n, synth.SEED_A)
    obj5 = cloud("objects", 5_000_000, synth.SEED_A) if ("radius" in legs or "clusters" in legs) else None
    t2 = cloud("both", 2_000_000, synth.SEED_A) if "c4" in legs else None
    s2 = torch.from_numpy(synth.rigid_offset(synth.corridor_cloud(2_000_000, synth.SEED_B))).cuda() if "c4" in legs else None
    for axes in AXES:
        tag = f"axes {NAMES[axes]:10s}"
        if "c4" in legs:
            ix = capi.Index(t2, engine=capi.ENGINE_GRID, auto_sync=False)
            ix.set_option(capi.OPT_GRID_AXES, axes); ix.set_input(t2)
            ms = wall(ix, lambda: ix.icp_align(s2, max_iter=50, fixed=True), 3)
            print(f"c4 icp   {tag}: 50 passes + fitness {ms:7.2f} ms", flush=True)
            ix.close()
        if "knn" in legs:
            for nm, pts in (("corridor", a), ("room", room)):
                ix = capi.Index(pts, engine=capi.ENGINE_GRID, auto_sync=False)
                ix.set_option(capi.OPT_GRID_AXES, axes); ix.set_input(pts)
                for K in (51,):
                    ms = wall(ix, lambda: ix.knn(pts, K), 5)
                    print(f"knn      {tag}: {nm} 1M self K = {K}: {ms:7.3f} ms", flush=True)
                ms = wall(ix, lambda: ix.sor(mean_k=50, device="cuda"), 5)
                print(f"sor      {tag}: {nm} 1M: {ms:7.3f} ms", flush=True)
                ms = wall(ix, lambda: ix.normals(50, device="cuda"), 5)
                print(f"normals  {tag}: {nm} 1M K = 50: {ms:7.3f} ms", flush=True)
                ix.close()
        if "radius" in legs:
            ix = capi.Index(obj5, engine=capi.ENGINE_GRID, auto_sync=False)
            ix.set_option(capi.OPT_GRID_AXES, axes); ix.set_input(obj5)
            ms = wall(ix, lambda: ix.radius_count(obj5, 0.05), 3)
            print(f"radius   {tag}: 5M objects r = 0.05 count {ms:7.3f} ms", flush=True)
            ms = wall(ix, lambda: ix.radius_search(obj5, 0.05, sorted=True), 2)
            print(f"radius   {tag}: 5M objects r = 0.05 count + sorted fill {ms:7.3f} ms", flush=True)
            ix.close()
        if "clusters" in legs:
            ix = capi.Index(obj5, engine=capi.ENGINE_GRID, auto_sync=False)
            ix.set_option(capi.OPT_GRID_AXES, axes); ix.set_input(obj5)
            lab = torch.empty(5_000_000, dtype=torch.int32, device="cuda")
            ix.enable_timing(2)
            ms = wall(ix, lambda: ix.euclidean_clusters(0.05, 100, 250000, device_out=lab), 5)
            tm = ix.timing()
            print(f"clusters {tag}: 5M objects r = 0.05: call {ms:7.3f} ms  (union-find kernels {tm[0]:6.3f} ms)", flush=True)
            ix.close()


if "c3" in legs:
    nn1_leg("c3", "both", 10_000_000, True)
if "c2" in legs:
    nn1_leg("c2", "both", 1_000_000, False)
if "room" in legs:
    nn1_leg("room10M", "room", 10_000_000, False)
if set(legs) & {"c4", "knn", "radius", "clusters"}:
    other_legs()
