"""One pass over every non-k=1 operation at benchmark scale, for `rocprofv3 --kernel-trace --stats`
(summary committed as profiles/<round>_ops_kernel_stats.csv) and for wall-clock numbers in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth


def timed(name, fn, reps=2):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{name}: {best * 1e3:.2f} ms", flush=True)
    return r


n = 1_000_000
a = synth.corridor_cloud(n, synth.SEED_A)
ta = torch.from_numpy(a).cuda()
ix = capi.Index(ta)
timed("knn k=51 1M (device)", lambda: (ix.knn(ta, 51), ix.sync()))
timed("sor k=50 1M (device outputs)", lambda: ix.sor(50, 1.5, device="cuda:0"))
timed("sor k=50 1M (host outputs)", lambda: ix.sor(50, 1.5))
timed("radius_count r=0.05 1M", lambda: (ix.radius_count(ta, 0.05), ix.sync()))
nrm = timed("normals k=50 1M (device)", lambda: ix.normals(50, device="cuda:0"))
timed("region_growing k=100 1M (device normals)", lambda: ix.region_growing(nrm, k=100))
timed("voxel_grid 0.025 1M (host)", lambda: ix.voxel_grid(a, 0.025))
timed("sac_plane 1M (device)", lambda: ix.sac_plane(ta))
timed("first_within 100k queries", lambda: (ix.first_within(ta[:100000] + 0.01, 0.05), ix.sync()))
ix.close()
room = torch.from_numpy(synth.room_cloud(synth.ROOM_SIZES[1], synth.SEED_A)).cuda()
ix = capi.Index(room)
timed(f"room scan {synth.ROOM_SIZES[1]} points: sor k=50 (device outputs)", lambda: ix.sor(50, 1.5, device="cuda:0"))
timed(f"room scan {synth.ROOM_SIZES[1]} points: knn k=51 (device)", lambda: (ix.knn(room, 51), ix.sync()))
ix.close()
obj = synth.corridor_cloud(5_000_000, synth.SEED_A, layer="objects")
ix = capi.Index(torch.from_numpy(obj).cuda())
r = timed("euclidean_clusters 5M object points", lambda: ix.euclidean_clusters(0.05, 100, 250000))
print("  clusters:", r[1])
ix.close()
m = 2_000_000
tgt = torch.from_numpy(synth.corridor_cloud(m, synth.SEED_A)).cuda()
src = torch.from_numpy(synth.rigid_offset(synth.corridor_cloud(m, synth.SEED_B))).cuda()
ix = capi.Index(tgt)
timed("icp 2M x 2M, 50 fixed iterations", lambda: ix.icp_align(src, max_iter=50, fixed=True), reps=1)
ix.close()
