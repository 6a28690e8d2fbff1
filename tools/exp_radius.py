"""dev helper: radius fill on the object layer (r = 0.05, rows of ~80 neighbours): wall time per call, sorted and unsorted.
usage: exp_radius.py [m] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudcomparator_amd import capi, synth
m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
r = float(os.environ.get("R", "0.05"))
obj = torch.from_numpy(synth.corridor_cloud(m, synth.SEED_A, layer="objects")).cuda()
ix = capi.Index(obj, auto_sync=False)
cnt = ix.radius_count(obj, r); ix.sync()
total = int(cnt.to(torch.int64).sum().item())
offs = torch.zeros(m + 1, dtype=torch.int64, device="cuda")
offs[1:] = torch.cumsum(cnt.to(torch.int64), 0)
idx = torch.empty(total, dtype=torch.int32, device="cuda"); d2 = torch.empty(total, dtype=torch.float32, device="cuda")
ptr, nn_, stride, mem = capi._points(obj)
for srt in [int(x) for x in os.environ.get("SORTED", "1,0").split(",")]:
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        capi._check(capi.LIB.pcc_radius_fill(ix._h, ptr, nn_, stride, mem, r, srt, offs.data_ptr(), idx.data_ptr(), d2.data_ptr()))
        ix.sync()
        best = min(best, time.perf_counter() - t0)
    print(f"radius fill m={m} r={r} rows {total / m:.1f} sorted={srt}: {best * 1e3:.3f} ms "
          f"({(40.0 * m + 8.0 * total) / best / 8e12 * 100:.1f} % of 8 TB/s); cells {ix.stats()[3]}", flush=True)
best = 1e9
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix.radius_count(obj, r); ix.sync()
    best = min(best, time.perf_counter() - t0)
print(f"radius count: {best * 1e3:.3f} ms", flush=True)
