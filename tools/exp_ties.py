"""dev helper: what PCC_TIES_FLANN costs -- 1M-point mm-quantised room scan with 10 % duplicated points, 1M queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth


def run(n=1_000_000, out=print):
    a = np.round(synth.room_cloud(n, synth.SEED_A) * 1000) / 1000
    a[::10] = a[1::10][: len(a[::10])]
    a = torch.from_numpy(a.astype(np.float32)).cuda()
    q = torch.from_numpy((np.round(synth.room_cloud(n, synth.SEED_B) * 1000) / 1000).astype(np.float32)).cuda()
    idx = torch.empty(n, dtype=torch.int32, device="cuda"); d2 = torch.empty(n, dtype=torch.float32, device="cuda")
    ix = capi.Index(a, engine=capi.ENGINE_GRID, auto_sync=False)

    def timed(k=20):
        for _ in range(3): ix.nn1(q, idx, d2)
        ix.sync(); t0 = time.perf_counter()
        for _ in range(k): ix.nn1(q, idx, d2)
        ix.sync()
        return (time.perf_counter() - t0) / k * 1e3

    t_low = timed()
    ix.set_tie_order(capi.TIES_FLANN)
    t0 = time.perf_counter(); ix.nn1(q, idx, d2); ix.sync(); t_first = (time.perf_counter() - t0) * 1e3
    t_flann = timed()
    st = ix.stats()
    out(f"ties: {n} x {n} mm-quantised room scan, 10 % duplicates: pcc_nn1 lowest-index {t_low:.3f} ms, PCC_TIES_FLANN {t_flann:.3f} ms "
        f"({t_flann / t_low:.2f} x; {st[5]} queries flagged, {st[6]} indices changed); first FLANN call incl. the tree build {t_first:.1f} ms")
    ix.close()


if __name__ == "__main__":
    run(int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000)
