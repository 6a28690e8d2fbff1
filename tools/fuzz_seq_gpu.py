"""Randomised SEQUENCES on one long-lived handle (test infrastructure, companion of tools/fuzz_gpu.py): the index is
rebuilt over clouds of very different sizes in between searches, and every kind of call follows every other -- what the
per-scene campaign cannot see: scratch buffers that shrank or grew, the query order ICP keeps, its warm start, the far-query
heuristic that remembers earlier searches, the tie-order mode, host and device (torch) arguments in turn.  Every result is
compared with the oracle on the cloud the handle holds at that moment (bit-exact where tools/fuzz_gpu.py is).

    python tools/fuzz_seq_gpu.py --seconds 300 [--seed 1]
"""
import argparse
import importlib.util
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402
from pointcloudcomparator_amd import capi  # noqa: E402

_spec = importlib.util.spec_from_file_location("fuzz_gpu", ROOT / "tools" / "fuzz_gpu.py")
fz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fz)
bits = fz.bits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-refs", type=int, default=40000)
    ap.add_argument("--trace", default=None)
    ap.add_argument("--no-icp-align", action="store_true",
                    help="run whole ICP loops but do not compare them with the oracle's (the comparison is statistical: "
                         "two solvers on noisy data; the GPU suite's short leg wants none of that)")
    args = ap.parse_args()
    import torch
    out = ROOT / "gpurun_out" / "fuzz"
    out.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    fails, counts = {}, {}
    step = 0

    def check(op, ok, **dump):
        counts[op] = counts.get(op, 0) + 1
        if not ok:
            if op not in fails:
                np.savez(out / f"seqfail_{op}_{step}.npz", **dump)
                print(f"FAIL {op} step {step}", flush=True)
            fails[op] = fails.get(op, 0) + 1

    def trace(msg):
        if args.trace:
            with open(args.trace, "a") as f:
                f.write(msg + "\n")

    def fresh_cloud():
        m = int(rng.choice([1, 3, 16, 64, 65])) if rng.random() < 0.1 else int(10 ** rng.uniform(1, np.log10(args.max_refs)))
        a = fz.make_cloud(rng, m)
        if float(np.nanmax(np.abs(np.where(np.isfinite(a), a, 0)))) > 1e15:   # (the overflow family has its own leg)
            a = (a * np.float32(1e-12)).astype(np.float32)
        return np.ascontiguousarray(a)

    a = fresh_cloud()
    ix = capi.Index(a, engine=capi.ENGINE_GRID if rng.random() < 0.85 else capi.ENGINE_BRUTE)
    ties = capi.TIES_LOWEST_INDEX
    try:
        while time.time() < t_end:
            step += 1
            if rng.random() < 0.25:
                a = fresh_cloud()
                src = torch.from_numpy(a).cuda() if rng.random() < 0.4 else a
                trace(f"step {step} set_input m {len(a)} {'device' if src is not a else 'host'}")
                ix.set_input(src)
                if rng.random() < 0.2:
                    ties = capi.TIES_FLANN if ties == capi.TIES_LOWEST_INDEX else capi.TIES_LOWEST_INDEX
                    ix.set_tie_order(ties)
            n_valid = int(np.isfinite(a).all(1).sum())
            nq = int(10 ** rng.uniform(0, 3.6))
            q = fz.make_queries(rng, a, nq)
            if float(np.nanmax(np.abs(np.where(np.isfinite(q), q, 0)))) > 1e15:
                q = np.where(np.isfinite(q), np.clip(q, -1e12, 1e12), q).astype(np.float32)
            qd = torch.from_numpy(q).cuda() if rng.random() < 0.4 else q
            op = int(rng.integers(0, 8))
            trace(f"step {step} op {op} m {len(a)} valid {n_valid} nq {nq} ties {ties} q {'device' if qd is not q else 'host'}")
            try:
                if op in (0, 1):
                    idx, d2 = ix.nn1(qd)
                    idx, d2 = (idx.cpu().numpy(), d2.cpu().numpy()) if hasattr(idx, "cpu") else (idx, d2)
                    oi, od = oracle.nn1_exhaustive(a, q)
                    ok = (bits(d2) == bits(od)).all()
                    if ties == capi.TIES_FLANN and np.isfinite(q).all():
                        # index: FLANN's choice wherever its walk reaches the float minimum (see tools/fuzz_gpu.py)
                        ti, td = oracle.KdTree(a).nn1_batch(q)
                        at_min = bits(td) == bits(od)
                        ok = ok and (idx[at_min] == ti[at_min]).all()
                    elif ties != capi.TIES_FLANN:
                        ok = ok and (idx == oi).all()
                    check("nn1", ok, a=a, q=q, ties=ties)
                elif op == 2 and n_valid >= 1:
                    k = int(rng.integers(1, min(n_valid, 70) + 1))
                    qs = q[:300]
                    ki, kd = ix.knn(qs, k)
                    oki, okd = oracle.knn_exhaustive(a, qs, k)
                    check("knn", (ki == oki).all() and (bits(kd) == bits(okd)).all(), a=a, q=qs, k=k)
                elif op == 3:
                    r = fz.scene_radius(rng, a)
                    qs = q[:300]
                    cnt = ix.radius_count(qs, r)
                    check("radius_count", (cnt == oracle.radius_count_exhaustive(a, qs, r)).all(), a=a, q=qs, r=r)
                elif op == 4 and np.isfinite(q).all():
                    r = fz.scene_radius(rng, a)
                    check("first_within", (ix.first_within(q, r) == oracle.first_within(a, q, r)).all(), a=a, q=q, r=r)
                elif op == 5 and n_valid >= 3 and np.isfinite(q).all():
                    i2, dd, sums = ix.icp_step(q)
                    oi, od = oracle.nn1_exhaustive(a, q)
                    check("icp_step", (i2 == oi).all() and (bits(dd) == bits(od)).all(), a=a, q=q)
                elif op == 6 and n_valid >= 10 and np.isfinite(q).all() and nq >= 10 and n_valid == len(a):
                    # the whole loop (device-resident, warm-started, chunked criteria) against the oracle's host loop
                    it_max = int(rng.integers(1, 12))
                    fixed = bool(rng.integers(0, 2))
                    T, fit, it, conv = ix.icp_align(qd, max_iter=it_max, fixed=fixed)
                    # only where the FIRST pass determines a rotation: with correspondences on a line or a single point
                    # (random queries against a small cloud) every rotation about that line is optimal, Horn's and
                    # Umeyama's pick different ones and the loops part ways
                    oi1 = oracle.nn1_exhaustive(a, q)[0]
                    P, Q = q[:, :3].astype(np.float64), a[oi1][:, :3].astype(np.float64)
                    sv = np.linalg.svd((P - P.mean(0)).T @ (Q - Q.mean(0)), compute_uv=False)
                    # ... and where there is something to align: queries within a few per cent of the cloud's extent of
                    # their neighbours (far-off starts are chaotic -- the float Umeyama and the double Horn part ways)
                    ext = float(np.linalg.norm(a[:, :3].max(0) - a[:, :3].min(0)))
                    close = float(np.mean(np.sum((P - Q) ** 2, axis=1))) < (0.05 * ext) ** 2
                    if not args.no_icp_align and close and sv[0] > 0 and sv[1] > 1e-3 * sv[0] and sv[2] > 1e-6 * sv[0]:   # (full rank: the oracle's Umeyama restatement is not reliable on planar correspondences)
                        oT, ofit, oit, _, _ = oracle.icp(q, a, max_iter=it_max, fixed=fixed)
                        scale = max(1.0, float(np.abs(a).max()))
                        fit_ok = abs(fit - ofit) <= 2e-3 * max(abs(ofit), 1e-30) + 1e-10 * scale * scale
                        # (with criteria the stop hinges on two consecutive mean squared errors being EQUAL to 1e-12; at
                        # the quantisation floor of large coordinates the two solvers reach that a pass apart)
                        # float positions quantised to > 1e-4 of the extent of what is being aligned: both loops chase
                        # quantisation noise there
                        ext_q = float(np.linalg.norm(Q.max(0) - Q.min(0)))
                        coarse = float(np.abs(Q).max()) > 1e3 * max(ext_q, 1e-30) or float(np.abs(a[:, :3]).max()) > 1e3 * ext
                        ok = coarse or (it == oit and fit_ok) or (not fixed and abs(it - oit) <= 1 and abs(fit - ofit) <= 0.1 * max(abs(ofit), 1e-30))   # (coarse: the loops only chase quantisation noise)
                        # (with criteria, both loops at the SAME alignment -- transforms equal to 1e-6 of the scene, fitness to 1e-6 -- but
                        # stopping passes apart: the stop is two consecutive mean squared errors being equal to 1e-12, which at a
                        # fitness of 1e5 (a far outlier among the correspondences) is a coincidence of the last bits; seed 1111, step 20085)
                        same_place = np.allclose(T, oT, rtol=1e-5, atol=1e-6 * scale) and abs(fit - ofit) <= 1e-6 * max(abs(ofit), 1e-30)
                        ok = ok or (not fixed and same_place)
                        check("icp_align", ok, a=a, q=q, it_max=it_max, fixed=fixed, T=T, oT=oT, fit=fit, ofit=ofit, it=it, oit=oit)
                elif op == 7 and len(a) <= 12000:
                    tol = fz.scene_radius(rng, a)
                    mn = int(rng.integers(1, 6))
                    labels, ncl, sizes = ix.euclidean_clusters(tol, mn, 100000)
                    ol, on, osz = oracle.euclidean_clusters(a, tol, mn, 100000)
                    check("clusters", ncl == on and (sizes == osz).all() and (labels == ol).all(), a=a, tol=tol, mn=mn)
            except capi.PccError as e:
                msg = str(e)
                check("status", ("empty" in msg) or ("no valid" in msg) or ("non-finite" in msg) or ("correspond" in msg), a=a, q=q, msg=np.array(msg))
            if step % 100 == 0:
                print(f"{step} steps, {sum(counts.values())} checks, failures {fails}", flush=True)
    finally:
        ix.close()
    print(f"DONE {step} steps; checks {counts}; failures {fails}", flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
