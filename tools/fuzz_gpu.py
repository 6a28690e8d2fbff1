"""Randomised parity campaign of the HIP path against the oracle (test infrastructure, like tests/): scene families
the unit tests only touch one at a time -- scales from 1e-4 to 1e4, large offsets (coordinates ~1e5 with a small
spread), planes, lines, duplicate piles, two clusters a long way apart, stray outliers, non-finite points, queries
inside / around / far from the cloud -- through nn1 (both engines), k-NN, radius search, first-within, clustering and
SOR.  Every comparison is bit-exact (d2 bits, lowest-index ties).

    python tools/fuzz_gpu.py --seconds 600 [--seed 1]

Prints one line per 50 cases and a summary; the first failing case of every operation is saved under gpurun_out/fuzz/.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402
from pointcloudcomparator_amd import capi  # noqa: E402


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def make_cloud(rng, n):
    fam = rng.integers(0, 9)
    scale = np.float32(10.0 ** rng.uniform(-4, 4)) if rng.random() < 0.3 else np.float32(rng.choice([0.5, 1.0, 5.0, 30.0]))
    if fam == 0:      # uniform box, anisotropic
        p = rng.random((n, 3), dtype=np.float32) * np.asarray(rng.choice([0.01, 0.3, 1.0], 3), np.float32)
    elif fam == 1:    # gaussian blobs
        c = rng.random((int(rng.integers(1, 8)), 3), dtype=np.float32)
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.02, (n, 3)).astype(np.float32)
    elif fam == 2:    # lattice with duplicates (exact ties)
        p = rng.integers(-8, 9, (n, 3)).astype(np.float32) * np.float32(0.125)
    elif fam == 3:    # plane
        p = rng.random((n, 3), dtype=np.float32)
        p[:, int(rng.integers(0, 3))] = np.float32(rng.random())
    elif fam == 4:    # line
        t = rng.random(n, dtype=np.float32)
        p = np.stack([t, t * np.float32(0.5), np.full(n, 0.25, np.float32)], 1)
        p = p[:, rng.permutation(3)]
    elif fam == 5:    # a pile of copies of few points
        base = rng.random((int(rng.integers(1, 6)), 3), dtype=np.float32)
        p = base[rng.integers(0, len(base), n)]
    elif fam == 6:    # two clusters a long way apart
        p = rng.random((n, 3), dtype=np.float32) * np.float32(0.2)
        p[rng.random(n) < 0.5] += np.float32(rng.choice([50.0, 1000.0]))
    elif fam == 7:    # mm-quantised scan
        p = np.round(rng.random((n, 3), dtype=np.float32) * 3000) / np.float32(1000)
        p = p.astype(np.float32)
    else:             # surface: sphere shell
        v = rng.normal(0, 1, (n, 3)).astype(np.float32)
        p = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-6).astype(np.float32)
    if rng.random() < 0.03:   # squared distances that overflow float
        scale = np.float32(10.0 ** rng.uniform(17, 19.5))
    p = (p * scale).astype(np.float32)
    if rng.random() < 0.25:   # large offset: few mantissa bits left for the spread
        p = (p + np.asarray(rng.choice([1e3, 1e5, -3e4], 3), np.float32)).astype(np.float32)
    if rng.random() < 0.3 and n > 20:   # stray outliers
        k = int(rng.integers(1, 4))
        p[rng.integers(0, n, k)] = (rng.choice([-1.0, 1.0], (k, 3)) * 10.0 ** rng.uniform(2, 6)).astype(np.float32)
    if rng.random() < 0.3:   # non-finite references
        k = int(rng.integers(1, max(2, n // 10)))
        idx = rng.integers(0, n, k)
        p[idx, rng.integers(0, 3, k)] = rng.choice([np.nan, np.inf, -np.inf], k).astype(np.float32)
    return np.ascontiguousarray(p)


def make_queries(rng, cloud, n):
    fin = cloud[np.isfinite(cloud).all(1)]
    if len(fin) == 0:
        fin = np.zeros((1, 3), np.float32)
    lo, hi = fin.min(0), fin.max(0)
    ext = np.maximum(hi - lo, np.float32(1e-6))
    mode = rng.integers(0, 5)
    if mode == 0:      # the cloud's own points
        q = fin[rng.integers(0, len(fin), n)].copy()
    elif mode == 1:    # jittered
        q = fin[rng.integers(0, len(fin), n)] + (rng.normal(0, 1, (n, 3)) * ext * 0.01).astype(np.float32)
    elif mode == 2:    # box around the cloud
        q = lo - ext * np.float32(0.5) + rng.random((n, 3), dtype=np.float32) * ext * np.float32(2.0)
    elif mode == 3:    # far away
        q = hi + ext * np.float32(rng.choice([3.0, 50.0, 1e4])) * rng.random((n, 3), dtype=np.float32)
    else:              # mixed with non-finite queries
        q = lo + rng.random((n, 3), dtype=np.float32) * ext
        k = max(1, n // 8)
        q[rng.integers(0, n, k), rng.integers(0, 3, k)] = np.nan
    return np.ascontiguousarray(q.astype(np.float32))


def scene_radius(rng, cloud):
    cloud = cloud[:, :3]
    fin = cloud[np.isfinite(cloud).all(1)]
    if len(fin) < 2:
        return 0.1
    ext = float(np.median(np.maximum(fin.max(0) - fin.min(0), 1e-9)))
    return float(np.float32(ext * 10.0 ** rng.uniform(-2.5, -0.3)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-refs", type=int, default=30000)
    ap.add_argument("--trace", default=None, help="append one line per case BEFORE it runs (to find a case that hangs)")
    args = ap.parse_args()
    out = ROOT / "gpurun_out" / "fuzz"
    out.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    n_cases = 0
    fails = {}
    counts = {}
    tolerated = {}

    def check(op, ok, **dump):
        counts[op] = counts.get(op, 0) + 1
        if not ok:
            if op not in fails:
                np.savez(out / f"fail_{op}_{n_cases}.npz", **dump)
                print(f"FAIL {op} case {n_cases}", flush=True)
            fails[op] = fails.get(op, 0) + 1

    while time.time() < t_end:
        n_cases += 1
        m = int(rng.choice([1, 2, 3, 14, 15, 16, 17, 63, 64, 65])) if rng.random() < 0.15 else int(10 ** rng.uniform(1, np.log10(args.max_refs)))
        nq = int(10 ** rng.uniform(0, 3.7))
        a = make_cloud(rng, m)
        q = make_queries(rng, a, nq)
        # point layouts of the boundary: 12-, 16- (PointXYZ), 32- (PointXYZRGB) and 128-byte (Histogram<32>) strides;
        # whatever sits beside x, y, z -- NaN included -- must not matter
        def widen(p):
            w = int(rng.choice([3, 3, 4, 8, 32]))
            if w == 3:
                return p
            o = rng.random((len(p), w), dtype=np.float32)
            if rng.random() < 0.5:
                o[:, 3:] = np.nan
            o[:, :3] = p
            return o
        a, q = widen(a), widen(q)
        n_valid = int(np.isfinite(a[:, :3]).all(1).sum())
        engine = capi.ENGINE_GRID if rng.random() < 0.8 else capi.ENGINE_BRUTE
        if args.trace:
            with open(args.trace, "a") as f:
                f.write(f"case {n_cases} m {m} nq {nq} valid {n_valid} engine {engine} t {time.time() - (t_end - args.seconds):.1f}\n")
        try:
            with capi.Index(a, engine=engine) as ix:
                # the forms of the pruned k = 1 kernel: one lane per query / rows drained flat (open lanes finished in
                # place or listed for a second kernel)
                ix.set_option(capi.OPT_NN1_KERNEL, int(rng.integers(0, 4)))
                ix.set_option(capi.OPT_NN1_OPEN_FLAT, int(rng.random() < 0.7))  # listed open lanes: drained flat / one lane each
                ix.set_option(capi.OPT_KNN_KERNEL, int(rng.random() < 0.8))
                ix.set_option(capi.OPT_EC_CELLS, int(rng.choice([3, 3, 4, 1, 2, 0])))
                if rng.random() < 0.2:
                    ix.set_option(capi.OPT_NN1_DENSE_MIN, int(rng.choice([1, 2, 1000000])))
                # the staging of a search that follows the build directly: behind it / beside it on the second stream (2: at every size)
                ix.set_option(capi.OPT_OVERLAP_PREP, int(rng.choice([0, 1, 2, 2])))
                # round 6: the grid's axis assignment drawn per scene (-1: by extent), the XCD run, the fused grid parameters; the
                # index is rebuilt under them (options that shape it act at the next set_input)
                ix.set_option(capi.OPT_GRID_AXES, int(rng.choice([-2, -2, -1, 0, 1, 2, 3, 4, 5])))
                ix.set_option(capi.OPT_XCD_RUN, int(rng.choice([1, 32, 256, 4096])))
                ix.set_option(capi.OPT_FUSE_PARAMS, int(rng.choice([0, 0, 1, 2, 3])))
                ix.set_option(capi.OPT_SCAN_CHAINED, int(rng.random() < 0.8))
                ix.set_option(capi.OPT_KNN_RUN, int(rng.choice([1, 2, 8, 8, 64])))
                ix.set_input(a)
                idx, d2 = ix.nn1(q)
                oi, od = oracle.nn1_exhaustive(a, q)
                check("nn1", (idx == oi).all() and (bits(d2) == bits(od)).all(), a=a, q=q, engine=engine)
                if rng.random() < 0.3:
                    # setInputCloud + search back to back on the used handle (the build still in flight when the staging starts)
                    ix.set_input(a)
                    i3, d3 = ix.nn1(q)
                    check("nn1_after_rebuild", (i3 == oi).all() and (bits(d3) == bits(od)).all(), a=a, q=q, engine=engine)
                op = rng.integers(0, 11)
                if args.trace:
                    with open(args.trace, "a") as f:
                        f.write(f"  nn1 done, op {op}\n")
                if op == 0 and n_valid >= 1:
                    # (one case in five with a large k: the selection kernel's 384- and 768-survivor forms, the merge kernel beyond)
                    kmax = 600 if rng.random() < 0.2 else 80
                    k = int(rng.integers(1, min(n_valid, kmax) + 1))
                    qs = q[:min(nq, 400 if kmax == 80 else 120)]
                    ki, kd = ix.knn(qs, k)
                    oki, okd = oracle.knn_exhaustive(a, qs, k)
                    check("knn", (ki == oki).all() and (bits(kd) == bits(okd)).all(), a=a, q=qs, k=k)
                elif op == 1:
                    r = scene_radius(rng, a)
                    qs = q[:min(nq, 300)]
                    cnt = ix.radius_count(qs, r)
                    ocnt = oracle.radius_count_exhaustive(a, qs, r)
                    ok = (cnt == ocnt).all()
                    if ok and int(ocnt.sum()) < 2_000_000:
                        offs, ri, rd = ix.radius_search(qs, r, sorted=True)
                        tree = oracle.KdTree(a)
                        for j in range(min(len(qs), 40)):
                            if not np.isfinite(qs[j, :3]).all():
                                continue
                            oi2, od2 = tree.radius(qs[j, :3], r)
                            if not ((ri[offs[j]:offs[j + 1]] == oi2).all() and (bits(rd[offs[j]:offs[j + 1]]) == bits(od2)).all()):
                                ok = False
                                break
                    check("radius", ok, a=a, q=qs, r=r)
                elif op == 2 and np.isfinite(q[:, :3]).all():
                    r = scene_radius(rng, a)
                    fw = ix.first_within(q, r)
                    ofw = oracle.first_within(a, q, r)
                    check("first_within", (fw == ofw).all(), a=a, q=q, r=r)
                elif op == 3 and m <= 12000:
                    tol = scene_radius(rng, a)
                    mn = int(rng.integers(1, 6))
                    labels, ncl, sizes = ix.euclidean_clusters(tol, mn, 100000)
                    ol, on, osz = oracle.euclidean_clusters(a, tol, mn, 100000)
                    check("clusters", ncl == on and (sizes == osz).all() and (labels == ol).all(), a=a, tol=tol, mn=mn)
                elif op == 4 and 60 <= n_valid and m <= 8000 and float(np.nanmax(np.abs(np.where(np.isfinite(a[:, :3]), a[:, :3], 0)))) < 1e15:
                    # (with overflowing distances a neighbour row comes back shorter than mean_k + 1, a case PCL's filter
                    # does not define -- it reads the row to its full length regardless)
                    mk = int(rng.integers(2, 51))
                    md, inl, thr, kept = ix.sor(mean_k=mk, stddev_mult=1.5)
                    omd, oinl, othr, okept = oracle.sor(a, mk, 1.5)
                    ok = (bits(md) == bits(omd)).all() and (np.asarray(inl) == oinl).all() and (thr == othr or (np.isnan(thr) and np.isnan(othr))) and kept == okept  # (a variance that rounds below zero: PCL's threshold is NaN, everything is kept)
                    if not ok:
                        # the oracle's SOR searches through its kd-tree as FLANN does, and FLANN's walk is not exact where
                        # squared distances are huge against their ulp (a stray point 1e5 away: the rounded branch bound prunes a
                        # subtree holding an equally near point).  The library's rows are the exhaustive ones: where the means
                        # differ, they must be the means of the exhaustive rows
                        bad = np.nonzero(bits(md) != bits(omd))[0]
                        if 0 < len(bad) <= 8:
                            _, xd = oracle.knn_exhaustive(a, a[bad], mk + 1)
                            xm = (np.sqrt(xd[:, 1:].astype(np.float64)).sum(1) / mk).astype(np.float32)
                            if (bits(xm) == bits(md[bad])).all():
                                counts["sor_flann_inexact"] = counts.get("sor_flann_inexact", 0) + 1
                                ok = True
                    check("sor", ok, a=a, mk=mk)
                elif op == 6 and 10 <= n_valid and m <= 15000 and n_valid == m:
                    # region growing on the library's own normals and neighbour rows (both checked elsewhere): labels
                    # must equal the oracle's sequential walk exactly
                    k = int(rng.integers(3, min(m, 60)))
                    nrm = ix.normals(min(k, 50))
                    if np.isfinite(nrm).all():
                        th = float(rng.choice([3.0, 6.0, 20.0])) / 180.0 * np.pi
                        ct = float(rng.choice([0.02, 0.1, 1.0]))
                        mn = int(rng.integers(1, 30))
                        labels, ncl = ix.region_growing(nrm, k=k, smoothness=th, curvature_threshold=ct, min_size=mn, max_size=1000000)
                        nbr, _ = ix.knn(a, k)
                        want, want_n = oracle.region_growing(nrm, nbr, th, ct, mn, 1000000)
                        check("region_growing", ncl == want_n and (labels == want).all(), a=a, k=k, th=th, ct=ct, mn=mn)
                elif op == 7 and m <= 20000 and n_valid == m and m >= 3:
                    thr = scene_radius(rng, a) * 0.2
                    opt = bool(rng.integers(0, 2))
                    mi = int(rng.choice([1, 20, 100]))
                    inl, coeff, its = ix.sac_plane(a, max_iterations=mi, threshold=thr, optimize=opt)
                    winl, wc, wits = oracle.sac_plane(a, max_iterations=mi, threshold=thr, optimize=opt)
                    # (coefficients: the same bits; where both sides are NaN -- coordinates whose products overflow -- the
                    # NaN's sign and payload are not part of the contract)
                    cg = np.asarray(coeff, np.float32)
                    same = (cg.view(np.uint32) == wc.view(np.uint32)) | (np.isnan(cg) & np.isnan(wc))
                    check("sac", its == wits and len(inl) == len(winl) and (np.asarray(inl) == winl).all() and same.all(),
                          a=a, thr=thr, opt=opt, mi=mi)
                elif op == 8 and m <= 20000 and n_valid >= 1:
                    leaf = scene_radius(rng, a) * 2.0
                    fin = a[:, :3][np.isfinite(a[:, :3]).all(1)]
                    if (np.ceil((fin.max(0) - fin.min(0)) / leaf) + 1).prod() < 5e6:
                        got = ix.voxel_grid(a, leaf)
                        vg, nv = oracle.voxel_grid(a, leaf)
                        # PCL adds a voxel's points in FLOAT (the oracle does too), the library in double: they differ by up to
                        # (points per voxel) x 2^-24 x |coordinate| (a pile of 12 757 copies at x = -3e4 is 2.8 apart)
                        tol = max(1e-5, 1.2e-7 * len(fin)) * max(1.0, float(np.abs(fin).max()))
                        ok = nv < 0 or (len(got) == nv and np.allclose(got[:, :3], vg[:, :3], rtol=0, atol=tol))
                        check("voxel", ok, a=a, leaf=leaf)
                elif op == 9 and n_valid >= 1 and np.isfinite(q[:, :3]).all() and max(float(np.nanmax(np.abs(q[:, :3]))), float(np.nanmax(np.abs(np.where(np.isfinite(a[:, :3]), a[:, :3], 0))))) < 1e15:
                    # (beyond ~1e18 FLANN's own branch bounds overflow and its tree walk stops being exact: the library's
                    # answer is then the true float minimum, FLANN's is not -- nothing to replay)
                    # PCC_TIES_FLANN: among equally near references the one FLANN's tree walk meets first -- the oracle's
                    # kd-tree restatement decides (lattices and piles of copies are full of ties)
                    rule = int(rng.integers(0, 3))  # which of FLANN's split rules shapes the tree (product and oracle alike)
                    ix.set_option(capi.OPT_FLANN_SPLIT, rule)
                    ix.set_tie_order(capi.TIES_FLANN)
                    fi, fd = ix.nn1(q)
                    ix.set_tie_order(capi.TIES_LOWEST_INDEX)
                    oracle.set_split_rule(rule)
                    try:
                        ti, td = oracle.KdTree(a).nn1_batch(q)
                    finally:
                        oracle.set_split_rule(0)
                    # (FLANN's walk is not always exact in float: its branch bounds round, and far from a tight cluster it
                    # can settle one ulp above the true minimum.  The library returns the minimum; the replay only decides
                    # among references AT the minimum, so compare indices where FLANN found it.)
                    xi, xd = oracle.nn1_exhaustive(a, q)
                    at_min = bits(td) == bits(xd)
                    check("ties_flann", (bits(fd) == bits(xd)).all() and (fi[at_min] == ti[at_min]).all(), a=a, q=q, engine=engine)
                elif op == 10 and 20 <= n_valid == m <= 15000 and float(np.abs(a[:, :3]).max()) < 1e6:
                    # normals on the library's own neighbour rows: bit-identical to the oracle, the three libm calls of the
                    # cubic's closed form included (csrc/libm_f32.hpp restates glibc's)
                    k = int(rng.integers(5, min(m, 51)))
                    nrm = ix.normals(k)
                    nbr, _ = ix.knn(a, k)
                    want = oracle.normals(a, k, neighbours=nbr)
                    fin_rows = np.isfinite(want).all(1) & np.isfinite(nrm).all(1)
                    ok = (np.isfinite(want).all(1) == np.isfinite(nrm).all(1)).all()
                    diff = (bits(nrm) != bits(want)).any(1) & fin_rows
                    ok = ok and not diff.any()
                    tolerated["normals_points_checked"] = tolerated.get("normals_points_checked", 0) + int(fin_rows.sum())
                    check("normals", ok, a=a, k=k)
                elif op == 5 and n_valid >= 3 and np.isfinite(q[:, :3]).all():
                    i2, dd, sums = ix.icp_step(q)
                    check("icp_step", (i2 == oi).all() and (bits(dd) == bits(od)).all(), a=a, q=q)
        except capi.PccError as e:
            # only the documented refusals are acceptable
            msg = str(e)
            check("status", ("empty" in msg) or ("no valid" in msg) or ("non-finite" in msg), a=a, q=q, msg=np.array(msg))
        if n_cases % 50 == 0:
            print(f"{n_cases} cases, {sum(counts.values())} checks, failures {fails}", flush=True)
    # the deviation class the harness tolerates, in the open (not a bit-identical result):
    #   sor_flann_inexact: SOR rows where the oracle's kd-tree walk (FLANN's) misses the float minimum and the library's row
    #                      was verified against the exhaustive search instead
    # (normals_points_checked: points whose normal and curvature carried the oracle's bits -- all of them, since round 5)
    tolerated["sor_flann_inexact"] = counts.get("sor_flann_inexact", 0)
    print(f"TOLERATED {tolerated}", flush=True)
    print(f"DONE {n_cases} cases; checks {counts}; failures {fails}", flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
