"""GPU parity (through the C-ABI) of k-NN, radius search, clustering, SOR, ICP blocks and
descriptor matching against the oracle's FLANN/PCL restatement."""
import numpy as np
import pytest

import oracle
from pointcloudcomparator_amd import capi, synth

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def _scene(n, seed=synth.SEED_A):
    return synth.corridor_cloud(n, seed)


@pytest.mark.parametrize("k", [1, 5, 51, 128, 129, 200, 256, 257, 400, 512, 513])
def test_knn_matches_exhaustive(gpu, k):
    a = _scene(30000)
    b = _scene(2000, synth.SEED_B)
    b[3, 0] = np.nan
    b[1500:1600] += np.float32(40.0)  # far outside: forces cube growth / whole-grid scan
    with capi.Index(a) as ix:
        idx, d2 = ix.knn(b, k)
    oi, od = oracle.knn_exhaustive(a, b, k)
    assert (idx[3] == -1).all() and np.isinf(d2[3]).all()
    assert (_bits(d2) == _bits(od)).all()
    assert (idx == oi).all()


def test_knn_k_larger_than_cloud(gpu):
    a = _scene(20)
    b = _scene(50, synth.SEED_B)
    with capi.Index(a) as ix:
        idx, d2 = ix.knn(b, 51)
    oi, od = oracle.knn_exhaustive(a, b, 51)
    assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()
    assert (idx[:, 20:] == -1).all()


def test_knn_duplicates_order(gpu):
    rng = np.random.default_rng(11)
    base = rng.random((1500, 3), dtype=np.float32)
    a = np.concatenate([base, base, base[:700]])
    with capi.Index(a) as ix:
        idx, d2 = ix.knn(base[:400], 8)
    oi, od = oracle.knn_exhaustive(a, base[:400], 8)
    assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()


@pytest.mark.parametrize("k", [1, 7, 51, 64, 65, 100, 128, 129, 200, 256, 257, 400, 512])
def test_knn_selection_kernel_and_merge_kernel_agree_with_the_oracle(gpu, k):
    """k <= 512 runs the bucket-selection kernel (PCC_OPT_KNN_KERNEL = 1, default) and hands queries it cannot take to
    the merge network; option 0 is the network alone.  Both must give the exhaustive oracle's rows bit for bit on a scene
    that exercises the hand-backs: a lattice (hundreds of equal distances: crowded buckets, survivors beyond the sort),
    dense blobs (cubes beyond the candidate buffer), far and non-finite queries, and a device-resident call."""
    import torch
    rng = np.random.default_rng(5)
    lattice = np.stack(np.meshgrid(*[np.arange(12, dtype=np.float32) * np.float32(0.05)] * 3), -1).reshape(-1, 3)
    blob = (rng.normal(0, 0.004, (6000, 3)) + np.array([2.0, 0.3, 0.3])).astype(np.float32)
    a = np.ascontiguousarray(np.concatenate([_scene(20000), lattice + np.float32(1.0), blob])[rng.permutation(27728)])
    q = np.concatenate([_scene(1500, synth.SEED_B), lattice[::3] + np.float32(1.0), blob[:300] + np.float32(0.001),
                        _scene(50, synth.SEED_B) + np.float32(60.0)]).astype(np.float32)
    q[11, 1] = np.nan
    q = np.ascontiguousarray(q)
    oi, od = oracle.knn_exhaustive(a, q, k)
    with capi.Index(a) as ix:
        for form in (1, 0):
            ix.set_option(capi.OPT_KNN_KERNEL, form)
            idx, d2 = ix.knn(q, k)
            assert (idx == oi).all() and (_bits(d2) == _bits(od)).all(), form
            assert (idx[11] == -1).all() and np.isinf(d2[11]).all()
        ix.set_option(capi.OPT_KNN_KERNEL, 1)
        di, dd = ix.knn(torch.from_numpy(q).cuda(), k)
        assert (di.cpu().numpy() == oi).all() and (_bits(dd.cpu().numpy()) == _bits(od)).all()
        with pytest.raises(capi.PccError):
            ix.set_option(capi.OPT_KNN_KERNEL, 2)


@pytest.mark.parametrize("radius", [0.05, 0.2, 0.7])  # (0.7: balls of hundreds of candidates -- the count pass takes its wave form)
def test_radius_count_and_fill(gpu, radius):
    a = _scene(40000)
    b = _scene(3000, synth.SEED_B)
    b[7, 2] = np.inf
    with capi.Index(a) as ix:
        cnt = ix.radius_count(b, radius)
        offs, idx, d2 = ix.radius_search(b, radius, sorted=True)
    oc = oracle.radius_count_exhaustive(a, b, radius)
    assert (cnt == oc).all() and cnt[7] == 0
    kd = oracle.KdTree(a)
    for i in list(range(0, 3000, 37)) + [7]:
        ri, rd = kd.radius(b[i], radius, sorted=True) if np.isfinite(b[i]).all() else (np.empty(0, np.int32), np.empty(0, np.float32))
        s, e = offs[i], offs[i + 1]
        assert e - s == len(ri)
        assert (idx[s:e] == ri).all() and (_bits(d2[s:e]) == _bits(rd)).all()


def test_radius_boundary_is_strict(gpu):
    # points exactly at d2 == r2 are excluded (RadiusResultSet: dist < radius)
    a = np.array([[0, 0, 0], [0.5, 0, 0], [0.25, 0, 0], [0, 0.5, 0]], np.float32)
    q = np.zeros((1, 3), np.float32)
    with capi.Index(a, engine=capi.ENGINE_BRUTE) as ix:
        cnt = ix.radius_count(q, 0.5)
    assert cnt[0] == 2


@pytest.mark.parametrize("form", [3, 4, 1, 2, 0])
def test_euclidean_clusters_known_partition(gpu, form):
    """object layer only: the balls are the clusters by construction.  PCC_OPT_EC_CELLS: 3 = union-find over the cells
    of the clique-cell grid, runs of touching cells along a row joined by stores (default), 4 = one union per cell and face
    (round 5), 1 = one parent per point with the lanes over a cell's neighbour cells, 2 = the same with
    the lanes over points, 0 = per-point ball scan"""
    pts = synth.corridor_cloud(120000, synth.SEED_A, layer="objects")
    with capi.Index(pts) as ix:
        ix.set_option(capi.OPT_EC_CELLS, form)
        labels, ncl, sizes = ix.euclidean_clusters(0.05, 100, 250000)
    ol, on, osz = oracle.euclidean_clusters(pts, 0.05, 100, 250000)
    assert ncl == on
    assert (sizes == osz).all()
    assert (labels == ol).all()


def test_euclidean_clusters_filters_and_nonfinite(gpu):
    rng = np.random.default_rng(5)
    blobs = [rng.normal(c, 0.02, (n, 3)).astype(np.float32)
             for c, n in [((0, 0, 0), 400), ((1, 0, 0), 90), ((0, 1, 0), 250), ((1, 1, 1), 3000), ((3, 3, 3), 150)]]
    pts = np.concatenate(blobs)
    rng.shuffle(pts)
    pts[10] = np.nan
    ol, on, osz = oracle.euclidean_clusters(pts, 0.05, 100, 2500)
    with capi.Index(pts) as ix:
        labels, ncl, sizes = ix.euclidean_clusters(0.05, 100, 2500)
    assert ncl == on and (sizes == osz).all() and (labels == ol).all()
    assert labels[10] == -1


def test_kept_self_knn_rows_serve_normals_and_region_growing(gpu):
    """PCC_OPT_KNN_CACHE_K: the self k-NN rows are searched once with K neighbours and kept; normals (50) takes their
    prefix, region growing (100) the rows themselves; a larger request searches again; a new cloud drops them.  Every
    result must equal the uncached one bit for bit."""
    pts = _scene(40000)
    other = _scene(30000, synth.SEED_B)
    with capi.Index(pts) as ix:
        n0 = ix.normals(50)
        l0, c0 = ix.region_growing(n0, k=100)
        l1, c1 = ix.region_growing(n0, k=120)
        ix.set_option(capi.OPT_KNN_CACHE_K, 100)
        n2 = ix.normals(50)                       # searches 100, takes the prefix
        l2, c2 = ix.region_growing(n0, k=100)     # the kept rows as they are
        n3 = ix.normals(30)                       # prefix again
        l3, c3 = ix.region_growing(n0, k=120)     # more than kept: searched again (and kept)
        n4 = ix.normals(50)
        assert (_bits(n2) == _bits(n0)).all() and (_bits(n4) == _bits(n0)).all()
        assert c2 == c0 and (l2 == l0).all() and c3 == c1 and (l3 == l1).all()
        ix.set_option(capi.OPT_KNN_CACHE_K, 0)
        assert (_bits(ix.normals(30)) == _bits(n3)).all()
        ix.set_option(capi.OPT_KNN_CACHE_K, 64)
        ix.set_input(other)                       # a new cloud: nothing kept may survive
        na = ix.normals(50)
    with capi.Index(other) as ix:
        assert (_bits(ix.normals(50)) == _bits(na)).all()


def test_sor_matches_oracle(gpu):
    pts = _scene(30000)
    pts[100] = np.nan
    md, inl, thr, kept = None, None, None, None
    with capi.Index(pts) as ix:
        md, inl, thr, kept = ix.sor(50, 1.5)
    omd, oinl, othr, okept = oracle.sor(pts, 50, 1.5)
    assert (_bits(md) == _bits(omd)).all()
    assert thr == othr
    assert (inl == oinl).all() and kept == okept


def test_sor_statistics_on_the_device_and_the_in_order_fallback(gpu):
    """sum, sq_sum, threshold and mask are taken on the device whenever no addition of PCL's in-order double sums can round
    (then every order gives the same bits); a cloud whose mean distances span micrometres to decimetres makes additions
    round, and the library must notice and take the sums in PCL's order on the host.  Both ways: the oracle's bits."""
    import torch
    pts = _scene(40000)
    with capi.Index(torch.from_numpy(pts).cuda()) as ix:
        md, inl, thr, kept = ix.sor(50, 1.5, device="cuda:0")
        assert ix.sor_on_device()
        omd, oinl, othr, okept = oracle.sor(pts, 50, 1.5)
        assert (_bits(md.cpu().numpy()) == _bits(omd)).all() and thr == othr
        assert (inl.cpu().numpy() == oinl).all() and kept == okept
    rng = np.random.default_rng(5)
    wide = _scene(60000)
    seeds = wide[:300]
    clumps = (seeds[:, None, :] + rng.normal(0, 2e-7, (300, 70, 3))).reshape(-1, 3).astype(np.float32)  # 70 points within a micrometre
    wide = np.concatenate([wide[300:], clumps]).astype(np.float32)
    with capi.Index(wide) as ix:
        md, inl, thr, kept = ix.sor(50, 1.5)
        on_dev = ix.sor_on_device()
    omd, oinl, othr, okept = oracle.sor(wide, 50, 1.5)
    assert not on_dev, "terms 2^-20 below a sum of 2^11: additions round, the in-order loop is required"
    assert (_bits(md) == _bits(omd)).all() and thr == othr and (inl == oinl).all() and kept == okept


@pytest.mark.parametrize("max_nn", [1, 7, 60, 600])
def test_radius_search_with_max_nn_keeps_the_nearest(gpu, max_nn):
    """KdTreeFLANN::radiusSearch(..., max_nn): FLANN's KNNRadiusResultSet keeps the max_nn nearest within the radius,
    ascending; counts are min(count, max_nn).  Against the oracle's full sorted rows cut at max_nn."""
    import torch
    rng = np.random.default_rng(2)
    a = np.concatenate([rng.normal(0, 0.12, (15000, 3)), rng.random((5000, 3)) * 2 - 1]).astype(np.float32)  # a blob: rows of up to ~1500
    q = np.concatenate([a[:400] + np.float32(0.003), np.array([[50, 50, 50], [np.nan, 0, 0]], np.float32)])
    r = 0.12
    tree = oracle.KdTree(a)
    with capi.Index(a) as ix:
        for queries in (q, torch.from_numpy(q).cuda()):
            offs, idx, d2 = ix.radius_search(queries, r, sorted=False, max_nn=max_nn)
            if not isinstance(offs, np.ndarray):
                offs, idx, d2 = offs.cpu().numpy(), idx.cpu().numpy(), d2.cpu().numpy()
            full = ix.radius_count(q, r)
            assert (np.diff(offs) == np.minimum(full, max_nn)).all() and (full.max() > max_nn or max_nn >= 600)
            for j in range(len(q) - 1):
                oi, od = tree.radius(q[j], r)
                assert (idx[offs[j]:offs[j + 1]] == oi[:max_nn]).all() and (_bits(d2[offs[j]:offs[j + 1]]) == _bits(od[:max_nn])).all()
            assert offs[-1] == offs[-2]            # the non-finite query finds nothing
        # max_nn at or beyond the cloud's size means "all"
        o1, i1, e1 = ix.radius_search(q[:50], r, sorted=True, max_nn=len(a))
        o2, i2, e2 = ix.radius_search(q[:50], r, sorted=True)
        assert (o1 == o2).all() and (i1 == i2).all() and (_bits(e1) == _bits(e2)).all()


@pytest.mark.parametrize("nq,max_nn", [(1, 5), (300, 1), (1, 1), (2000, 3)])
def test_radius_fill_max_with_offsets_computed_elsewhere_on_a_fresh_handle(gpu, nq, max_nn):
    """pcc_radius_fill_max WITHOUT the count call before it (offsets from the oracle), on a handle that has never sorted a
    query cloud: the k-NN rows must sit in buffers the query sort inside the search does not re-reserve (they were in the
    sort's pair scratch -- freed by its growing reserve with nq = 1, max_nn = 1 or a fresh handle).  Also the count / fill
    pair agrees on what "all" means: max_nn against the FINITE points, as PCL's total_nr_points_."""
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.normal(0, 0.1, (6000, 3)), rng.random((3000, 3)) - 0.5]).astype(np.float32)
    q = (a[rng.integers(0, len(a), nq)] + np.float32(0.002)).astype(np.float32)
    r = 0.08
    tree = oracle.KdTree(a)
    rows = [tree.radius(qq, r) for qq in q]
    cnt = np.array([min(len(ri), max_nn) for ri, _ in rows], np.int64)
    offs = np.zeros(nq + 1, np.int64)
    np.cumsum(cnt, out=offs[1:])
    with capi.Index(a) as ix:                                   # fresh handle: the fill is its first search
        idx, d2 = ix.radius_fill(q, r, offs, sorted=True, max_nn=max_nn)
    for j, (oi, od) in enumerate(rows):
        assert (idx[offs[j]:offs[j + 1]] == oi[:max_nn]).all() and (_bits(d2[offs[j]:offs[j + 1]]) == _bits(od[:max_nn])).all()
    # "all" is decided against the finite points: 40 points of which 10 are NaN, max_nn = 30 -> the plain radius rows
    b = a[:40].copy()
    b[::4] = np.nan
    with capi.Index(b, engine=capi.ENGINE_GRID) as ix:
        assert ix.stats()[2] == 30
        o1, i1, e1 = ix.radius_search(b[1:9], 0.5, sorted=True, max_nn=30)
        o2, i2, e2 = ix.radius_search(b[1:9], 0.5, sorted=True)
        assert (o1 == o2).all() and (i1 == i2).all() and (_bits(e1) == _bits(e2)).all()


def test_sor_threshold_nan_when_the_variance_rounds_below_zero(gpu):
    """tests/golden/sor_negative_variance.npz (a scene tools/fuzz_gpu.py drew: two tight clumps 7 km apart at coordinates of
    1e5, mean_k beyond a clump's size): every mean distance is ~2824 and (sq_sum - sum^2 / n) / (n - 1) rounds to a small
    NEGATIVE number -- PCL's threshold is the sqrt of it, NaN, and `distance > NaN` is false for every point: all are kept.
    The library says the same on both output routes (the fuzz checker had compared the two NaNs with ==)."""
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "sor_negative_variance.npz")
    a, mk = g["cloud"], int(g["mean_k"])
    omd, oinl, othr, okept = oracle.sor(a, mk, 1.5)
    assert np.isnan(othr) and okept == len(a) == int(g["kept"]) and (_bits(omd) == g["mean_dist_bits"]).all()
    with capi.Index(a) as ix:
        for dev in (None, "cuda:0"):
            md, inl, thr, kept = ix.sor(mk, 1.5, device=dev)
            md = md if dev is None else md.cpu().numpy()
            inl = inl if dev is None else inl.cpu().numpy()
            assert np.isnan(thr) and kept == okept and (inl == oinl).all() and (_bits(md) == _bits(omd)).all()


def test_transform_bits(gpu):
    pts = _scene(5000)
    a = np.deg2rad(7.0)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    T[:3, 3] = [0.3, -0.2, 0.1]
    with capi.Index(pts[:10]) as ix:
        out = ix.transform(T, pts)
    ref = oracle.transform(T, pts)
    assert (_bits(out) == _bits(ref)).all()


def test_icp_step_sums(gpu):
    tgt = _scene(40000)
    src = synth.rigid_offset(_scene(8000, synth.SEED_B))
    with capi.Index(tgt) as ix:
        idx, d2, sums = ix.icp_step(src)
    kd = oracle.KdTree(tgt)
    oi, od, osums = kd.icp_step_sums(tgt, src)
    assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()
    assert np.allclose(sums, osums, rtol=1e-12, atol=1e-9)
    rc, T = oracle.umeyama_from_sums(osums)
    assert rc == 0


def test_icp_align_recovers_offset(gpu):
    base = _scene(30000)
    tgt = base
    src = synth.rigid_offset(base[:12000], jitter=0.0005)
    with capi.Index(tgt) as ix:
        T, fit, it, conv = ix.icp_align(src, max_iter=20)
    oT, ofit, oit, ocorr, omse = oracle.icp(src, tgt, max_iter=20)
    assert conv and it == oit
    assert np.allclose(T, oT, atol=2e-5)
    assert abs(fit - ofit) <= 1e-6 * max(1.0, abs(ofit)) + 1e-9
    # the recovered transform undoes the synthetic offset: fitness near the jitter level
    assert fit < 1e-5


@pytest.mark.parametrize("m", [513, 1024, 1025, 3000, 70000])
def test_sorted_rows_longer_than_the_register_sort(gpu, m):
    """rows of more than 512 neighbours are sorted by a bitonic network over global memory (any length, +inf padding
    implied): a radius that swallows the whole cloud, keys with many equal distances (lattice) -- against the oracle's
    kd-tree rows, which are sorted by (d2, index)"""
    rng = np.random.default_rng(m)
    a = (rng.integers(-20, 21, (m, 3)) * 0.01).astype(np.float32)      # lattice: plenty of exact ties
    a[: m // 2] += rng.random((m // 2, 3), dtype=np.float32) * np.float32(0.003)
    q = np.concatenate([a[:3], np.array([[0.05, -0.02, 0.01]], np.float32)])
    r = 1.0
    with capi.Index(a) as ix:
        offs, idx, d2 = ix.radius_search(q, r, sorted=True)
    assert (np.diff(offs) == m).all()
    tree = oracle.KdTree(a)
    for j in range(len(q)):
        oi, od = tree.radius(q[j], r)
        assert (idx[offs[j]:offs[j + 1]] == oi).all() and (_bits(d2[offs[j]:offs[j + 1]]) == _bits(od)).all()


def test_icp_align_criteria_run_on_the_device(gpu):
    """the loop, its running transform and DefaultConvergenceCriteria live on the device (k_icp_solve), passes are enqueued
    in chunks of 5: iteration count, verdict and transform follow the oracle's host loop -- stop on the first chunk's
    second pass (identical clouds), in a later chunk (slow convergence), at the cap, and on too few correspondences"""
    base = _scene(30000)
    with capi.Index(base) as ix:
        # identical clouds: mse is 0 from the first pass, |mse - previous| < 1e-12 at the second
        T, fit, it, conv = ix.icp_align(base[:9000], max_iter=20)
        oT, ofit, oit, _, _ = oracle.icp(base[:9000], base, max_iter=20)
        assert conv and it == oit == 2 and fit == 0.0
        assert np.allclose(T, np.eye(4), atol=1e-7)
        # mm-quantised clouds settle on fixed correspondences: the criteria fire somewhere past the first chunk
        q = (np.round(base * 1000) / 1000).astype(np.float32)
        with capi.Index(q) as iq:
            src = synth.rigid_offset(q[:6000], jitter=0.0)
            T, fit, it, conv = iq.icp_align(src, max_iter=100)
            oT, ofit, oit, _, _ = oracle.icp(src, q, max_iter=100)
            assert conv and it == oit and 2 < it < 100, (it, oit)
            assert np.allclose(T, oT, atol=2e-5)
            # the cap: same clouds, fewer iterations than they need
            cap = max(3, it - 2)
            T2, _, it2, conv2 = iq.icp_align(src, max_iter=cap)
            oT2, _, oit2, _, _ = oracle.icp(src, q, max_iter=cap)
            assert conv2 and it2 == oit2 == cap
            assert np.allclose(T2, oT2, atol=2e-5)
        # two source points: min_number_correspondences_ = 3 is never met
        T, fit, it, conv = ix.icp_align(base[:2], max_iter=10)
        assert not conv and it == 0 and np.array_equal(T, np.eye(4, dtype=np.float32))


@pytest.mark.parametrize("fixed", [False, True])
def test_icp_align_options_do_not_change_a_bit(gpu, fixed):
    """PCC_OPT_ICP_WARM = 0 (every pass searches from scratch) and PCC_OPT_ICP_DEVICE_LOOP = 0 (the host drives the loop,
    reduces the sums and solves the transform) are other routes to the same numbers: transform, fitness, iteration count
    and verdict carry identical bits, with criteria active and with a fixed count.  A source that starts
    well off the target exercises the far walk and the warm start's empty-cube branch."""
    base = _scene(60000)
    src = synth.rigid_offset(base[:20000], jitter=0.002)
    src[:500] += np.float32([0.8, -0.5, 0.3])  # stragglers beyond the cell walk
    runs = {}
    with capi.Index(base, engine=capi.ENGINE_GRID) as ix:
        for warm in (1, 0):
            for loop in (1, 0):
                ix.set_option(capi.OPT_ICP_WARM, warm)
                ix.set_option(capi.OPT_ICP_DEVICE_LOOP, loop)
                T, fit, it, conv = ix.icp_align(src, max_iter=12, fixed=fixed)
                runs[(warm, loop)] = (T.view(np.uint32).copy(), np.float64(fit).view(np.uint64), it, conv)
    first = runs[(1, 1)]
    assert first[2] > 1
    for key, r in runs.items():
        assert (r[0] == first[0]).all() and r[1] == first[1] and r[2] == first[2] and r[3] == first[3], key
    # and the loop as such: the oracle's host loop from the same start, compared as floats (Horn vs Jacobi SVD)
    oT, ofit, oit, _, _ = oracle.icp(src, base, max_iter=12, fixed=fixed)
    assert first[2] == oit
    assert np.allclose(first[0].view(np.float32), oT, atol=5e-5)


def test_icp_align_far_from_the_origin(gpu):
    """a small cloud at geo-referenced coordinates (8 cm across at (1e3, 1e5, 1e5)): the rotation comes from sum q p^T
    - n pm qm^T, 4e11 against 0.3 -- summed about the origin it lost five digits and an exact copy of the cloud came out
    7 mm off (tools/fuzz_seq_gpu.py found it).  The loop sums about the centre of the target's bounding box."""
    rng = np.random.default_rng(5)
    base = (np.array([1000.0, 100000.0, 100000.0]) + rng.random((43, 3)) * 0.08).astype(np.float32)
    with capi.Index(base) as ix:
        T, fit, it, conv = ix.icp_align(base, max_iter=6)
        assert conv and it == 2 and fit == 0.0
        assert np.allclose(T[:3, :3], np.eye(3), atol=1e-9) and np.abs(T[:3, 3]).max() < 1e-4
        # and a real offset on a larger cloud out there: against the oracle's loop
        big = (np.array([1000.0, 100000.0, 100000.0]) + rng.random((4000, 3)) * 2.0).astype(np.float32)
        ix.set_input(big)
        c, s_ = np.cos(0.02), np.sin(0.02)
        R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
        ctr = big.mean(0).astype(np.float64)
        src = ((big[:1500].astype(np.float64) - ctr) @ R.T + ctr + np.array([0.01, -0.02, 0.015])).astype(np.float32)
        T, fit, it, conv = ix.icp_align(src, max_iter=15)
        oT, ofit, oit, _, _ = oracle.icp(src, big, max_iter=15)
        assert it == oit
        truth = big[:1500].astype(np.float64)       # the source is a moved copy of these
        moved = (src.astype(np.float64) @ T[:3, :3].astype(np.float64).T + T[:3, 3].astype(np.float64))
        omoved = (src.astype(np.float64) @ oT[:3, :3].astype(np.float64).T + oT[:3, 3].astype(np.float64))
        err, oerr = np.abs(moved - truth).max(), np.abs(omoved - truth).max()
        # float positions out there resolve 0.008: both loops end ~0.015 from the truth, in their own directions
        assert err < 0.03 and err <= oerr + 0.005
        assert fit < 2 * ofit + 1e-6


def test_icp_step_about_a_centre(gpu):
    """pcc_icp_step_about / pcc_rigid_from_sums_about: the stepwise (sharded-loop) route with the sums taken about a
    point of the cloud reproduces pcc_icp_align's first pass bit for bit at geo-referenced coordinates"""
    rng = np.random.default_rng(6)
    big = (np.array([1000.0, 100000.0, 100000.0]) + rng.random((3000, 3)) * 1.0).astype(np.float32)
    src = (big[:1000] + np.float32(0.02)).astype(np.float32)
    with capi.Index(big) as ix:
        T1, _, it, _ = ix.icp_align(src, max_iter=1, fixed=True)
        idx, d2, sums = ix.icp_step(src, center=src[0].astype(np.float64))
        oi, od = oracle.nn1_exhaustive(big, src)
        assert (idx == oi).all() and (_bits(d2) == _bits(od)).all() and sums[16] == 1000
        Ts = capi.rigid_from_sums(sums, center=src[0].astype(np.float64))
    assert it == 1 and (np.asarray(T1, np.float32).view(np.uint32) == Ts.view(np.uint32)).all()


def test_match_knn_mirrors_reference_quirks(gpu):
    rng = np.random.default_rng(2)
    des1 = np.zeros((600, 32), np.float32)     # RIFT32 = pcl::Histogram<32>, 128-byte stride
    des2 = np.zeros((450, 32), np.float32)
    des1[:, :] = rng.random((600, 32), dtype=np.float32)
    des2[:, :] = rng.random((450, 32), dtype=np.float32)
    with capi.Index(des1) as ix:
        out = ix.match_knn(des2, 0.05)
    ref = oracle.match_rift_knn(des1, des2)
    assert out[0] == 0 and len(out) == len(ref)  # size() == matches + 1 (reference :568)
    assert (out == ref).all()


def test_first_within_matches_linear_scan(gpu):
    """processRIFTwithSIFT's keypoint snap (reference src/comparator.cpp:696-713)"""
    cloud = _scene(50000)
    rng = np.random.default_rng(9)
    keys = cloud[rng.integers(0, len(cloud), 700)] + rng.normal(0, 0.02, (700, 3)).astype(np.float32)
    keys[::50] += np.float32(3.0)      # no point within 0.05
    keys[7, 1] = np.nan
    with capi.Index(cloud) as ix:
        idx = ix.first_within(keys, 0.05)
    ref = oracle.first_within(cloud, keys, 0.05)
    assert (idx == ref).all()
    assert idx[7] == -1 and (idx[::50][1:] == -1).all() and (idx >= 0).sum() > 500


@pytest.mark.parametrize("leaf", [0.025, 0.1])
def test_voxel_grid_matches_pcl_restatement(gpu, leaf):
    """pcl::VoxelGrid (reference src/segmentation.cpp:69-74): same voxels, same order, same colours;
    coordinates within float rounding of the float-accumulated centroid"""
    pts = synth.with_rgb_stride(synth.corridor_cloud(200000, synth.SEED_A, layer="objects"))
    pts[17, 0] = np.nan
    with capi.Index(pts[:64]) as ctx:
        out = ctx.voxel_grid(pts, leaf, has_rgb=True)
    ref, nv = oracle.voxel_grid(pts, leaf, has_rgb=True)
    assert len(out) == nv and nv < len(pts)
    # PCL sums a voxel's points in float (error ~ count * 6e-8 * |coordinate|, larger voxels hold more
    # points); the GPU sums in double and rounds once
    assert np.allclose(out[:, :3], ref[:, :3], rtol=0, atol=1e-5 if leaf < 0.05 else 3e-4)
    assert (out[:, 4].view(np.uint32) == ref[:, 4].view(np.uint32)).all()
    # every centroid lies in its own voxel, voxels ascend like PCL's sorted index
    inv = np.float32(1.0) / np.float32(leaf)
    lo = np.floor(np.nanmin(pts[:, :3], axis=0) * inv)
    ijk = (np.floor(out[:, :3] * inv) - lo).astype(np.float64)
    dims = (np.floor(np.nanmax(pts[:, :3], axis=0) * inv) - lo + 1).astype(np.float64)
    key = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    assert (np.diff(key) > 0).mean() > 0.999  # a centroid may round onto a voxel face


@pytest.mark.parametrize("n", [2047, 2048, 4096, 1_048_575, 1_048_576])
def test_voxel_grid_sizes_around_the_scan_forms(gpu, n):
    """pcc_voxel_grid scans n + 1 flags: 2048 words are one workgroup of the scan, up to 512 workgroups (1 048 576 words) take
    the single-launch chained form (csrc/pack.hip k_scan_chained: tagged totals, a workgroup waits for the ones before it),
    one word more the two-level form -- every form and every edge of them must give the oracle's voxels"""
    pts = synth.corridor_cloud(n, synth.SEED_B)
    with capi.Index(pts[:64]) as ctx:
        out = ctx.voxel_grid(pts, 0.05)
        again = ctx.voxel_grid(pts, 0.05)          # (a second scan on the handle: the next epoch over the same flag words)
        ctx.set_option(capi.OPT_SCAN_CHAINED, 0)   # the two-launch scan kept reachable (round 6): same voxels
        plain = ctx.voxel_grid(pts, 0.05)
    ref, nv = oracle.voxel_grid(pts, 0.05)
    assert len(out) == nv == len(again) == len(plain)
    assert np.allclose(out[:, :3], ref[:, :3], rtol=0, atol=1e-4) and (out.view(np.uint32) == again.view(np.uint32)).all()
    assert (out.view(np.uint32) == plain.view(np.uint32)).all()


def test_voxel_grid_refuses_absurd_leaf(gpu):
    pts = synth.corridor_cloud(5000, synth.SEED_A)
    with capi.Index(pts) as ctx:
        with pytest.raises(capi.PccError) as e:
            ctx.voxel_grid(pts, 1e-4)
    assert e.value.status == -5


def test_sorted_radius_rows_of_every_length_class(gpu):
    """rows of 1..64, 65..128, 129..256, 257..512 entries (one wave sorts them in 1, 2, 4, 8 registers per lane)
    and longer ones (in-place fallback), with duplicated distances: order must be (d2, index) everywhere"""
    rng = np.random.default_rng(17)
    # blobs of very different densities around the query points
    centres = rng.random((40, 3)).astype(np.float32) * 4
    sizes = rng.integers(5, 900, 40)
    pts = np.concatenate([c + rng.normal(0, 0.02, (s, 3)) for c, s in zip(centres, sizes)]).astype(np.float32)
    pts = np.ascontiguousarray(np.round(pts, 2))  # 1 cm lattice: many equal distances, ties decided by the index
    pts = np.ascontiguousarray(pts[rng.permutation(len(pts))])
    q = np.ascontiguousarray(np.round(centres, 2))
    with capi.Index(pts) as ix:
        offs, idx, d2 = ix.radius_search(q, 0.08, sorted=True)
    lens = np.diff(offs)
    assert lens.max() > 512 and ((lens > 64) & (lens <= 128)).any() and ((lens > 128) & (lens <= 512)).any()
    for i in range(len(q)):
        s, e = offs[i], offs[i + 1]
        dd = ((pts - q[i]) ** 2).astype(np.float32)
        want_d2 = (dd[:, 0] + dd[:, 1]) + dd[:, 2]
        inside = np.nonzero(want_d2 < np.float32(0.08 * 0.08))[0]
        order = np.lexsort((inside, want_d2[inside]))
        assert e - s == len(inside)
        assert (idx[s:e] == inside[order]).all()
        assert (_bits(d2[s:e]) == _bits(want_d2[inside][order])).all()


def test_bad_arguments_are_refused_with_a_message(gpu):
    """error behaviour of the widened entry points on a live handle: PCC_ERR_INVALID (-1) / UNSUPPORTED (-5),
    never a crash, and pcc_last_error() says why"""
    import ctypes as C
    pts = _scene(5000)
    L = capi.LIB
    with capi.Index(pts) as ix:
        h = ix._h
        out = np.zeros(4 * len(pts), np.float32)
        lab = np.zeros(len(pts), np.int32)
        n32, cnt = C.c_int32(0), C.c_size_t(0)
        cases = [
            (L.pcc_knn(h, pts.ctypes.data, 10, 12, 0, 0, lab.ctypes.data, out.ctypes.data), -5),              # k = 0
            (L.pcc_knn(h, pts.ctypes.data, 10, 12, 0, 1 << 20, lab.ctypes.data, out.ctypes.data), -5),        # k too large
            (L.pcc_knn(h, pts.ctypes.data, 10, 10, 0, 3, lab.ctypes.data, out.ctypes.data), -1),              # stride
            (L.pcc_radius_count(h, pts.ctypes.data, 10, 12, 0, C.c_double(-1.0), lab.ctypes.data), -1),       # negative radius
            (L.pcc_radius_count(h, pts.ctypes.data, 10, 12, 7, C.c_double(0.1), lab.ctypes.data), -1),        # memory space
            (L.pcc_normals(h, 0, None, 0, out.ctypes.data), -5),
            (L.pcc_normals(h, 10, None, 0, None), -1),
            (L.pcc_region_growing(h, None, 0, 10, C.c_float(0.1), C.c_float(1.0), 1, 10, lab.ctypes.data, C.byref(n32)), -1),
            (L.pcc_region_growing(h, out.ctypes.data, 0, 0, C.c_float(0.1), C.c_float(1.0), 1, 10, lab.ctypes.data, C.byref(n32)), -5),
            (L.pcc_sac_plane(h, pts.ctypes.data, 100, 12, 0, 10, C.c_double(0.02), C.c_double(1.5), 1, lab.ctypes.data,
                             C.byref(cnt), out.ctypes.data, None), -1),                                            # probability
            (L.pcc_sac_plane(h, pts.ctypes.data, 100, 12, 0, 10, C.c_double(0.02), C.c_double(0.99), 1, None,
                             C.byref(cnt), out.ctypes.data, None), -1),                                            # null inliers
            (L.pcc_voxel_grid(h, pts.ctypes.data, 100, 12, 0, C.c_float(0.0), 0, out.ctypes.data, 12, C.byref(cnt)), -1),
            (L.pcc_voxel_grid(h, pts.ctypes.data, 100, 12, 0, C.c_float(0.1), 1, out.ctypes.data, 12, C.byref(cnt)), -1),  # rgb needs 20 B
            (L.pcc_first_within(h, pts.ctypes.data, 10, 12, 0, C.c_double(float("nan")), lab.ctypes.data), -1),
            (L.pcc_sor(h, 0, C.c_double(1.0), 0, None, None, None, None), -5),
        ]
        for i, (got, want) in enumerate(cases):
            assert got == want, (i, got, want)
        assert len(L.pcc_last_error()) > 0
        # the handle is still usable afterwards
        idx, d2 = ix.nn1(pts[:10])
        assert (idx == np.arange(10)).all() and (d2 == 0).all()


def test_sharded_icp_loop_equals_icp_align(gpu):
    """pcc_rigid_from_sums + the python sharded loop (one rank here) reproduce pcc_icp_align; the solver agrees
    with the oracle's Umeyama on the same sums"""
    from pointcloudcomparator_amd import sharding
    tgt = _scene(20000)
    src = synth.rigid_offset(tgt[:8000].copy(), jitter=0.001)
    with capi.Index(tgt) as ix:
        T_ref, fit, its, conv = ix.icp_align(src, max_iter=6, fixed=True)
        T, it, mse = sharding.icp_align_sharded(lambda p: ix.icp_step(p, want_corr=False)[2], ix.transform,
                                                capi.rigid_from_sums, src, 6, None)
        _, _, sums = ix.icp_step(src)
    assert it == its == 6
    assert (T.view(np.uint32) == np.asarray(T_ref, np.float32).reshape(4, 4).view(np.uint32)).all()
    rc, To = oracle.umeyama_from_sums(sums)
    assert rc == 0
    np.testing.assert_allclose(capi.rigid_from_sums(sums), To.reshape(4, 4), atol=2e-6)
    with pytest.raises(capi.PccError):
        capi.rigid_from_sums(np.zeros(17))
