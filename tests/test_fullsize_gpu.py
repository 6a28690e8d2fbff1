"""Parity at BASELINE.json's full sizes through size-independent properties: engine-vs-engine
bit identity, self-query idempotence, a CPU-oracle check on a fixed sample of the queries,
shard-concatenation identity, and the known-by-construction cluster partition."""
import numpy as np
import pytest

import oracle
from pointcloudcomparator_amd import capi, sharding, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def _sample_check(ref, qry, idx, d2, n_sample=20000, seed=123):
    """fixed sample of the queries against the CPU kd-tree restatement (bit-exact)"""
    rng = np.random.default_rng(seed)
    sel = np.sort(rng.choice(len(qry), size=min(n_sample, len(qry)), replace=False))
    tree = oracle.KdTree(ref)
    oi, od = tree.nn1_batch(np.ascontiguousarray(qry[sel]))
    assert (_bits(d2[sel]) == _bits(od)).all()
    diff = np.nonzero(idx[sel] != oi)[0]
    for j in diff:  # an index may differ only between exact-distance ties, lowest index on the GPU side
        assert idx[sel][j] < oi[j]
        assert _bits(oracle.nn1_exhaustive(ref[oi[j]:oi[j] + 1], qry[sel][j:j + 1])[1])[0] == _bits(od)[j]


def test_c2_1m_x_1m_grid_vs_brute_vs_oracle_sample(gpu):
    a = synth.corridor_cloud(1_000_000, synth.SEED_A)
    b = synth.corridor_cloud(1_000_000, synth.SEED_B)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    with capi.Index(ta, engine=capi.ENGINE_GRID) as ix:
        gi, gd = ix.nn1(tb)
        ix.sync()
        assert ix.stats()[1] == 0 or ix.stats()[1] < 1000  # the cell walk resolves (almost) everything
        ix.set_engine(capi.ENGINE_BRUTE)
        bi, bd = ix.nn1(tb)
        ix.sync()
    assert bool((gi == bi).all()) and bool((gd.view(torch.int32) == bd.view(torch.int32)).all())
    _sample_check(a, b, gi.cpu().numpy(), gd.cpu().numpy())


def test_self_query_is_identity(gpu):
    a = synth.corridor_cloud(2_000_000, synth.SEED_A)
    ta = torch.from_numpy(a).cuda()
    with capi.Index(ta) as ix:
        idx, d2 = ix.nn1(ta)
        ix.sync()
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    assert (d2 == 0).all()
    # a point's NN in its own cloud is itself unless an exact duplicate with a lower index exists
    moved = np.nonzero(idx != np.arange(len(a)))[0]
    assert (idx[moved] < moved).all() and (a[idx[moved]] == a[moved]).all()


def test_c3_10m_xyzrgb_stride_sampled(gpu):
    a = synth.with_rgb_stride(synth.corridor_cloud(10_000_000, synth.SEED_A))
    b = synth.with_rgb_stride(synth.corridor_cloud(10_000_000, synth.SEED_B))
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    with capi.Index(ta) as ix:
        idx, d2 = ix.nn1(tb)
        ix.sync()
    _sample_check(a, b, idx.cpu().numpy(), d2.cpu().numpy(), n_sample=10000)


def test_search_staged_beside_the_build_equals_the_search_behind_it(gpu):
    """PCC_OPT_OVERLAP_PREP: a k = 1 search that follows setInputCloud directly packs and sorts its queries on the library's
    second stream while the build's sort runs.  Same bits as one stream; calls in between fall back to one stream; a
    producer stream announced between the two calls is honoured by the second stream as well."""
    a = synth.corridor_cloud(3_000_000, synth.SEED_A)
    a2 = synth.corridor_cloud(2_500_000, synth.SEED_A + 5)
    b = synth.corridor_cloud(2_600_000, synth.SEED_B)
    b[7] = np.nan
    ta, ta2, tb = torch.from_numpy(a).cuda(), torch.from_numpy(a2).cuda(), torch.from_numpy(b).cuda()
    torch.cuda.synchronize()

    def same(x, y):
        return bool((x[0] == y[0]).all()) and bool((x[1].view(torch.int32) == y[1].view(torch.int32)).all())

    def outs(n):
        return torch.empty(n, dtype=torch.int32, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda")

    # (auto_sync off: nothing orders the library's stream against torch's but ix.sync(), so every result has buffers of its own)
    nq = len(b)
    ref_a, ref_a2, got_a, got_a2, got, small, mid = outs(nq), outs(nq), outs(nq), outs(nq), outs(nq), outs(100_000), outs(2_200_000)
    with capi.Index(ta, engine=capi.ENGINE_GRID, auto_sync=False) as ix:
        ix.set_option(capi.OPT_OVERLAP_PREP, 0)
        ix.set_input(ta)
        ix.nn1(tb, *ref_a)
        ix.set_input(ta2)
        ix.nn1(tb, *ref_a2)
        ix.sync()
        _sample_check(a, b[100:], ref_a[0][100:].cpu().numpy(), ref_a[1][100:].cpu().numpy(), n_sample=3000)
        assert int(ref_a[0][7]) == -1
        ix.set_option(capi.OPT_OVERLAP_PREP, 1)
        for _ in range(3):  # back to back, no host wait in between: the next build follows the search on the stream
            ix.set_input(ta)
            ix.nn1(tb, *got_a)
            ix.set_input(ta2)
            ix.nn1(tb, *got_a2)
        ix.sync()
        assert same(got_a, ref_a) and same(got_a2, ref_a2)
        # something else between the build and the search: the search runs behind it on the one stream
        ix.set_input(ta)
        ki, kd = ix.knn(tb[:50_000], 4)
        ix.nn1(tb, *got)
        ix.sync()
        assert same(got, ref_a)
        assert bool((ki[:, 0] == ref_a[0][:50_000]).all())
        # 2.2M queries: beside the build too, and the TWO-level query sort (below PCC_OPT_SORT_MP_MIN_Q) on the second scratch set
        ix.set_input(ta)
        ix.nn1(tb[:2_200_000], *mid)
        ix.sync()
        assert same(mid, (ref_a[0][:2_200_000], ref_a[1][:2_200_000]))
        # a second search on the same build, and one of a shard below the staging's crossover
        got[0].fill_(-5)
        torch.cuda.synchronize()
        ix.nn1(tb, *got)
        ix.nn1(tb[:100_000], *small)
        ix.sync()
        assert same(got, ref_a) and same(small, (ref_a[0][:100_000], ref_a[1][:100_000]))
        # the queries are produced on another stream, announced between the build and the search
        producer = torch.cuda.Stream()
        tq = torch.zeros_like(tb)
        big = torch.randn(4096, 4096, device="cuda")
        got[0].fill_(-5)
        torch.cuda.synchronize()
        with torch.cuda.stream(producer):
            for _ in range(30):
                big = (big @ big) * 1e-3
            tq.copy_(tb)
        ix.set_input(ta2)
        ix.wait_stream(producer.cuda_stream)
        ix.nn1(tq, *got)
        ix.sync()
        torch.cuda.synchronize()
        assert same(got, ref_a2)


def test_c3_object_layer_clusters_known_partition(gpu):
    # 5M points of the object layer: the 256 balls are the clusters by construction
    pts = synth.corridor_cloud(5_000_000, synth.SEED_A, layer="objects")
    with capi.Index(torch.from_numpy(pts).cuda()) as ix:
        labels, ncl, sizes = ix.euclidean_clusters(0.05, 100, 250000)
    assert ncl == synth.N_BALLS and sizes.sum() == len(pts) and (np.diff(sizes) <= 0).all()
    centres = synth.ball_centres()
    # every cluster's points lie within one ball, and distinct clusters sit in distinct balls
    first = np.array([np.argmax(labels == k) for k in range(0, ncl, 17)])
    owner = np.linalg.norm(pts[first][:, None, :] - centres[None], axis=2).argmin(1)
    assert len(set(owner)) == len(owner)
    sub = np.arange(0, len(pts), 997)
    ball_of = np.linalg.norm(pts[sub][:, None, :] - centres[None], axis=2).argmin(1)
    lab = labels[sub]
    for bidx in np.unique(ball_of)[:40]:
        assert len(np.unique(lab[ball_of == bidx])) == 1


def test_c5_shards_concatenate_to_the_unsharded_result(gpu):
    a = synth.corridor_cloud(2_000_000, synth.SEED_A)
    b = synth.corridor_cloud(1_000_003, synth.SEED_B)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    with capi.Index(ta) as ix:
        fi, fd = ix.nn1(tb)
        ix.sync()
        parts = []
        for r in range(8):
            s, c = sharding.shard_range(len(b), r, 8)
            i_, d_ = ix.nn1(tb[s:s + c])
            ix.sync()
            parts.append((i_.clone(), d_.clone()))
    ci = torch.cat([p[0] for p in parts])
    cd = torch.cat([p[1] for p in parts])
    assert bool((ci == fi).all()) and bool((cd.view(torch.int32) == fd.view(torch.int32)).all())


def test_c4_icp_fixed_iterations_consistent_with_stepwise(gpu):
    tgt = synth.corridor_cloud(500_000, synth.SEED_A)
    src = synth.rigid_offset(synth.corridor_cloud(200_000, synth.SEED_B), rot_deg=0.3, t=(0.01, -0.01, 0.005))
    tt, ts = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
    with capi.Index(tt) as ix:
        T, fit, it, conv = ix.icp_align(ts, max_iter=10, fixed=True)
        # replay the same loop through the step API: identical correspondences => identical transform
        cur = ts.clone()
        Tacc = np.eye(4, dtype=np.float32)
        for _ in range(10):
            _, _, sums = ix.icp_step(cur, want_corr=False)
            rc, Ti = oracle.umeyama_from_sums(sums)
            assert rc == 0
            cur = ix.transform(Ti, cur)
            Tacc = (Ti @ Tacc).astype(np.float32)
    assert it == 10 and conv
    assert np.allclose(T, Tacc, atol=5e-5)


def _mat4_mul_f32(a, b):
    """float 4x4 product with the accumulation order of the library's mat4_mul_f (k ascending, one float accumulator)"""
    out = np.zeros((4, 4), dtype=np.float32)
    for r in range(4):
        for c in range(4):
            acc = np.float32(0)
            for k in range(4):
                acc = np.float32(acc + np.float32(a[r, k] * b[k, c]))
            out[r, c] = acc
    return out


def test_c4_icp_50_iterations_2m_x_2m(gpu):
    """BASELINE configs[3] at its full size: -i ICP, 50 fixed iterations, 2M source points against 2M targets
    (reference src/comparator.cpp:1089-1099).  The device-resident loop is replayed pass by pass through the step
    API (same correspondences => the same transform, bit for bit), the correspondences of the first and of the
    last pass are checked against the CPU kd-tree on a fixed sample, and the oracle's Umeyama agrees with the
    library's Horn solution on those passes."""
    tgt = synth.corridor_cloud(2_000_000, synth.SEED_A)
    src = synth.rigid_offset(synth.corridor_cloud(2_000_000, synth.SEED_B))
    tt, ts = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
    tree = oracle.KdTree(tgt)
    sel = np.sort(np.random.default_rng(50).choice(len(src), size=20000, replace=False))
    with capi.Index(tt) as ix:
        T, fit, it, conv = ix.icp_align(ts, max_iter=50, fixed=True)
        assert it == 50 and conv
        # (the loop as it was until round 5: the source in the caller's order, the sums added up in the order pcc_icp_step adds
        # them -- the replay below reproduces THIS one to the last bit of the fitness too)
        ix.set_option(capi.OPT_ICP_SORTED, 0)
        T0, fit0, it0, conv0 = ix.icp_align(ts, max_iter=50, fixed=True)
        ix.set_option(capi.OPT_ICP_SORTED, 1)
        assert it0 == 50 and conv0
        cur = ts.clone()
        Tacc = np.eye(4, dtype=np.float32)
        mse = []
        for p in range(50):
            want_corr = p in (0, 49)
            ci, cd, sums = ix.icp_step(cur, want_corr=want_corr)
            assert sums[16] == len(src)  # every source point has a correspondence (no distance threshold)
            mse.append(sums[15] / sums[16])
            Ti = capi.rigid_from_sums(sums)
            if want_corr:
                q = np.ascontiguousarray(cur.cpu().numpy()[sel])
                oi, od = tree.nn1_batch(q)
                gi, gd = ci.cpu().numpy()[sel], cd.cpu().numpy()[sel]
                assert (_bits(gd) == _bits(od)).all()
                for j in np.nonzero(gi != oi)[0]:  # exact-distance ties only, lowest index on the GPU side
                    assert gi[j] < oi[j] and _bits(oracle.nn1_exhaustive(tgt[oi[j]:oi[j] + 1], q[j:j + 1])[1])[0] == _bits(od)[j]
                rc, To = oracle.umeyama_from_sums(sums)
                assert rc == 0 and np.allclose(Ti, To, atol=2e-5)
            cur = ix.transform(Ti, cur)
            Tacc = _mat4_mul_f32(Ti, Tacc)
        # getFitnessScore: the INPUT moved by the final matrix, one more NN pass, mean squared distance
        _, _, fs = ix.icp_step(ix.transform(T, ts), want_corr=False)
    assert (T0.view(np.uint32) == Tacc.view(np.uint32)).all() and fit0 == fs[15] / fs[16]
    # the default loop keeps its source in the target grid's cell order: the same correspondences, the 17 double sums added up in
    # another order -- the transform comes out with the same float bits here, the fitness within the rounding of a 2M-term sum
    assert (T.view(np.uint32) == Tacc.view(np.uint32)).all()
    assert abs(fit - fs[15] / fs[16]) <= 1e-12 * fit
    assert mse[-1] < mse[0]  # (the floor is the sampling distance of two different samples of the scene, not zero)
    R = T[:3, :3].astype(np.float64)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)
    ang = np.degrees(np.arctan2(R[1, 0], R[0, 0]))
    # rigid_offset turned cloud B by +2 degrees: the loop turns it back (point-to-point ICP between two different
    # samples of a scene creeps -- about half of the angle after 50 passes -- so only the direction is pinned)
    assert -2.2 < ang < -0.5 and T[0, 3] < 0 < T[1, 3], (ang, T[:3, 3])


def test_c5_one_full_shard_4m_vs_8m(gpu):
    """BASELINE configs[4]: one of the 8 shards at its real size -- 4M queries (shard 3 of cloud B's 32M) against
    the 8M-point reference cloud -- grid against the CPU kd-tree on a fixed sample, and the neighbouring shard
    boundary: the last queries of shard 3 searched alone equal the same rows of the full shard."""
    a = np.concatenate([synth.corridor_cloud(4_000_000, synth.SEED_A, start=o) for o in (0, 4_000_000)])
    start, count = sharding.shard_range(32_000_000, 3, 8)
    assert (start, count) == (12_000_000, 4_000_000)
    b = synth.corridor_cloud(count, synth.SEED_B, start=start)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    with capi.Index(ta) as ix:
        idx, d2 = ix.nn1(tb)
        tail_i, tail_d = ix.nn1(tb[-1000:])
        assert ix.stats()[2] == len(a)
        hi, hd = idx.cpu().numpy(), d2.cpu().numpy()
        assert (tail_i.cpu().numpy() == hi[-1000:]).all() and (_bits(tail_d.cpu().numpy()) == _bits(hd[-1000:])).all()
    assert (hi >= 0).all() and np.isfinite(hd).all()
    _sample_check(a, b, hi, hd, n_sample=20000, seed=5)


def test_1m_knn_normals_region_growing_against_the_sequential_walk(gpu):
    """1M corridor points: the k-NN rows (sampled against the exhaustive oracle), the normals built on them and
    the GPU's order-free region growing against PCL's sequential walk (oracle) over the same rows"""
    n, k = 1_000_000, 20
    pts = synth.corridor_cloud(n, synth.SEED_A)
    with capi.Index(pts) as ix:
        nbr, d2 = ix.knn(pts, k)
        nrm = ix.normals(k)
        labels, ncl = ix.region_growing(nrm, k=k, smoothness=8.0 / 180.0 * np.pi, curvature_threshold=1.0,
                                        min_size=30, max_size=n)
    # rows: sorted by (d2, idx), self first, and a sample equals the exhaustive oracle
    assert (nbr[:, 0] == np.arange(n)).mean() > 0.9999 and (np.diff(d2.astype(np.float64), axis=1) >= 0).all()
    sample = np.arange(0, n, 4001)
    oi, od = oracle.knn_exhaustive(pts, pts[sample], k)
    assert (oi == nbr[sample]).all() and (od.view(np.uint32) == d2[sample].view(np.uint32)).all()
    # normals: the oracle's bits on the same rows, all 1M of them (the libm calls included: DESIGN 4.6)
    want = oracle.normals(pts, k, neighbours=nbr)
    same = (nrm.view(np.uint32) == want.view(np.uint32)).all(axis=1) | (np.isnan(nrm).all(axis=1) & np.isnan(want).all(axis=1))
    assert same.all(), int((~same).sum())
    ok = np.isfinite(nrm).all(axis=1) & np.isfinite(want).all(axis=1)
    assert (nrm[ok, 3] >= 0).all() and (nrm[ok, 3] <= 1 / 3 + 1e-6).all()
    # region growing: label for label
    want_labels, want_n = oracle.region_growing(nrm, nbr, 8.0 / 180.0 * np.pi, 1.0, 30, n)
    assert ncl == want_n and (labels == want_labels).all()
    assert ncl > 10


def test_2m_ransac_plane_against_the_restatement(gpu):
    rng = np.random.default_rng(5)
    n = 2_000_000
    plane = np.stack([rng.random(n // 2) * 20 - 5, rng.random(n // 2) * 20 - 5, 0.3 + rng.normal(0, 0.006, n // 2)], 1)
    rest = rng.random((n - n // 2, 3)) * [20, 20, 3] - [5, 5, 0]
    pts = np.ascontiguousarray(np.concatenate([plane, rest]).astype(np.float32)[rng.permutation(n)])
    with capi.Index(pts[:1]) as ctx:
        inl, coeff, its = ctx.sac_plane(pts)
    w_inl, w_c, w_its = oracle.sac_plane(pts)
    assert its == w_its and (coeff.view(np.uint32) == w_c.view(np.uint32)).all()
    assert len(inl) == len(w_inl) and (inl == w_inl).all()
    assert len(inl) > 0.49 * n


def test_room_scan_surface_sampled_scene(gpu):
    """the reference's real inputs are scans of a room -- points on 2-D surfaces (reference build/results.txt:4-5: 346 911
    and 1 379 736 points) -- not the volumetric corridor scene: k = 1 with both kernel forms against each other and
    against the CPU kd-tree on a sample, 51-NN rows (the -n noise pass) on a sample against the exhaustive oracle, and the
    furniture's known partition under the -e clustering"""
    n = synth.ROOM_SIZES[1]
    a, b = synth.room_cloud(n, synth.SEED_A), synth.room_cloud(n, synth.SEED_B)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    res = []
    with capi.Index(ta, engine=capi.ENGINE_GRID) as ix:
        for form in (1, 0, 2, 3):
            ix.set_option(capi.OPT_NN1_KERNEL, form)
            idx, d2 = ix.nn1(tb)
            ix.sync()
            res.append((idx.cpu().numpy(), d2.cpu().numpy()))
        assert ix.stats()[1] < 1000
        ix.set_option(capi.OPT_NN1_KERNEL, 1)
        sel = np.sort(np.random.default_rng(5).choice(n, 200, replace=False))
        ki, kd = ix.knn(np.ascontiguousarray(b[sel]), 51)
        # radius search on a surface scan: the points pile up in the cells the surfaces cross, so the count pass takes its
        # wave form (the rule looks at the filling of OCCUPIED cells); counts of every query, rows of the sample
        cnt = ix.radius_count(tb, 0.05)
        ix.sync()
        cnt = cnt.cpu().numpy()
        ro, ri, rd = ix.radius_search(np.ascontiguousarray(b[sel]), 0.05, sorted=True)
    ecnt = oracle.radius_count_exhaustive(a, np.ascontiguousarray(b[sel]), 0.05)
    assert (cnt[sel] == ecnt).all() and (np.diff(ro) == ecnt).all() and cnt.mean() > 50
    for j in range(0, 200, 20):
        dd = ((a - b[sel[j]]) ** 2).astype(np.float32)
        want = (dd[:, 0] + dd[:, 1]) + dd[:, 2]
        inside = np.nonzero(want < np.float32(0.05 * 0.05))[0]
        order = np.lexsort((inside, want[inside]))
        assert (ri[ro[j]:ro[j + 1]] == inside[order]).all() and (_bits(rd[ro[j]:ro[j + 1]]) == _bits(want[inside][order])).all()
    for idx, d2 in res[1:]:
        assert (idx == res[0][0]).all() and (_bits(d2) == _bits(res[0][1])).all()
    _sample_check(a, b, res[0][0], res[0][1])
    ei, ed = oracle.knn_exhaustive(a, np.ascontiguousarray(b[sel]), 51)
    assert (_bits(kd) == _bits(ed)).all() and (ki == ei).all()
    m = 400_000
    f = synth.room_cloud(m, synth.SEED_A, part="furniture")
    with capi.Index(f) as fx:
        labels, ncl, sizes = fx.euclidean_clusters(0.05, 100, 2_000_000_000)
    assert ncl == len(synth._ROOM_BOXES) and int(sizes.sum()) == m and (labels >= 0).all()
