"""GPU parity of pcc_nn1 (through the C-ABI) against the oracle: bit-exact indices and d2."""
import numpy as np
import pytest

import oracle
from pointcloudcomparator_amd import capi, synth

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def _check(ref, qry, engine):
    with capi.Index(ref, engine=engine) as ix:
        idx, d2 = ix.nn1(qry)
        stats = ix.stats()
    oi, od = oracle.nn1_exhaustive(ref, qry)
    assert (_bits(d2) == _bits(od)).all(), f"d2 bits differ at {np.nonzero(_bits(d2) != _bits(od))[0][:5]}"
    assert (idx == oi).all(), f"idx differ at {np.nonzero(idx != oi)[0][:5]}"
    return stats


@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
@pytest.mark.parametrize("m,n", [(10000, 10000), (1, 7), (15, 100), (16, 1), (1025, 513), (40000, 3000)])
def test_nn1_corridor(gpu, engine, m, n):
    a = synth.corridor_cloud(m, synth.SEED_A)
    b = synth.corridor_cloud(n, synth.SEED_B)
    _check(a, b, engine)


@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
def test_nn1_strides_and_nonfinite(gpu, engine):
    a = synth.with_rgb_stride(synth.corridor_cloud(5000, synth.SEED_A))  # 32-byte PointXYZRGB stride
    b = synth.with_rgb_stride(synth.corridor_cloud(3000, synth.SEED_B))
    a[17, 0] = np.nan
    a[99, 2] = np.inf
    a[4999, 1] = -np.inf
    b[5, 1] = np.nan
    b[2999, 0] = np.inf
    with capi.Index(a, engine=engine) as ix:
        assert ix.size == 4997
        idx, d2 = ix.nn1(b)
    oi, od = oracle.nn1_exhaustive(a, b)
    assert idx[5] == -1 and np.isinf(d2[5]) and idx[2999] == -1
    assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()
    assert not np.isin(idx, [17, 99, 4999]).any()


@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
def test_nn1_pointxyz_16_byte_stride(gpu, engine):
    """pcl::PointXYZ: x, y, z + one float of padding = 16-byte elements, host and device"""
    import torch
    a = np.full((6000, 4), 1.0, dtype=np.float32)
    b = np.full((2500, 4), 1.0, dtype=np.float32)
    a[:, :3] = synth.corridor_cloud(6000, synth.SEED_A)
    b[:, :3] = synth.corridor_cloud(2500, synth.SEED_B)
    a[:, 3] = np.nan  # the padding word is never read as a coordinate
    assert a.strides == (16, 4)
    oi, od = oracle.nn1_exhaustive(a, b)
    with capi.Index(a, engine=engine) as ix:
        idx, d2 = ix.nn1(b)
        ki, kd = ix.knn(b[:300], 7)
    assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()
    ei, ed = oracle.knn_exhaustive(a, b[:300], 7)
    assert (ki == ei).all() and (_bits(kd) == _bits(ed)).all()
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    with capi.Index(ta, engine=engine) as ix:
        ti, td = ix.nn1(tb)
        assert (ti.cpu().numpy() == oi).all() and (_bits(td.cpu().numpy()) == _bits(od)).all()
        # a strided VIEW of a wider device array (every second row of 32-byte elements = 64-byte stride)
        wide = torch.zeros((5000, 8), dtype=torch.float32, device="cuda")
        wide[::2, :3] = tb[:, :3]
        vi, vd = ix.nn1(wide[::2])
        assert (vi.cpu().numpy() == oi).all() and (_bits(vd.cpu().numpy()) == _bits(od)).all()


@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
def test_nn1_duplicates_lowest_index(gpu, engine):
    rng = np.random.default_rng(7)
    base = rng.random((2000, 3), dtype=np.float32)
    a = np.concatenate([base, base[::-1], base[:500]])  # every point duplicated, some tripled
    q = base[rng.integers(0, 2000, 1500)] + np.float32(0)
    with capi.Index(a, engine=engine) as ix:
        idx, d2 = ix.nn1(q)
    oi, od = oracle.nn1_exhaustive(a, q)
    assert (d2 == 0).all()
    assert (idx == oi).all()


@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
def test_nn1_queries_far_outside(gpu, engine):
    a = synth.corridor_cloud(20000, synth.SEED_A)
    b = synth.corridor_cloud(2000, synth.SEED_B)
    b[:500] += np.float32(50.0)      # far outside the reference bounding box
    b[500:1000] *= np.float32(-3.0)
    stats = _check(a, b, engine)


@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
def test_nn1_degenerate_shapes(gpu, engine):
    rng = np.random.default_rng(3)
    q = rng.random((700, 3), dtype=np.float32)
    plane = rng.random((6000, 3), dtype=np.float32)
    plane[:, 2] = 0.25
    line = np.zeros((5000, 3), np.float32)
    line[:, 0] = rng.random(5000, dtype=np.float32)
    same = np.full((4500, 3), 0.5, np.float32)
    for ref in (plane, line, same):
        _check(ref, q, engine)


def test_device_memory_path(gpu):
    torch = pytest.importorskip("torch")
    a = synth.corridor_cloud(30000, synth.SEED_A)
    b = synth.corridor_cloud(20000, synth.SEED_B)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    with capi.Index(ta, engine=capi.ENGINE_GRID) as ix:
        idx, d2 = ix.nn1(tb)
        ix.sync()
        idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    oi, od = oracle.nn1_exhaustive(a, b)
    assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()


def test_far_queries_take_the_seed_and_ball_route(gpu):
    """queries far outside the reference cloud (misaligned ICP source): seed scan + ball walk (PCC_OPT_FAR_MODE 1),
    the exhaustive fallback alone (0) and the heuristic (-1) -- all bit-identical to the oracle, and the counters show
    that the modes really took different routes"""
    a = synth.corridor_cloud(60000, synth.SEED_A)
    b = synth.corridor_cloud(6000, synth.SEED_B)
    b[:2000] += np.float32([1.5, -2.0, 0.7])
    b[2000:3000] += np.float32(30.0)
    oi, od = oracle.nn1_exhaustive(a, b)
    left = {}
    for mode in (1, -1, 0):
        with capi.Index(a, engine=capi.ENGINE_GRID) as ix:
            ix.set_option(capi.OPT_FAR_MODE, mode)
            assert ix.get_option(capi.OPT_FAR_MODE) == mode
            for _ in range(2):  # second call: the heuristic has seen the first call's fallbacks
                idx, d2 = ix.nn1(b)
                assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()
            left[mode] = ix.stats()[1]  # queries the cell walk did not resolve
    assert left[0] == left[1] == left[-1] > 0


# ---- PCC_TIES_FLANN: among equally near references, the one pcl::KdTreeFLANN's tree walk reaches first ---------------
def _tie_clouds():
    rng = np.random.default_rng(17)
    base = rng.random((3000, 3), dtype=np.float32)
    dup = np.concatenate([base, base[::-1], base[:700]])                      # exact duplicates: d2 = 0 ties
    lattice = (rng.integers(0, 12, (6000, 3)) * np.float32(0.25)).astype(np.float32)  # many equidistant references
    plane = rng.random((5000, 3), dtype=np.float32)
    plane[:, 2] = 0.5
    plane[::3] = np.round(plane[::3] * 8) / 8
    small = (rng.integers(0, 5, (300, 3)) * np.float32(0.5)).astype(np.float32)       # BRUTE engine under AUTO
    q_dup = np.concatenate([base[rng.integers(0, 3000, 800)], rng.random((400, 3), dtype=np.float32)])
    q_lat = np.concatenate([(rng.integers(0, 12, (1500, 3)) * np.float32(0.25) + np.float32(0.125)).astype(np.float32),
                            lattice[:500], rng.random((300, 3), dtype=np.float32) * 3])
    q_plane = np.concatenate([plane[:600] + np.float32([0, 0, 0.25]), rng.random((600, 3), dtype=np.float32)])
    q_small = (rng.integers(0, 9, (500, 3)) * np.float32(0.25)).astype(np.float32)
    return [("duplicates", dup, q_dup), ("lattice", lattice, q_lat), ("plane", plane, q_plane), ("small", small, q_small)]


@pytest.mark.parametrize("engine", [capi.ENGINE_AUTO, capi.ENGINE_BRUTE, capi.ENGINE_GRID])
def test_ties_flann_order_equals_the_kdtree_walk(gpu, engine):
    for name, a, q in _tie_clouds():
        fi, fd = oracle.KdTree(a).nn1_batch(q)          # FLANN restatement: first visited
        li, ld = oracle.nn1_exhaustive(a, q)            # lowest index
        assert (_bits(fd) == _bits(ld)).all()
        assert (fi != li).sum() > 20, name              # the case is not vacuous
        with capi.Index(a, engine=engine) as ix:
            idx, d2 = ix.nn1(q)
            assert (idx == li).all() and (_bits(d2) == _bits(ld)).all(), name
            ix.set_tie_order(capi.TIES_FLANN)
            idx, d2 = ix.nn1(q)
            st = ix.stats()
            assert (_bits(d2) == _bits(ld)).all(), name
            assert (idx == fi).all(), (name, np.nonzero(idx != fi)[0][:5])
            assert st[6] == (fi != li).sum() and st[5] >= st[6], (name, st[5], st[6])
            ix.set_input(a[::-1].copy())                # a new cloud: the host tree is rebuilt
            idx, _ = ix.nn1(q)
            assert (idx == oracle.KdTree(a[::-1].copy()).nn1_batch(q)[0]).all(), name
            ix.set_tie_order(capi.TIES_LOWEST_INDEX)
            assert (ix.nn1(q)[0] == oracle.nn1_exhaustive(a[::-1].copy(), q)[0]).all()


@pytest.mark.parametrize("rule", [0, 1, 2])
def test_ties_flann_split_rules_and_scale(gpu, rule):
    """PCC_OPT_FLANN_SPLIT: the tree the tied queries are walked through (on the device, k_tie_walk) is built with
    middleSplit_ as recalled from FLANN 1.8.4 (0, default), middleSplit (1) or middleSplit_ with the loop variable (2) --
    each against the oracle's recursive tree under the same rule, on a 300k-point mm-quantised scan with 10 % duplicated
    points (tens of thousands of tied queries, the forked build) and after switching the rule on a live handle"""
    import torch
    n = 300_000
    a = np.round(synth.room_cloud(n, synth.SEED_A) * 1000) / 1000
    a[::10] = a[1::10][: len(a[::10])]                      # 10 % exact duplicates
    a = a.astype(np.float32)
    q = (np.round(synth.room_cloud(60_000, synth.SEED_B) * 1000) / 1000).astype(np.float32)
    oracle.set_split_rule(rule)
    try:
        fi, fd = oracle.KdTree(a).nn1_batch(q)
    finally:
        oracle.set_split_rule(0)
    li, ld = oracle.nn1_exhaustive(a[:1], q[:1])  # (touch the exhaustive oracle's loader)
    with capi.Index(torch.from_numpy(a).cuda()) as ix:
        ix.set_tie_order(capi.TIES_FLANN)
        if rule != 0:
            ix.nn1(torch.from_numpy(q[:100]).cuda())          # builds the default tree first: the option must replace it
        ix.set_option(capi.OPT_FLANN_SPLIT, rule)
        idx, d2 = ix.nn1(torch.from_numpy(q).cuda())
        st = ix.stats()
        idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    assert (_bits(d2) == _bits(fd)).all()
    assert (idx == fi).all(), (rule, np.nonzero(idx != fi)[0][:5])
    assert st[5] > 5000 and st[6] > 100, st


def test_ties_flann_device_buffers_and_match_knn(gpu):
    import torch
    name, a, q = _tie_clouds()[1]
    fi, fd = oracle.KdTree(a).nn1_batch(q)
    with capi.Index(torch.from_numpy(a).cuda()) as ix:
        ix.set_tie_order(capi.TIES_FLANN)
        idx, d2 = ix.nn1(torch.from_numpy(q).cuda())
        assert (idx.cpu().numpy() == fi).all() and (_bits(d2.cpu().numpy()) == _bits(fd)).all()
    # matchRIFTFeaturesKnn returns the matched indices to its caller (reference src/comparator.cpp:576-580)
    rng = np.random.default_rng(3)
    d1 = np.zeros((900, 32), np.float32)
    d1[:, :3] = (rng.integers(0, 6, (900, 3)) * np.float32(0.1))
    d2_ = np.zeros((700, 32), np.float32)
    d2_[:, :3] = (rng.integers(0, 11, (700, 3)) * np.float32(0.05))
    want = oracle.match_rift_knn(d1, d2_)
    with capi.Index(d1) as ix:
        low = ix.match_knn(d2_, 0.05)
        ix.set_tie_order(capi.TIES_FLANN)
        got = ix.match_knn(d2_, 0.05)
    assert len(low) == len(want) and (low != want).any()
    assert (got == want).all()


@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
def test_overflowed_distances_are_no_neighbours(gpu, engine):
    """coordinates around 1e19-1e21: every squared distance of the far queries overflows.  FLANN rejects dist >= FLT_MAX
    (its initial worst distance), so nearestKSearch finds nothing: idx -1, d2 +inf, in both engines and tie orders, in
    k-NN rows and in the ICP sums -- and the cube bound of the grid walk must not mistake its own overflow for "the cube
    is the whole grid" (tools/fuzz_gpu.py found that one)."""
    rng = np.random.default_rng(3)
    ref = (rng.random((20000, 3), dtype=np.float32) * np.float32(2e18)).astype(np.float32)
    ref[:, 2] = np.float32(3e17)                                  # a plane: one layer of cells
    far = (rng.random((300, 3), dtype=np.float32) * np.float32(1e21) + np.float32(5e20)).astype(np.float32)
    far[:, 2] = np.float32(3e17)
    near = (ref[:300] * np.float32(1.0001)).astype(np.float32)
    q = np.concatenate([far, near])
    oi, od = oracle.nn1_exhaustive(ref, q)
    assert (oi[:300] == -1).all() and (oi[300:] >= 0).all()
    with capi.Index(ref, engine=engine) as ix:
        for ties in (capi.TIES_LOWEST_INDEX, capi.TIES_FLANN):
            ix.set_tie_order(ties)
            idx, d2 = ix.nn1(q)
            assert (idx[:300] == -1).all() and np.isinf(d2[:300]).all()
            if ties == capi.TIES_LOWEST_INDEX:
                assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()
            else:
                ti, td = oracle.KdTree(ref).nn1_batch(q)
                assert (idx == ti).all() and (_bits(d2) == _bits(td)).all()
        ix.set_tie_order(capi.TIES_LOWEST_INDEX)
        ki, kd = ix.knn(q, 7)
        oki, okd = oracle.knn_exhaustive(ref, q, 7)
        assert (ki == oki).all() and (_bits(kd) == _bits(okd)).all()
        _, _, sums = ix.icp_step(q)
        assert sums[16] == 300          # only the near points have a correspondence



@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
def test_ties_flann_with_non_finite_queries_at_wave_leaders(gpu, engine):
    """a lattice cloud (every query tied) with NaN / inf queries exactly where a wave's lane 0 sits (indices 0, 64, 128...):
    the tie list is appended once per wave by a leader lane, which must be taken from the lanes that carry a tie -- not
    lane 0, which has none here (round-3 advisor finding).  Indices = the kd-tree walk's, counters = the tied queries."""
    g = np.arange(12, dtype=np.float32) * np.float32(0.25)
    ref = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    rng = np.random.default_rng(11)
    cells = rng.integers(0, 11, (1024, 3)).astype(np.float32) * np.float32(0.25)
    q = (cells + np.float32(0.125)).astype(np.float32)          # cell centres: eight references at the same distance
    bad = np.arange(0, 1024, 64)
    q[bad[0::2], 0] = np.nan
    q[bad[1::2], 2] = np.inf
    good = np.ones(len(q), bool)
    good[bad] = False
    fi, fd = oracle.KdTree(ref).nn1_batch(q[good])
    li, _ = oracle.nn1_exhaustive(ref, q[good])
    assert (fi != li).any()                                       # FLANN's order differs from lowest-index somewhere
    with capi.Index(ref, engine=engine) as ix:
        ix.set_tie_order(capi.TIES_FLANN)
        idx, d2 = ix.nn1(q)
        st = ix.stats()
    assert (idx[bad] == -1).all() and np.isinf(d2[bad]).all()
    assert (_bits(d2[good]) == _bits(fd)).all()
    assert (idx[good] == fi).all(), np.nonzero(idx[good] != fi)[0][:8]
    assert st[5] == good.sum() and st[6] == (fi != li).sum(), (st[5], st[6], good.sum(), (fi != li).sum())


@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
@pytest.mark.parametrize("m,n", [(5000, 1), (5000, 63), (4097, 4097), (70000, 20011), (120000, 40000)])
def test_nn1_staged_beside_the_build_at_every_size(gpu, engine, m, n):
    """PCC_OPT_OVERLAP_PREP = 2: the search that follows setInputCloud stages its queries on the second stream whatever the
    size (the default acts from 2M queries on) -- small grids, the two-level sort, host and device clouds, non-finite
    queries, rebuilds over other clouds on the same handle.  The exhaustive engine has no grid: the option must not act."""
    torch = pytest.importorskip("torch")
    a = synth.corridor_cloud(m, synth.SEED_A)
    a2 = synth.corridor_cloud(max(m // 2, 4097), synth.SEED_A + 3)
    b = synth.with_rgb_stride(synth.corridor_cloud(n, synth.SEED_B))
    b[n // 2, 1] = np.nan
    want = {id(a): oracle.nn1_exhaustive(a, b), id(a2): oracle.nn1_exhaustive(a2, b)}

    def ok(got, ref):
        return (np.asarray(got[0]) == ref[0]).all() and (_bits(got[1]) == _bits(ref[1])).all()

    with capi.Index(a, engine=engine) as ix:
        ix.set_option(capi.OPT_OVERLAP_PREP, 2)
        assert ok(ix.nn1(b), want[id(a)])            # first call on a fresh handle (the build has finished)
        for cloud in (a2, a, a2):                    # host clouds: build and search back to back
            ix.set_input(cloud)
            assert ok(ix.nn1(b), want[id(cloud)])
        ta, ta2, tb = torch.from_numpy(a).cuda(), torch.from_numpy(a2).cuda(), torch.from_numpy(b).cuda()
        for cloud, t in ((a, ta), (a2, ta2), (a, ta)):   # device clouds through torch's stream (auto_sync edges on both sides)
            ix.set_input(t)
            gi, gd = ix.nn1(tb)
            ix.sync()
            assert ok((gi.cpu().numpy(), gd.cpu().numpy()), want[id(cloud)])
        ix.set_input(a2)
        ki, _ = ix.knn(b[:50], 2)                    # another call in between: one stream
        assert ok(ix.nn1(b), want[id(a2)])
        assert (ki[:, 0] == want[id(a2)][0][:50]).all()


def test_nn1_on_a_caller_stream_waits_for_the_callers_producer(gpu):
    """pcc_index_set_stream hands the caller's stream over: everything the caller enqueues on it between setInputCloud and the
    search -- here the kernels that PRODUCE the queries -- is ordered before the search's first read, whatever
    PCC_OPT_OVERLAP_PREP says (the staging on the library's second stream waits for the build's grid parameters only, so it
    must not act on a stream it does not own).  The producer is made slow (a chain of passes over a large buffer) and the
    query buffer holds far-away garbage until its last kernel writes it."""
    torch = pytest.importorskip("torch")
    m = n = 300_000
    a = torch.from_numpy(synth.corridor_cloud(m, synth.SEED_A)).cuda()
    b_host = synth.corridor_cloud(n, synth.SEED_B)
    want = oracle.KdTree(synth.corridor_cloud(m, synth.SEED_A)).nn1_batch(b_host[:5000])
    src = torch.from_numpy(b_host).cuda()
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    with capi.Index(a, engine=capi.ENGINE_GRID, auto_sync=False) as ix:
        ix.set_stream(stream.cuda_stream)
        for prep in (2, 1, 0):
            ix.set_option(capi.OPT_OVERLAP_PREP, prep)
            q = torch.full((n, 3), 1.0e6, dtype=torch.float32, device="cuda")
            idx = torch.empty(n, dtype=torch.int32, device="cuda")
            d2 = torch.empty(n, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(stream):
                ix.set_input(a)
                junk = torch.zeros(64_000_000, dtype=torch.float32, device="cuda")
                for _ in range(12):          # a few milliseconds of work ahead of the write
                    junk.add_(1.0)
                q.copy_(src + junk[:1] * 0)  # the queries exist only now
                ix.nn1(q, idx, d2)
            stream.synchronize()
            ix.sync()
            gi, gd = idx[:5000].cpu().numpy(), d2[:5000].cpu().numpy()
            assert (_bits(gd) == _bits(want[1])).all(), prep
            assert (gi == want[0]).all(), prep


@pytest.mark.parametrize("floats", [3, 4, 8])
def test_large_host_clouds_cross_pcie_through_the_pinned_pipe(gpu, floats):
    """PCC_OPT_HOST_PIPE: pageable host clouds and results of 8 MB and more are staged by the library (chunks gathered by host
    threads into pinned buffers, x / y / z only when the stride is 24 bytes or more); the plain hipMemcpyAsync path must return
    the same arrays -- and both the oracle's bits on a sample.  Non-finite points, sizes that are no multiple of the chunk."""
    m, n = 1_500_011, 2_100_003          # 2.1M x 4 B results = 8.4 MB: the download side acts too
    a = synth.corridor_cloud(m, synth.SEED_A)
    b = synth.corridor_cloud(n, synth.SEED_B)
    if floats > 3:
        pad = np.full((1, floats - 3), 7.0, np.float32)
        a = np.ascontiguousarray(np.concatenate([a, np.broadcast_to(pad, (m, floats - 3))], axis=1))
        b = np.ascontiguousarray(np.concatenate([b, np.broadcast_to(pad, (n, floats - 3))], axis=1))
    a[5, 1] = np.nan
    a[m - 1, 0] = np.inf
    b[n - 1, 2] = np.nan
    res = {}
    for pipe in (1, 0):
        with capi.Index(a[:4096], engine=capi.ENGINE_GRID) as ix:
            ix.set_option(capi.OPT_HOST_PIPE, pipe)
            ix.set_input(a)
            res[pipe] = ix.nn1(b)
    assert (res[1][0] == res[0][0]).all() and (_bits(res[1][1]) == _bits(res[0][1])).all()
    sel = np.concatenate([np.arange(0, n, 997), [n - 2, n - 1]])
    oi, od = oracle.KdTree(a[:, :3]).nn1_batch(np.ascontiguousarray(b[sel[:-1], :3]))
    assert (_bits(res[1][1][sel[:-1]]) == _bits(od)).all() and (res[1][0][sel[:-1]] == oi).all()
    assert res[1][0][n - 1] == -1 and np.isinf(res[1][1][n - 1])


def _records(pts, floats):
    """the points as records of `floats` floats (the first three are x, y, z; the rest is payload that must never be read)"""
    rec = np.full((len(pts), floats), np.nan, dtype=np.float32)
    rec[:, :3] = pts
    return rec


@pytest.mark.parametrize("floats", [3, 4, 32])
@pytest.mark.parametrize("m,n", [(1, 1), (4, 4), (100, 100), (63, 65), (1000, 777), (4096, 2500), (4095, 64), (17, 8000)])
def test_small_calls_take_one_launch_and_the_same_bits(gpu, floats, m, n):
    """The reference's descriptor matching (src/comparator.cpp:560-588): small host clouds, 128-byte records.  Up to 4096 indexed
    points the query side of a host call is ONE launch (small.hip) -- against the oracle, and against the separate launches
    (PCC_OPT_HOST_PIPE = 0) bit for bit; non-finite queries and references, either output alone, and pcc_match_knn on top."""
    if floats == 32 and (n - 1) * 128 + 12 > (1 << 20):
        pytest.skip("beyond the pinned small-call buffer: the staged path, covered elsewhere")
    rng = np.random.default_rng(m * 131 + n)
    a = synth.corridor_cloud(m, synth.SEED_A)
    b = synth.corridor_cloud(n, synth.SEED_B)
    if m >= 100:
        a[rng.integers(0, m, 5)] = np.nan
        a[m - 1, 1] = np.inf
        a[7] = a[3]  # an exact tie: the lower index wins
    if n >= 64:
        b[0, 0] = np.nan
        b[63, 2] = -np.inf
        b[n - 1, 1] = np.nan
        b[5] = a[3]
    ra, rb = _records(a, floats), _records(b, floats)
    oi, od = oracle.nn1_exhaustive(a, b)
    with capi.Index(ra) as ix:  # PCC_ENGINE_AUTO: exhaustive below 4096 points
        idx, d2 = ix.nn1(rb)
        assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()
        only_i = np.full(n, 12345, dtype=np.int32)
        capi._check(capi.LIB.pcc_nn1(ix._h, rb.ctypes.data, n, rb.strides[0], capi.MEM_HOST, only_i.ctypes.data, None))
        assert (only_i == oi).all()
        only_d = np.full(n, 7.0, dtype=np.float32)
        capi._check(capi.LIB.pcc_nn1(ix._h, rb.ctypes.data, n, rb.strides[0], capi.MEM_HOST, None, only_d.ctypes.data))
        assert (_bits(only_d) == _bits(od)).all()
        got = ix.match_knn(rb, 0.05)
        want = [0] + [int(i) for i, d in zip(oi, od) if i >= 0 and d < np.float32(0.05)]
        assert list(got) == want
        # FLANN's tie order: the kernel counts the tied queries; none -> the call is complete, some -> the replay on what it left
        fi, fd = oracle.KdTree(a).nn1_batch(b)
        ix.set_tie_order(capi.TIES_FLANN)
        ix.set_input(ra)
        idx_f, d2_f = ix.nn1(rb)
        st = ix.stats()
        assert (idx_f == fi).all() and (_bits(d2_f) == _bits(fd)).all()
        assert st[6] == (fi != oi).sum() and st[5] >= st[6]
        ix.set_tie_order(capi.TIES_LOWEST_INDEX)
        ix.set_option(capi.OPT_HOST_PIPE, 0)
        ix.set_input(ra)
        idx0, d20 = ix.nn1(rb)
        assert (idx0 == idx).all() and (_bits(d20) == _bits(d2)).all()
        assert list(ix.match_knn(rb, 0.05)) == want
        # the handle is left as the separate launches leave it: a device-side consumer of the same index goes on working
        ix.set_option(capi.OPT_HOST_PIPE, 1)
        ix.set_input(ra)
        ix.nn1(rb)
        if m >= 100:
            ki, kd = ix.knn(rb[:50], 3)
            oki, okd = oracle.knn_exhaustive(a, b[:50], 3)
            assert (ki == oki).all() and (_bits(kd) == _bits(okd)).all()


def test_small_calls_with_one_valid_reference_and_overflow(gpu):
    """all references but one non-finite: that one answers every query; distances that overflow: nothing found (as the
    exhaustive kernel); a cloud without any finite point is refused at creation, as PCL refuses an empty tree"""
    a = np.full((50, 3), np.nan, dtype=np.float32)
    b = synth.corridor_cloud(70, synth.SEED_B)
    with pytest.raises(capi.PccError):
        capi.Index(a)
    a[31] = [1.0, 1.0, 1.0]
    with capi.Index(a) as ix:
        idx, d2 = ix.nn1(b)
        oi, od = oracle.nn1_exhaustive(a, b)
        assert (idx == 31).all() and (idx == oi).all() and (_bits(d2) == _bits(od)).all()
        a2 = np.zeros((10, 3), dtype=np.float32)
        ix.set_input(a2)
        far = np.full((70, 3), 3e20, dtype=np.float32)
        far[1] = [1.0, 2.0, 3.0]
        idx, d2 = ix.nn1(far)
        oi, od = oracle.nn1_exhaustive(a2, far)
        assert (idx == oi).all() and (_bits(d2) == _bits(od)).all() and idx[0] == -1 and idx[1] == 0


@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
def test_ties_flann_without_any_tie_skips_the_tree_and_keeps_the_answer(gpu, engine):
    """PCC_TIES_FLANN on host data whose minima are all unique: the tied-query count is read back before the host tree is built
    and the call ends there -- same indices as the kd-tree walk (which can only differ at ties); a later query WITH ties on the
    same cloud then builds the tree"""
    a = synth.corridor_cloud(18381, synth.SEED_A)
    q = synth.corridor_cloud(5000, synth.SEED_B)
    fi, fd = oracle.KdTree(a).nn1_batch(q)
    li, _ = oracle.nn1_exhaustive(a, q)
    assert (fi == li).all()  # (no tie in this pair of clouds)
    with capi.Index(a, engine=engine) as ix:
        ix.set_tie_order(capi.TIES_FLANN)
        idx, d2 = ix.nn1(q)
        st = ix.stats()
        assert (idx == fi).all() and (_bits(d2) == _bits(fd)).all() and st[5] == 0 and st[6] == 0
        # midpoints of pairs of references: two references at exactly the same distance wherever the midpoint is exact
        mid = ((a[:2000].astype(np.float64) + a[1:2001].astype(np.float64)) / 2).astype(np.float32)
        fi2, fd2 = oracle.KdTree(a).nn1_batch(mid)
        idx2, d22 = ix.nn1(mid)
        assert (idx2 == fi2).all() and (_bits(d22) == _bits(fd2)).all()


@pytest.mark.parametrize("m,n", [(4, 4), (100, 100), (1000, 777), (4096, 2000)])
def test_small_calls_replay_flann_ties_on_the_device(gpu, m, n):
    """lattice descriptors (duplicates, exact ties everywhere) in FLANN's tie order through the small-call form: the search kernel
    flags the tied queries, the tree is built from the raw records the call still holds in pinned memory, one more launch walks
    the flagged queries -- against the kd-tree walk of the oracle; again on the same index (the tree is kept), under another split
    rule (it is rebuilt), and after a large host query has gone through the same pinned buffers (the records are gone: the
    replay of the separate launches takes over)"""
    rng = np.random.default_rng(m + 7)
    a = (rng.integers(0, 9, (m, 3)) * np.float32(0.125)).astype(np.float32)
    q = (rng.integers(0, 17, (n, 3)) * np.float32(0.0625)).astype(np.float32)
    if n >= 100:
        q[3] = np.nan
    ra, rq = _records(a, 32), _records(q, 32)
    fi, fd = oracle.KdTree(a).nn1_batch(q)
    li, ld = oracle.nn1_exhaustive(a, q)
    assert m < 100 or (fi != li).sum() > 5
    with capi.Index(ra) as ix:
        ix.set_tie_order(capi.TIES_FLANN)
        for _ in range(2):
            idx, d2 = ix.nn1(rq)
            st = ix.stats()
            assert (idx == fi).all(), np.nonzero(idx != fi)[0][:5]
            assert (_bits(d2) == _bits(fd)).all()
            assert st[6] == (fi != li).sum() and st[5] >= st[6]
        got = ix.match_knn(rq, 0.05)
        assert list(got) == [0] + [int(i) for i, d in zip(fi, fd) if i >= 0 and d < np.float32(0.05)]
        oracle.set_split_rule(1)
        try:
            f1, _ = oracle.KdTree(a).nn1_batch(q)
        finally:
            oracle.set_split_rule(0)
        ix.set_option(capi.OPT_FLANN_SPLIT, 1)
        assert (ix.nn1(rq)[0] == f1).all()
        ix.set_option(capi.OPT_FLANN_SPLIT, 0)
        assert (ix.nn1(rq)[0] == fi).all()
        if m == 1000:
            big = synth.corridor_cloud(800_000, synth.SEED_B)      # 9.6 MB: staged through the pinned chunk buffers
            bi, bd = ix.nn1(big)
            obi, obd = oracle.KdTree(a).nn1_batch(big)
            assert (bi == obi).all() and (_bits(bd) == _bits(obd)).all()
            ix.set_option(capi.OPT_FLANN_SPLIT, 1)                   # (a new tree is needed, and the raw records are gone)
            assert (ix.nn1(rq)[0] == f1).all()
            ix.set_input(ra)
            ix.set_option(capi.OPT_FLANN_SPLIT, 0)
            assert (ix.nn1(rq)[0] == fi).all()
