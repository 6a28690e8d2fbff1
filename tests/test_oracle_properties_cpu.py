"""Property tests (hypothesis) of the oracle: the FLANN kd-tree restatement against the exhaustive
definition on small adversarial clouds -- lattice coordinates force exact ties, duplicates,
collinear / coplanar sets and queries outside the root bounding box."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle


def _bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


coord = st.integers(min_value=-4, max_value=4).map(lambda v: np.float32(v) * np.float32(0.25))
point = st.tuples(coord, coord, coord)
cloud = st.lists(point, min_size=1, max_size=120).map(lambda l: np.array(l, dtype=np.float32))
queries = st.lists(st.tuples(*([st.floats(-2.0, 2.0, width=32)] * 3)), min_size=1, max_size=40).map(
    lambda l: np.array(l, dtype=np.float32))


@settings(max_examples=60, deadline=None)
@given(cloud, queries)
def test_kdtree_nn1_equals_exhaustive_up_to_ties(a, q):
    ei, ed = oracle.nn1_exhaustive(a, q)
    ki, kd = oracle.KdTree(a).nn1_batch(q)
    assert (_bits(ed) == _bits(kd)).all()
    for j in np.nonzero(ei != ki)[0]:  # FLANN: first visited; exhaustive: lowest index -- only among exact ties
        alt = oracle.nn1_exhaustive(a[ki[j]:ki[j] + 1], q[j:j + 1])[1]
        assert _bits(alt)[0] == _bits(ed)[j] and ei[j] < ki[j]


@settings(max_examples=40, deadline=None)
@given(cloud, queries, st.integers(1, 9))
def test_kdtree_knn_distances_equal_exhaustive(a, q, k):
    ei, ed = oracle.knn_exhaustive(a, q, k)
    tree = oracle.KdTree(a)
    for j in range(len(q)):
        ti, td = tree.knn(q[j], k)
        m = min(k, len(a))
        assert len(ti) == m and (np.diff(td) >= 0).all()
        assert (_bits(td) == _bits(ed[j, :m])).all()
        assert sorted(set(ti.tolist())) == sorted(ti.tolist())  # no point reported twice


@settings(max_examples=40, deadline=None)
@given(cloud, queries, st.sampled_from([0.25, 0.5, 0.75, 1.0]))
def test_radius_counts_and_sorted_order(a, q, r):
    cnt = oracle.radius_count_exhaustive(a, q, r)
    tree = oracle.KdTree(a)
    r2 = np.float32(np.float64(r) * np.float64(r))
    for j in range(len(q)):
        ri, rd = tree.radius(q[j], r)
        assert len(ri) == cnt[j] and (rd < r2).all()
        order = np.lexsort((ri, rd))
        assert (order == np.arange(len(ri))).all()  # sorted by (d2, index) like RadiusResultSet + std::sort


@settings(max_examples=25, deadline=None)
@given(cloud, st.sampled_from([0.25, 0.3, 0.5]), st.integers(1, 4))
def test_euclidean_clusters_are_the_connected_components(a, tol, min_size):
    labels, ncl, sizes = oracle.euclidean_clusters(a, tol, min_size, 10_000)
    r2 = np.float32(np.float64(np.float32(tol)) ** 2)
    d = ((a[:, None, :] - a[None]) ** 2).astype(np.float32)
    adj = ((d[..., 0] + d[..., 1]) + d[..., 2]) < r2
    comp = np.arange(len(a))
    changed = True
    while changed:  # label propagation to the lowest index
        new = np.where(adj, comp[None, :], len(a)).min(1)
        new = np.minimum(new, comp)
        changed = (new != comp).any()
        comp = new
    kept = {c for c in np.unique(comp) if (comp == c).sum() >= min_size}
    assert ncl == len(kept)
    for c in kept:
        ids = np.unique(labels[comp == c])
        assert len(ids) == 1 and ids[0] >= 0 and sizes[ids[0]] == (comp == c).sum()
    assert (labels[~np.isin(comp, list(kept))] == -1).all()
    assert (np.diff(sizes) <= 0).all()
