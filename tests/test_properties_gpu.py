"""Property tests (hypothesis) of the HIP path against the definitional oracle on small adversarial
clouds: lattice coordinates (exact ties, duplicates, flat and linear sets), queries far outside the
cloud, both engines.  Lowest-index tie-break on both sides => indices must agree exactly."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle
from pointcloudcomparator_amd import capi

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


coord = st.integers(min_value=-6, max_value=6).map(lambda v: np.float32(v) * np.float32(0.125))
cloud = st.lists(st.tuples(coord, coord, coord), min_size=1, max_size=400).map(lambda l: np.array(l, dtype=np.float32))
queries = st.lists(st.tuples(*([st.floats(-3.0, 3.0, width=32)] * 3)), min_size=1, max_size=80).map(
    lambda l: np.array(l, dtype=np.float32))
SET = dict(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


@settings(**SET)
@given(cloud, queries, st.sampled_from([capi.ENGINE_BRUTE, capi.ENGINE_GRID]))
def test_nn1_matches_definition(gpu, a, q, engine):
    with capi.Index(a, engine=engine) as ix:
        idx, d2 = ix.nn1(q)
    oi, od = oracle.nn1_exhaustive(a, q)
    assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()


@settings(**SET)
@given(cloud, queries, st.integers(1, 70))
def test_knn_matches_definition(gpu, a, q, k):
    with capi.Index(a) as ix:
        idx, d2 = ix.knn(q, k)
    oi, od = oracle.knn_exhaustive(a, q, k)
    assert (idx == oi).all() and (_bits(d2) == _bits(od)).all()


@settings(**SET)
@given(cloud, queries, st.sampled_from([0.125, 0.25, 0.5, 1.0]))
def test_radius_matches_definition(gpu, a, q, r):
    with capi.Index(a) as ix:
        offs, idx, d2 = ix.radius_search(q, r, sorted=True)
    cnt = oracle.radius_count_exhaustive(a, q, r)
    assert (np.diff(offs) == cnt).all()
    tree = oracle.KdTree(a)
    for j in range(len(q)):
        ri, rd = tree.radius(q[j], r)
        assert (idx[offs[j]:offs[j + 1]] == ri).all() and (_bits(d2[offs[j]:offs[j + 1]]) == _bits(rd)).all()


@settings(**SET)
@given(cloud, st.sampled_from([0.125, 0.2, 0.3]), st.integers(1, 5))
def test_clusters_match_oracle(gpu, a, tol, min_size):
    with capi.Index(a) as ix:
        labels, ncl, sizes = ix.euclidean_clusters(tol, min_size, 100000)
    ol, on, osz = oracle.euclidean_clusters(a, tol, min_size, 100000)
    assert ncl == on and (sizes == osz).all() and (labels == ol).all()
