"""NormalEstimation (K neighbours) and RegionGrowing on the GPU against the oracle's restatements
(reference: src/segmentation.cpp:232-271 -- K = 50 normals, RegionGrowing with 100 neighbours, 3 degrees,
curvature threshold 1, cluster sizes 50..1000000)."""
import numpy as np
import pytest

import oracle
from pointcloudcomparator_amd import capi, synth

pytestmark = pytest.mark.gpu


def _room(n_per=3000, seed=3, noise=0.002):
    """three orthogonal walls + a sphere: planar regions with distinct normals and a curved blob"""
    rng = np.random.default_rng(seed)
    u, v = rng.random((2, n_per)).astype(np.float32) * 2.0
    floor = np.stack([u, v, np.zeros_like(u)], 1)
    u, v = rng.random((2, n_per)).astype(np.float32) * 2.0
    wall1 = np.stack([u, np.zeros_like(u), v], 1)
    u, v = rng.random((2, n_per)).astype(np.float32) * 2.0
    wall2 = np.stack([np.zeros_like(u), u, v], 1)
    d = rng.normal(size=(n_per, 3))
    ball = (d / np.linalg.norm(d, axis=1, keepdims=True) * 0.3 + np.array([1.0, 1.0, 1.0])).astype(np.float32)
    pts = np.concatenate([floor, wall1, wall2, ball]).astype(np.float32)
    pts += rng.normal(0, noise, pts.shape).astype(np.float32)
    pts += np.float32(3.0)  # away from the origin: PCL's single-pass float covariance is sensitive here
    return np.ascontiguousarray(pts[rng.permutation(len(pts))])


def _same_bits(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def _assert_normals_close(got, want, min_same=None):
    """Covariance, scaling, cross products and the flip are the same float operations on both sides, and so -- since
    round 5 -- are the three transcendental calls of the closed-form roots: csrc/libm_f32.hpp restates the host libm's
    atan2f / cosf / sinf (glibc 2.35) operation for operation, tests/test_libm_cpu.py pins it.  Every normal and every
    curvature carries the oracle's bits; rows that are NaN are NaN on both sides.  (Until round 5 the device rounded
    double results once and 2.3 % of all points differed in their last bits.)"""
    same = _same_bits(got, want).all(axis=1)
    assert same.all(), (int((~same).sum()), len(same), np.nonzero(~same)[0][:5])


@pytest.mark.parametrize("k", [3, 10, 50])
def test_normals_match_oracle_on_the_same_neighbourhoods(k):
    pts = _room()
    ix = capi.Index(pts)
    got = ix.normals(k)
    nbr, _ = ix.knn(pts, k)
    want = oracle.normals(pts, k, neighbours=nbr)
    _assert_normals_close(got, want)
    # unit length, flipped towards the origin
    ok = np.isfinite(got).all(axis=1)  # k = 3 leaves a few degenerate covariances (0/0), as in PCL
    assert ok.mean() > 0.99
    ln = np.linalg.norm(got[ok, :3].astype(np.float64), axis=1)
    np.testing.assert_allclose(ln, 1.0, atol=1e-5)
    assert ((got[ok, :3] * -pts[ok]).sum(1) >= -1e-6).all()


def test_normals_against_the_kdtree_oracle():
    """the full restatement (FLANN tree order) differs only where neighbours tie in distance"""
    pts = _room(1500, seed=5)
    ix = capi.Index(pts)
    got = ix.normals(50)
    want = oracle.normals(pts, 50)
    dots = np.abs((got[:, :3].astype(np.float64) * want[:, :3]).sum(1))
    assert (dots > 1 - 1e-6).mean() > 0.999
    assert np.median(np.abs(got[:, 3] - want[:, 3])) < 1e-7


def test_normals_viewpoint_and_device_output():
    import torch
    pts = _room(1000, seed=7)
    ix = capi.Index(pts)
    vp = np.array([10.0, 10.0, 10.0], np.float32)
    host = ix.normals(20, viewpoint=vp)
    dev = ix.normals(20, viewpoint=vp, device="cuda:0").cpu().numpy()
    assert _same_bits(host, dev).all()
    assert ((host[:, :3] * (vp - pts)).sum(1) >= -1e-6).all()
    want = oracle.normals(pts, 20, viewpoint=vp, neighbours=ix.knn(pts, 20)[0])
    _assert_normals_close(host, want)


def test_normals_edge_cases():
    # fewer than three points -> NaN (PCL: indices.size() < 3); k larger than the cloud is clamped
    two = np.array([[0, 0, 0], [1, 0, 0]], np.float32)
    assert np.isnan(capi.Index(two).normals(50)).all()
    pts = _room(200, seed=11)[:40].copy()
    pts[7] = np.nan
    ix = capi.Index(pts)
    got = ix.normals(100)
    assert np.isnan(got[7]).all() and np.isfinite(np.delete(got, 7, 0)).all()
    nbr, _ = ix.knn(pts, 100)
    want = oracle.normals(pts, 100, neighbours=nbr)
    _assert_normals_close(got, want, 0.8)
    with pytest.raises(capi.PccError):
        ix.normals(0)


@pytest.mark.parametrize("k,theta_deg,curv_thr,min_size", [(100, 3.0, 1.0, 50), (30, 6.0, 0.05, 20), (10, 3.0, 1.0, 1)])
def test_region_growing_matches_oracle(k, theta_deg, curv_thr, min_size):
    pts = _room()
    ix = capi.Index(pts)
    nrm = ix.normals(50)
    labels, ncl = ix.region_growing(nrm, k=k, smoothness=theta_deg / 180.0 * np.pi, curvature_threshold=curv_thr,
                                    min_size=min_size, max_size=1000000)
    nbr, _ = ix.knn(pts, k)
    want, want_n = oracle.region_growing(nrm, nbr, theta_deg / 180.0 * np.pi, curv_thr, min_size, 1000000)
    assert ncl == want_n
    np.testing.assert_array_equal(labels, want)
    if min_size == 50:
        # the three walls come out as large regions
        sizes = np.sort(np.bincount(labels[labels >= 0]))[::-1]
        assert ncl >= 3 and sizes[2] > 1500


def test_region_growing_device_normals_and_size_filter():
    import torch
    pts = _room(1500, seed=13)
    ix = capi.Index(pts)
    nrm = ix.normals(50, device="cuda:0")
    lab_d, n_d = ix.region_growing(nrm, k=40, min_size=100, max_size=1400)
    lab_h, n_h = ix.region_growing(nrm.cpu().numpy(), k=40, min_size=100, max_size=1400)
    assert n_d == n_h
    np.testing.assert_array_equal(lab_d.cpu().numpy(), lab_h)
    cnt = np.bincount(lab_h[lab_h >= 0], minlength=max(n_h, 1))
    assert n_h == 0 or ((cnt >= 100) & (cnt <= 1400)).all()


@pytest.mark.parametrize("seed,k,theta_deg,nan_frac,curv_thr", [
    (1, 8, 25.0, 0.0, 1.0), (2, 16, 35.0, 0.01, 1.0), (3, 5, 50.0, 0.0, 1.0), (4, 40, 15.0, 0.002, 1.0), (5, 3, 80.0, 0.05, 1.0),
    # points above the curvature threshold join a region without spreading -- unless they seed one themselves
    (6, 8, 40.0, 0.0, 0.15), (7, 12, 60.0, 0.01, 0.05), (8, 4, 85.0, 0.03, 0.25), (9, 20, 30.0, 0.0, 0.0), (10, 6, 89.0, 0.02, 0.29)])
def test_region_growing_order_free_form_equals_pcl_walk(seed, k, theta_deg, nan_frac, curv_thr):
    """Adversarial graphs for the GPU formulation (label = lowest-ranked ancestor): random normals make
    the smooth-edge graph sparse, strongly one-directional and full of small components, NaN normals
    accept every edge, duplicated curvatures exercise the rank tie-break.  The result must equal the
    sequential walk of the oracle, label for label."""
    rng = np.random.default_rng(seed)
    n = 20000
    pts = rng.random((n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    nrm = np.zeros((n, 4), np.float32)
    nrm[:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True)
    nrm[:, 3] = np.round(rng.random(n) * 0.3, 3)  # many equal curvatures
    bad = rng.random(n) < nan_frac
    nrm[bad] = np.nan
    ix = capi.Index(pts)
    labels, ncl = ix.region_growing(nrm, k=k, smoothness=theta_deg / 180.0 * np.pi, curvature_threshold=curv_thr,
                                    min_size=1, max_size=n)
    nbr, _ = ix.knn(pts, k)
    want, want_n = oracle.region_growing(nrm, nbr, theta_deg / 180.0 * np.pi, curv_thr, 1, n)
    assert ncl == want_n
    np.testing.assert_array_equal(labels, want)
    assert ncl > 10


@pytest.mark.parametrize("radius", [0.03, 0.06, 0.15])
def test_normals_with_radius_search(radius):
    """NormalEstimation::setRadiusSearch (the RIFT pipeline's normals, src/comparator.cpp:628-635): neighbourhood =
    sorted radius search result"""
    pts = _room(2500, seed=21)
    pts[11] = np.nan
    ix = capi.Index(pts)
    got = ix.normals_radius(radius)
    want = oracle.normals_radius(pts, radius)
    assert np.isnan(got[11]).all() and np.isnan(want[11]).all()
    # same rows (sorted by (d2, index); FLANN's tie order differs only where distances tie), same arithmetic
    both_nan = np.isnan(got).all(1) & np.isnan(want).all(1)
    assert (np.isnan(got).all(1) == np.isnan(want).all(1)).all()
    ok = ~both_nan
    same = (got[ok].view(np.uint32) == want[ok].view(np.uint32)).all(axis=1)
    assert same.mean() > 0.99                  # (all but the rows where two neighbours tie in distance)
    dots = np.abs((got[ok, :3].astype(np.float64) * want[ok, :3]).sum(1))
    assert (dots > 1 - 1e-5).all()
    np.testing.assert_allclose(got[ok, 3], want[ok, 3], rtol=0, atol=1e-6)
    if radius == 0.03:
        assert both_nan.sum() > 1  # sparse corners: fewer than 3 points within 3 cm
