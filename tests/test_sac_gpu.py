"""RANSAC plane segmentation (pcc_sac_plane) against the oracle's restatement of pcl::SACSegmentation
(reference: src/segmentation.cpp:79-99 -- plane model, RANSAC, optimize on, 100 iterations, threshold 0.02)."""
import numpy as np
import pytest

import oracle
from pointcloudcomparator_amd import capi

pytestmark = pytest.mark.gpu


def _scene(n_plane, n_clutter, seed, noise=0.005, tilt=(0.1, -0.2, 1.0), offset=0.5):
    rng = np.random.default_rng(seed)
    nrm = np.asarray(tilt, np.float64)
    nrm /= np.linalg.norm(nrm)
    e1 = np.cross(nrm, [1.0, 0, 0] if abs(nrm[0]) < 0.9 else [0, 1.0, 0]); e1 /= np.linalg.norm(e1)
    e2 = np.cross(nrm, e1)
    uv = rng.random((n_plane, 2)) * 3
    plane = uv[:, :1] * e1 + uv[:, 1:] * e2 + offset * nrm + rng.normal(0, noise, (n_plane, 1)) * nrm
    clutter = rng.random((n_clutter, 3)) * 3 - 0.5
    pts = np.concatenate([plane, clutter]).astype(np.float32)
    return np.ascontiguousarray(pts[rng.permutation(len(pts))])


def _ctx():
    return capi.Index(np.zeros((1, 3), np.float32))


@pytest.mark.parametrize("n_plane,n_clutter,seed,optimize", [(20000, 10000, 0, True), (20000, 10000, 1, False),
                                                             (3000, 30000, 2, True), (50000, 500, 3, True),
                                                             (40, 10, 4, True)])
def test_sac_plane_matches_oracle(n_plane, n_clutter, seed, optimize):
    pts = _scene(n_plane, n_clutter, seed)
    inl, coeff, its = _ctx().sac_plane(pts, 100, 0.02, 0.99, optimize)
    want_inl, want_c, want_its = oracle.sac_plane(pts, 100, 0.02, 0.99, optimize)
    assert its == want_its                      # same samples, same counts, same stopping point
    assert (coeff.view(np.uint32) == want_c.view(np.uint32)).all(), (coeff, want_c)
    np.testing.assert_array_equal(inl, want_inl)
    assert (np.diff(inl) > 0).all()
    # the plane is found
    if n_plane >= n_clutter:
        assert len(inl) > 0.9 * n_plane
        d = np.abs(pts[inl].astype(np.float64) @ coeff[:3].astype(np.float64) + coeff[3])
        assert (d < 0.02 + 1e-6).all()


def test_sac_plane_many_iterations_and_strides():
    # weak support (10 % inliers) needs more than one batch of 32 candidates; 32-byte point stride
    pts = _scene(2000, 18000, 7)
    wide = np.zeros((len(pts), 8), np.float32)
    wide[:, :3] = pts
    inl, coeff, its = _ctx().sac_plane(wide, 100, 0.02, 0.99, True)
    want_inl, want_c, want_its = oracle.sac_plane(wide, 100, 0.02, 0.99, True)
    assert its == want_its and its > 32
    np.testing.assert_array_equal(inl, want_inl)
    assert (coeff.view(np.uint32) == want_c.view(np.uint32)).all()
    # fewer iterations allowed than RANSAC would like: stops at max_iterations + 1, as PCL does
    inl, coeff, its = _ctx().sac_plane(pts, 5, 0.02, 0.99, False)
    want_inl, want_c, want_its = oracle.sac_plane(pts, 5, 0.02, 0.99, False)
    assert its == want_its == 6
    np.testing.assert_array_equal(inl, want_inl)


def test_sac_plane_device_memory_and_edge_cases():
    import torch
    pts = _scene(8000, 4000, 9)
    pts[5] = np.nan
    pts[77, 1] = np.inf
    d = torch.from_numpy(pts).cuda()
    ctx = _ctx()
    inl_d, c_d, its_d = ctx.sac_plane(d)
    inl_h, c_h, its_h = ctx.sac_plane(pts)
    want_inl, want_c, want_its = oracle.sac_plane(pts)
    np.testing.assert_array_equal(inl_d.cpu().numpy(), want_inl)
    np.testing.assert_array_equal(inl_h, want_inl)
    assert its_d == its_h == want_its
    assert (c_d.view(np.uint32) == want_c.view(np.uint32)).all() and (c_h.view(np.uint32) == want_c.view(np.uint32)).all()
    assert 5 not in inl_h and 77 not in inl_h
    # fewer than three points: no model
    inl, c, its = ctx.sac_plane(pts[:2])
    assert len(inl) == 0 and its == 0 and (c == 0).all()
    inl, c, its = ctx.sac_plane(np.zeros((0, 3), np.float32))
    assert len(inl) == 0
    # all points identical: (p1-p0)/(p2-p0) is 0/0 = NaN on every lane, which PCL's equality test does NOT flag
    # as collinear; the plane normalises to NaN, supports nothing, and RANSAC runs out its iterations
    same = np.ones((50, 3), np.float32)
    inl, c, its = ctx.sac_plane(same)
    w_inl, w_c, w_its = oracle.sac_plane(same)
    assert len(inl) == len(w_inl) == 0 and its == w_its == 101
    # two distinct points repeated: differences are 0 or +-d, ratios equal on all lanes -> degenerate samples only
    two = np.tile(np.array([[0, 0, 0], [1, 1, 1]], np.float32), (20, 1))
    inl, c, its = ctx.sac_plane(two)
    w_inl, w_c, w_its = oracle.sac_plane(two)
    assert len(inl) == len(w_inl) and its == w_its
    # exactly coplanar lattice: threshold 0 keeps nothing (strict <), tiny threshold keeps all
    g = np.stack(np.meshgrid(np.arange(10.0), np.arange(10.0)), -1).reshape(-1, 2)
    flat = np.concatenate([g, np.full((100, 1), 2.0)], 1).astype(np.float32)
    inl, c, its = ctx.sac_plane(flat, 100, 1e-6, 0.99, False)
    w_inl, w_c, w_its = oracle.sac_plane(flat, 100, 1e-6, 0.99, False)
    np.testing.assert_array_equal(inl, w_inl)
    assert len(inl) == 100 and its == w_its


def test_plane_removal_loop_like_the_reference():
    """src/segmentation.cpp:88-117: remove planes until <= 30 % of the points remain"""
    rng = np.random.default_rng(11)
    a = _scene(15000, 0, 1, tilt=(0, 0, 1), offset=0.0)
    b = _scene(12000, 0, 2, tilt=(1, 0, 0), offset=-0.5)
    c = (rng.random((6000, 3)) * 0.5 + 1.0).astype(np.float32)
    cloud = np.ascontiguousarray(np.concatenate([a, b, c])[rng.permutation(33000)])
    ctx = _ctx()
    got, want = cloud.copy(), cloud.copy()
    n0 = len(cloud)
    rounds = 0
    while len(got) > 0.3 * n0:
        inl, coeff, _ = ctx.sac_plane(got)
        w_inl, w_c, _ = oracle.sac_plane(want)
        np.testing.assert_array_equal(inl, w_inl)
        assert (coeff.view(np.uint32) == w_c.view(np.uint32)).all()
        if len(inl) == 0:
            break
        got = np.ascontiguousarray(np.delete(got, inl, 0))
        want = np.ascontiguousarray(np.delete(want, w_inl, 0))
        rounds += 1
    assert rounds == 2 and len(got) < 0.3 * n0


def test_sac_plane_in_device_memory_gathers_its_points():
    """a cloud in HBM: the sampled points of a batch and the inliers of the refit are gathered on the device (no copy of the
    cloud to the host); a degenerate sample -- PCL redraws it at once -- cannot be replayed that way and sends the call
    through a host copy.  Both routes: the oracle's samples, counts, inliers and coefficient bits."""
    import torch
    ctx = _ctx()
    # weak support: more than one batch of candidates, refit on
    pts = _scene(2000, 18000, 7)
    inl, c, its = ctx.sac_plane(torch.from_numpy(pts).cuda(), 100, 0.02, 0.99, True)
    w_inl, w_c, w_its = oracle.sac_plane(pts, 100, 0.02, 0.99, True)
    assert its == w_its and its > 32
    np.testing.assert_array_equal(inl.cpu().numpy(), w_inl)
    assert (c.view(np.uint32) == w_c.view(np.uint32)).all()
    # 32-byte stride in device memory, non-finite points among the samples' candidates
    wide = np.zeros((len(pts), 8), np.float32)
    wide[:, :3] = pts
    wide[::7, 2] = np.nan
    inl, c, its = ctx.sac_plane(torch.from_numpy(wide).cuda(), 100, 0.02, 0.99, True)
    w_inl, w_c, w_its = oracle.sac_plane(wide, 100, 0.02, 0.99, True)
    assert its == w_its
    np.testing.assert_array_equal(inl.cpu().numpy(), w_inl)
    assert (c.view(np.uint32) == w_c.view(np.uint32)).all()
    # duplicates: degenerate samples are drawn (p1 == p0 gives equal ratios 0) -> the retry with a host copy
    rng = np.random.default_rng(3)
    base = _scene(300, 100, 5)
    dup = np.ascontiguousarray(base[rng.integers(0, len(base), 4000)])
    inl, c, its = ctx.sac_plane(torch.from_numpy(dup).cuda(), 100, 0.02, 0.99, True)
    w_inl, w_c, w_its = oracle.sac_plane(dup, 100, 0.02, 0.99, True)
    assert its == w_its
    np.testing.assert_array_equal(inl.cpu().numpy(), w_inl)
    assert (c.view(np.uint32) == w_c.view(np.uint32)).all()
    two = np.tile(np.array([[0, 0, 0], [1, 1, 1]], np.float32), (20, 1))
    inl, c, its = ctx.sac_plane(torch.from_numpy(two).cuda())
    w_inl, w_c, w_its = oracle.sac_plane(two)
    assert len(inl) == len(w_inl) and its == w_its
