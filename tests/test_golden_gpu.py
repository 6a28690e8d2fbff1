"""GPU path against the committed golden fixtures (tests/golden, made by tools/gen_golden.py)."""
from pathlib import Path

import numpy as np
import pytest

from pointcloudcomparator_amd import capi

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def _bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("engine", [capi.ENGINE_BRUTE, capi.ENGINE_GRID])
def test_golden_nn1(gpu, engine):
    g = np.load(G / "nn1_4096.npz")
    with capi.Index(g["ref"], engine=engine) as ix:
        idx, d2 = ix.nn1(g["qry"])
    assert (idx == g["idx"]).all() and (_bits(d2) == g["d2_bits"]).all()


def test_golden_knn_radius(gpu):
    g = np.load(G / "knn51_radius.npz")
    n = np.load(G / "nn1_4096.npz")
    with capi.Index(n["ref"]) as ix:
        ki, kd = ix.knn(n["qry"][:256], 51)
        c005 = ix.radius_count(n["qry"], 0.05)
        c025 = ix.radius_count(n["qry"], 0.25)
    assert (ki == g["knn_idx"]).all() and (_bits(kd) == g["knn_d2_bits"]).all()
    assert (c005 == g["radius_005_counts"]).all() and (c025 == g["radius_025_counts"]).all()


def test_golden_clusters(gpu):
    g = np.load(G / "clusters_8192.npz")
    with capi.Index(g["pts"]) as ix:
        labels, ncl, sizes = ix.euclidean_clusters(0.05, 100, 250000)
    assert ncl == 8 and (sizes == g["sizes"]).all() and (labels == g["labels"]).all()


def test_golden_icp_iterations(gpu):
    g = np.load(G / "icp_2048.npz")
    with capi.Index(g["tgt"]) as ix:
        cur = g["src"].copy()
        for it in range(3):
            idx, d2, sums = ix.icp_step(cur)
            assert (idx == g["corr"][it]).all()       # identical inputs -> identical correspondences
            cur = ix.transform(g["T"][it], cur)       # golden transform keeps the inputs identical
        assert (_bits(cur) == _bits(g["final"])).all()


def test_golden_segmentation_rows(gpu):
    """the widened rows on the committed room scene: voxel grid, RANSAC plane, normals K=50,
    region growing K=30, keypoint snap"""
    g = np.load(G / "segmentation_6000.npz")
    room, vox = g["room"], g["voxels"]
    with capi.Index(room) as ix:
        got_vox = ix.voxel_grid(room, 0.025)
        fw = ix.first_within(vox[:200] + np.float32(0.01), 0.05)
    assert got_vox.shape == vox.shape
    np.testing.assert_allclose(got_vox, vox, rtol=0, atol=1e-5)  # PCL sums in float, the GPU in double
    assert (fw == g["first_within"]).all()
    with capi.Index(vox) as ix:
        inl, coeff, its = ix.sac_plane(vox, 100, 0.02, 0.99, True)
        assert its == int(g["sac_iterations"]) and (inl == g["sac_inliers"]).all()
        assert (_bits(coeff) == g["sac_coeff_bits"]).all()
        nrm = ix.normals(50)
        want = g["normals_bits"].view(np.float32)
        # every bit: the device evaluates glibc's atan2f / cosf / sinf (csrc/libm_f32.hpp), the fixture holds
        # what the oracle computed with the host's libm
        same = (_bits(nrm) == g["normals_bits"]).all(axis=1) | (np.isnan(nrm).all(axis=1) & np.isnan(want).all(axis=1))
        assert same.all(), int((~same).sum())
        labels, ncl = ix.region_growing(want, k=30, smoothness=3.0 / 180.0 * np.pi, curvature_threshold=1.0,
                                        min_size=50, max_size=1000000)
    assert ncl == int(g["rg_clusters"]) and (labels == g["rg_labels"]).all()
