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
