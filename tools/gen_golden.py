#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (committed, small).

The reference repository holds no golden vectors for this path and PCL/FLANN
are not installed, so these come from the float32 DEFINITIONAL oracle (numpy exhaustive,
each ufunc rounding separately, lowest index on ties), cross-checked here against the C
restatement (exhaustive + FLANN kd-tree) and, for indices, against float64 scipy cKDTree.
Re-running this script must reproduce the committed files bit for bit.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402
from pointcloudcomparator_amd import synth  # noqa: E402

OUT = ROOT / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)


def main():
    # (1) k = 1: two 4096-point clouds
    a = synth.corridor_cloud(4096, synth.SEED_A)
    b = synth.corridor_cloud(4096, synth.SEED_B)
    idx, d2 = oracle.nn1_numpy(a, b)
    ci, cd = oracle.nn1_exhaustive(a, b)
    ki, kd_ = oracle.KdTree(a).nn1_batch(b)
    assert (idx == ci).all() and (d2.view(np.uint32) == cd.view(np.uint32)).all()
    assert (idx == ki).all() and (d2.view(np.uint32) == kd_.view(np.uint32)).all()
    from scipy.spatial import cKDTree
    _, si = cKDTree(a.astype(np.float64)).query(b.astype(np.float64))
    assert (si == idx).all()
    np.savez_compressed(OUT / "nn1_4096.npz", ref=a, qry=b, idx=idx, d2_bits=d2.view(np.uint32))

    # (2) k = 51 for 256 queries (SOR neighbourhood), radius counts at the EC tolerance
    kq = b[:256]
    ki, kd2 = oracle.knn_exhaustive(a, kq, 51)
    tree = oracle.KdTree(a)
    for j in range(256):
        ti, td = tree.knn(kq[j], 51)
        assert (td.view(np.uint32) == kd2[j].view(np.uint32)).all()
    rc = oracle.radius_count_exhaustive(a, b, 0.05)
    rc2 = oracle.radius_count_exhaustive(a, b, 0.25)
    np.savez_compressed(OUT / "knn51_radius.npz", knn_idx=ki, knn_d2_bits=kd2.view(np.uint32),
                        radius_005_counts=rc, radius_025_counts=rc2)

    # (3) clustered scene: 8192 points of the object layer + expected partition
    c = synth.corridor_cloud(8192, synth.SEED_A, layer="objects")
    # at 8192 points the balls are too sparse for tol 0.05; use a scene compressed into 8 balls
    centres = synth.ball_centres()[:8]
    i = np.arange(8192, dtype=np.uint64)
    u = [synth.uniform24(0x60D, i * np.uint64(8) + np.uint64(j)) for j in range(3)]
    ball = (synth.splitmix64(0x60E, i) % np.uint64(8)).astype(np.int64)
    r = 0.12 * np.cbrt(u[0]); ct = 2 * u[1] - 1; st = np.sqrt(np.maximum(0, 1 - ct * ct)); ph = 2 * np.pi * u[2]
    c = (centres[ball] + np.stack([r * st * np.cos(ph), r * st * np.sin(ph), r * ct], -1)).astype(np.float32)
    labels, ncl, sizes = oracle.euclidean_clusters(c, 0.05, 100, 250000)
    assert ncl == 8, ncl
    np.savez_compressed(OUT / "clusters_8192.npz", pts=c, labels=labels, sizes=sizes)

    # (4) ICP pair: 2048 source points vs 4096 targets, correspondences of 3 iterations
    tgt = a
    src = synth.rigid_offset(a[:2048], jitter=0.001)
    cur = src.copy()
    corr, mats = [], []
    for _ in range(3):
        ii, dd, sums = tree.icp_step_sums(tgt, cur)
        rc_, T = oracle.umeyama_from_sums(sums)
        assert rc_ == 0
        corr.append(ii)
        mats.append(T)
        cur = oracle.transform(T, cur)
    np.savez_compressed(OUT / "icp_2048.npz", src=src, tgt=tgt, corr=np.stack(corr), T=np.stack(mats), final=cur)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
