"""The example CLI (examples/comparator_main.cpp: the reference's call sites over the pcc:: shim) end to end on the GPU:
flags, banner lines, section strings and the numbers behind them (checked against the oracle)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle
from ply_util import write_ply
from pointcloudcomparator_amd import synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "build" / "comparator"


def _scene(seed, shift=(0.0, 0.0, 0.0)):
    """A floor and four boxes of points on a 0.03 lattice with +-0.002 jitter: no two points share a 0.025
    voxel, so the VoxelGrid output is the input (centroids of single points are exact in any summation
    order) and everything downstream -- RANSAC samples included -- can be compared exactly."""
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(np.arange(60), np.arange(60), indexing="ij"), -1).reshape(-1, 2) * 0.03
    floor = np.concatenate([g, np.zeros((len(g), 1))], 1)
    c = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(-1, 3) * 0.03
    boxes = [c + o for o in [(0.3, 0.3, 0.3), (1.2, 0.3, 0.45), (0.3, 1.2, 0.6), (1.2, 1.2, 0.3)]]
    pts = np.concatenate([floor] + boxes)
    pts = pts + rng.uniform(-0.002, 0.002, pts.shape) + 0.01 + np.asarray(shift)
    return np.ascontiguousarray(pts[rng.permutation(len(pts))].astype(np.float32))


def _euclidean_path(cloud):
    """the reference's -e path on the oracle: VoxelGrid -> plane removal loop -> clusters"""
    lines = []
    vox, nv = oracle.voxel_grid(np.ascontiguousarray(cloud), 0.025)
    vox = np.ascontiguousarray(vox[:, :3])
    lines.append(f"PointCloud after filtering has: {nv} data points.")
    n0 = len(vox)
    while len(vox) > 0.3 * n0:
        inl, _, _ = oracle.sac_plane(vox, 100, 0.02, 0.99, True)
        if len(inl) == 0:
            lines.append("Could not estimate a planar model for the given dataset.")
            break
        lines.append(f"PointCloud representing the planar component: {len(inl)} data points.")
        vox = np.ascontiguousarray(np.delete(vox, inl, 0))
    _, ncl, sizes = oracle.euclidean_clusters(vox, 0.05, 100, 250000)
    lines += [f"PointCloud representing the Cluster: {s} data points." for s in sizes]
    return lines, ncl


def test_cli_icp_clusters_noise(gpu, tmp_path):
    if not EXE.exists():
        subprocess.check_call(["make", "cli"], cwd=ROOT)
    a = _scene(1)
    b = _scene(2, shift=(0.004, -0.003, 0.002))
    b[5, 0] = np.nan  # stripped by removeNaNFromPointCloud
    fa, fb, res = tmp_path / "a.ply", tmp_path / "b.ply", tmp_path / "results.txt"
    write_ply(fa, a, fmt="binary")
    write_ply(fb, b, fmt="ascii")
    r = subprocess.run([str(EXE), "-i", "-e", "-n", str(fa), str(fb), "--results", str(res)], capture_output=True,
                       text=True, timeout=300)
    out = r.stdout
    assert r.returncode == 1  # the reference always returns 1 (src/comparator.cpp:1704)
    for line in ["Visualization of clusters is off.", "Noise analysis is on.", "ICP matching pre-comparison is on.",
                 "Euclidean cluster segmentation was selected as main segmentation algorithm.", "has converged:1",
                 "ICP has converged; starting comparison of point clouds",
                 "ICP has converged. Point clouds segmentation is as follows",
                 "Both pcl have the same percentage of noisy points: 0"]:
        assert line in out, line
    bf = np.ascontiguousarray(b[np.isfinite(b).all(1)])
    ncls = []
    pos = 0
    for cloud in (a, bf):
        lines, ncl = _euclidean_path(cloud)
        ncls.append(ncl)
        for line in lines:  # same lines, same order
            pos = out.index(line, pos) + 1
        assert any("planar component" in l for l in lines)
    assert out.count("PointCloud representing the Cluster:") == sum(ncls) == 8
    kept = [oracle.sor(c, 50, 1.5)[3] for c in (a, bf)]
    assert f"Noise pass removed {len(a) - kept[0]} / {len(bf) - kept[1]} points" in out
    txt = res.read_text()
    assert txt.startswith(f"Results of comparison between {fa} and {fb}\n" + "-" * 80)
    assert f"Number of points of PCL 1: {len(a)}\n" in txt and f"Number of points of PCL 2: {len(bf)}\n" in txt
    assert f"Number of clusters of PCL 1: {ncls[0]}\n" in txt and f"Number of clusters of PCL 2: {ncls[1]}\n" in txt
    assert "----------------------------------------\n Noise analysis: \n" in txt


def _walls(n_side, seed):
    """Three mutually perpendicular plates (floor z = 0, walls y = 1 and x = 1) sampled on a 0.03 lattice with
    +-0.002 in-plane jitter and +-0.001 of depth noise.  As in _scene() no two points share a 0.025 voxel, so the
    VoxelGrid output is the input cloud exactly and the rest of the default path can be pinned without tolerance;
    the plates are 0.28 m apart, further than a K = 100 neighbourhood reaches, so no row mixes two plates."""
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(np.arange(n_side), np.arange(n_side), indexing="ij"), -1).reshape(-1, 2) * 0.03
    n = len(g)
    jit = lambda: rng.uniform(-0.002, 0.002, (n, 2))
    dep = lambda: rng.uniform(-0.001, 0.001, n)
    a, b, c = g + jit(), g + jit(), g + jit()
    floor = np.stack([a[:, 0], a[:, 1], dep()], 1)
    wall1 = np.stack([b[:, 0], 1.0 + dep(), b[:, 1] + 0.2], 1)
    wall2 = np.stack([1.0 + dep(), c[:, 0], c[:, 1] + 0.2], 1)
    pts = np.concatenate([floor, wall1, wall2]) + 1.0
    return np.ascontiguousarray(pts[rng.permutation(len(pts))].astype(np.float32))


def test_cli_region_growing_default(gpu, tmp_path):
    """no -e: VoxelGrid 0.025 -> normals K=50 -> RegionGrowing (src/segmentation.cpp:218-327)"""
    import re
    a, b = _walls(27, 3), _walls(24, 4)
    fa, fb, res = tmp_path / "a.ply", tmp_path / "b.ply", tmp_path / "results.txt"
    write_ply(fa, a, fmt="binary")
    write_ply(fb, b, fmt="binary")
    r = subprocess.run([str(EXE), str(fa), str(fb), "--results", str(res)], capture_output=True, text=True, timeout=300)
    out = r.stdout
    assert r.returncode == 1
    assert "Region growing segmentation was selected (default) as main segmentation algorithm." in out
    got = [int(x) for x in re.findall(r"Number of clusters is equal to (\d+)", out)]
    assert len(got) == 2
    txt = res.read_text()
    for i, cloud in enumerate((a, b)):
        vox, nv = oracle.voxel_grid(cloud, 0.025)
        assert nv == len(cloud)  # one point per voxel: the centroids are the points themselves
        assert f"PointCloud after filtering has: {nv} data points." in out
        vox = np.ascontiguousarray(vox[:, :3])
        nrm = oracle.normals(vox, 50)
        nbr, _ = oracle.knn_exhaustive(vox, vox, 100)
        lab, ncl = oracle.region_growing(nrm, nbr, 3.0 / 180.0 * np.pi, 1.0, 50, 1000000)
        assert ncl == 3 and got[i] == ncl, (got[i], ncl)  # the three plates, exactly
        assert f"Number of clusters of PCL {i + 1}: {ncl}\n" in txt


def test_cli_missing_file_and_usage(gpu, tmp_path):
    r = subprocess.run([str(EXE), "-e", str(tmp_path / "x.ply"), str(tmp_path / "y.ply")], capture_output=True, text=True)
    assert "Was not able to open file" in r.stderr and r.returncode == 1
    r = subprocess.run([str(EXE), "-h"], capture_output=True, text=True)
    assert "Usage: [options] </pathToScene1.ply> </pathToScene2.ply>" in r.stdout and r.returncode == 1


# ---- the cluster sections of results.txt (reference src/comparator.cpp:1220-1255, 1257-1384, 1386-1518, 1570-1635) -----
def _read_cluster_ply(path):
    raw = Path(path).read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.decode().splitlines() if l.startswith("element vertex")][0].split()[-1])
    rec = np.frombuffer(body, dtype=np.dtype([("p", "<f4", 3), ("c", "u1", 3)]), count=n)
    return np.ascontiguousarray(rec["p"])


def _centroid_f32(pts):
    """float accumulators in point order, one division at the end (reference :1238-1250)"""
    s = np.add.accumulate(pts.astype(np.float32), axis=0, dtype=np.float32)[-1]
    return (s / np.float32(len(pts))).astype(np.float32)


def _g(x):
    return "%g" % float(x)  # std::ostream's default float formatting


def _write_descriptors(path, per_cluster):
    with open(path, "w") as f:
        f.write("pcc_descriptors 1\n")
        for j, d in per_cluster.items():
            f.write(f"cluster {j} {len(d)}\n")
            for row in d:
                f.write(" ".join(repr(float(v)) for v in row) + "\n")


def test_cli_cluster_sections_matches_scores_verdict(gpu, tmp_path):
    """-e run with precomputed descriptor files: the per-cluster sections, the matching of clusters through
    matchRIFTFeaturesKnn (GPU k=1 search on the descriptors), the score block and the 0/1/2 verdict."""
    a, b = _scene(1), _scene(2, shift=(0.004, -0.003, 0.002))
    fa, fb, res = tmp_path / "a.ply", tmp_path / "b.ply", tmp_path / "results.txt"
    write_ply(fa, a, fmt="binary")
    write_ply(fb, b, fmt="binary")
    # run 1: no descriptors -> clusters dumped, no verdict
    r = subprocess.run([str(EXE), "-e", str(fa), str(fb), "--results", str(res), "--dump-clusters", str(tmp_path / "cl")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 1
    assert "no verdict" in r.stdout and "same information" not in r.stdout
    txt = res.read_text()
    assert "(no descriptor files given: clusters could not be matched, no verdict)" in txt
    cl = [[_read_cluster_ply(tmp_path / f"cl_{k}_{j}.ply") for j in range(4)] for k in (1, 2)]
    assert all(len(c) == 216 for k in range(2) for c in cl[k])
    cen = [[_centroid_f32(c) for c in cl[k]] for k in range(2)]
    # which cluster of scene 2 sits where cluster i of scene 1 sits
    twin = [int(np.argmin([np.linalg.norm(cen[0][i] - c2) for c2 in cen[1]])) for i in range(4)]
    rng = np.random.default_rng(9)
    des1, des2 = {}, {}
    mk = lambda n: rng.random((n, 32)).astype(np.float32)
    d = mk(10); des1[0] = d; des2[twin[0]] = d.copy()                       # 10 / 10, all match: 11 / 10 -> accepted
    d = mk(12); des1[1] = d; des2[twin[1]] = d[:10].copy()                  # 12 / 10: 11 / 12 -> 0, rejected
    des1[2] = mk(3); des2[twin[2]] = mk(9)                                  # 3 descriptors: never tried
    d = mk(8); des1[3] = d; des2[twin[3]] = np.concatenate([d, d[:1]])      # 8 / 9, all 9 match: 10 / 9 -> accepted
    _write_descriptors(tmp_path / "d1.txt", des1)
    _write_descriptors(tmp_path / "d2.txt", des2)
    r = subprocess.run([str(EXE), "-e", str(fa), str(fb), "--results", str(res), "--descriptors1", str(tmp_path / "d1.txt"),
                        "--descriptors2", str(tmp_path / "d2.txt")], capture_output=True, text=True, timeout=300)
    out, txt = r.stdout, res.read_text()
    assert r.returncode == 1
    rule = "\n" + "-" * 36 + "\n"
    # per-cluster sections, PCL2 first
    want = rule + "Information of clusters of PCL2:\n" + "-" * 36 + "\n"
    for j in range(4):
        c = cen[1][j]
        want += (f"PCL2 cluster {j}:\n\tNumber of points: 216\n\tNumber of descriptors: {len(des2[j])}\n"
                 f"\tCoordinates of centroid: [{_g(c[0])},{_g(c[1])},{_g(c[2])}]\n")
    want += rule + "Information of clusters of PCL 1:\n" + "-" * 36 + "\n"
    for i in range(4):
        c = cen[0][i]
        want += (f"PCL1 cluster {i}:\n\tNumber of points: 216\n\tNumber of descriptors: {len(des1[i])}\n"
                 f"\tCoordinates of centroid: [{_g(c[0])},{_g(c[1])},{_g(c[2])}]\n")
    assert want in txt
    # matches: clusters 0 and 3
    plus = "      " + "+" * 58 + "\t\n"
    # (color_growing_segmentation of a matched pair: one-coloured clusters of 216 points are one colour segment each)
    colour = "\t\tSegment of PCL 1 and segment of PCL 2 have the same number of elements based on color differences: 1\n"
    m = rule + "Information of matches of clusters of PCL 1 and PCL 2:\n" + "-" * 36 + "\n"
    m += (f"\tMatched cluster 0 of PCL 1 with cluster {twin[0]} of PCL 2:\n\t\tBoth segments have the same number of points: 216\n"
          "\t\tBoth segments have the same number of descriptors: 10\n" + colour + plus)
    m += "\t\tCluster 1 of PCL 1 has no match in PCL 2\n" + plus
    m += "\t\tCluster 2 of PCL 1 has no match in PCL 2\n" + plus
    m += (f"\tMatched cluster 3 of PCL 1 with cluster {twin[3]} of PCL 2:\n\t\tBoth segments have the same number of points: 216\n"
          "\t\tSegment of PCL 2 has more descriptors: 9 over: 8\n" + colour + plus)
    m += "Total number of matches found: 2\n\n"
    assert m in txt
    # scores, ratios, verdict: points tie (432 / 432), descriptors 18 vs 19 -> the second cloud wins
    s = ("\n" + "-" * 28 + "\n\npoints score pcl1: 432\npoints score pcl2: 432\n\ndescriptors score pcl1: 18\n"
         "descriptors score pcl2: 19\n\ncolor elements score pcl1: 2\ncolor elements score pcl2: 2\n\n" + "-" * 28 + "\n\n")
    ratio = (432 / 432 + 18 / 19 + 2 / 2) / 3
    s += f"Ratio of similarity over the 2 matches: {_g(ratio)}\nRatio of general similarity of pcl 1 over pcl 2: {_g(ratio * (2 / 4))}\n"
    assert txt.endswith(s)
    assert f"Percentage of RIFT correspondences of clusters 0 and {twin[0]} is: 100" in out
    assert f"Percentage of RIFT correspondences of clusters 1 and {twin[1]} is: 0" in out
    assert out.count("Match accepted") == 0 and out.count("No match") == 2  # (the reference prints "Match accepted" only when des1 > des2)
    assert "The second point cloud has more information" in out
    # --gpus 2 (replicas on the devices present; one here): the same report
    res2 = tmp_path / "results2.txt"
    r2 = subprocess.run([str(EXE), "-e", "-n", "--gpus", "2", str(fa), str(fb), "--results", str(res2), "--descriptors1",
                         str(tmp_path / "d1.txt"), "--descriptors2", str(tmp_path / "d2.txt")], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 1 and "The second point cloud has more information" in r2.stdout
    t2 = res2.read_text()
    assert want in t2 and m in t2 and t2.endswith(s) and " Noise analysis: \n" in t2
