"""The oracle's restatement of pcl::RegionGrowingRGB (color_growing_segmentation, reference src/segmentation.cpp:161-216)
on hand-built scenes whose answer follows from the algorithm's definition -- no GPU."""
import numpy as np

import oracle


def _blob(rng, centre, n, spread=0.05):
    return (np.asarray(centre, np.float32) + rng.normal(0, spread, (n, 3))).astype(np.float32)


def test_two_colours_two_segments_and_a_small_one_is_folded_in():
    rng = np.random.default_rng(1)
    a, b = _blob(rng, (0, 0, 0), 300), _blob(rng, (0.6, 0, 0), 300)   # apart: no row mixes the colours
    pts = np.concatenate([a, b])
    rgb = np.concatenate([np.tile([200, 30, 30], (300, 1)), np.tile([30, 30, 200], (300, 1))]).astype(np.uint8)
    labels, ncl = oracle.region_growing_rgb(pts, rgb)
    assert ncl == 2 and (labels[:300] == labels[0]).all() and (labels[300:] == labels[300]).all() and labels[0] != labels[300]
    # the second blob with 60 points only: below the minimum of 200 it joins the region of its nearest neighbouring segment
    pts2, rgb2 = np.concatenate([a, b[:60]]), np.concatenate([rgb[:300], rgb[300:360]])
    labels, ncl = oracle.region_growing_rgb(pts2, rgb2)
    assert ncl == 1 and (labels == 0).all()
    # ... however far away it is: rows of 100 neighbours reach past a blob of 60 points, so the big segment IS its
    # neighbour, and the folding of small regions asks for no distance threshold
    far = np.concatenate([a, b[:60] + np.float32(50.0)])
    labels, ncl = oracle.region_growing_rgb(far, rgb2)
    assert ncl == 1 and (labels == 0).all()
    # a small region that stays small: both blobs below the minimum fold together and are dropped as one cluster of 120
    labels, ncl = oracle.region_growing_rgb(np.concatenate([a[:60], b[:60]]), np.concatenate([rgb[:60], rgb[300:360]]))
    assert ncl == 0 and (labels == -1).all()


def test_a_colour_ramp_chains_into_one_segment_and_similar_regions_merge():
    rng = np.random.default_rng(2)
    n = 600
    x = np.sort(rng.random(n)).astype(np.float32)
    pts = np.stack([x, rng.normal(0, 0.002, n), rng.normal(0, 0.002, n)], 1).astype(np.float32)
    ramp = np.clip((x * 250).astype(np.int32), 0, 255)
    rgb = np.stack([ramp, ramp, ramp], 1).astype(np.uint8)          # neighbours differ by a few grey levels: one chain
    labels, ncl = oracle.region_growing_rgb(pts, rgb)
    assert ncl == 1 and (labels == 0).all()
    # two blobs 3 grey levels apart per channel: point threshold 6 is exceeded (27 + ... no: 3^2 * 3 = 27 <= 36, they chain)
    a, b = _blob(rng, (0, 0, 0), 250, 0.03), _blob(rng, (0.1, 0, 0), 250, 0.03)
    rgb = np.concatenate([np.tile([100, 100, 100], (250, 1)), np.tile([103, 103, 103], (250, 1))]).astype(np.uint8)
    labels, ncl = oracle.region_growing_rgb(np.concatenate([a, b]), rgb)
    assert ncl == 1
    # 4 levels apart: 48 > 36, two segments; their mean colours differ by 48 >= 25: no merge either
    rgb[250:] = 104
    labels, ncl = oracle.region_growing_rgb(np.concatenate([a, b]), rgb)
    assert ncl == 2
    # 2 levels apart, two tight blobs of 80 points 0.3 apart: the 30 neighbours a point grows through are all of its own
    # blob (two segments), but its row of 100 reaches the other one -- the segments are neighbours, 0.09 apart in d2, and
    # the MERGE step joins them (colour difference 12 < 25, distance below 10^2)
    far_a, far_b = _blob(rng, (0, 0, 0), 80, 0.01), _blob(rng, (0.3, 0, 0), 80, 0.01)
    rgb = np.concatenate([np.tile([100, 100, 100], (80, 1)), np.tile([102, 102, 102], (80, 1))]).astype(np.uint8)
    labels, ncl = oracle.region_growing_rgb(np.concatenate([far_a, far_b]), rgb, min_size=50)
    assert ncl == 1 and (labels == 0).all()
    labels, ncl = oracle.region_growing_rgb(np.concatenate([far_a, far_b]), rgb, min_size=50, distance=0.2)
    assert ncl == 2, "with a distance threshold of 0.2 (0.04 in d2) the segments are too far apart to merge"
    rgb[80:] = 106                                                   # mean colours 6 apart per channel: 108 >= 25, no merge
    labels, ncl = oracle.region_growing_rgb(np.concatenate([far_a, far_b]), rgb, min_size=50)
    assert ncl == 2


def test_rows_given_equal_rows_searched_and_labels_partition_the_cloud():
    rng = np.random.default_rng(3)
    pts = rng.random((1500, 3)).astype(np.float32)
    rgb = rng.integers(0, 4, (1500, 3)).astype(np.uint8) * 20        # 64 colours 20 levels apart: many small segments
    l1, n1 = oracle.region_growing_rgb(pts, rgb, min_size=20)
    ki, kd = oracle.knn_exhaustive(pts, pts, 100)
    l2, n2 = oracle.region_growing_rgb(pts, rgb, neighbours=ki, neighbour_d2=kd, min_size=20)
    assert n1 == n2 and (l1 == l2).all()
    assert set(np.unique(l1)) <= set(range(-1, n1)) and all((l1 == c).sum() >= 20 for c in range(n1))


def big_segment_scene():
    """80 points of colour 250 (first in index order: segment 0) 0.4 from a tight blob of 72 001 points of colour 253.  The
    channel sums of the blob are 18.2M > 2^24: a float accumulation rounds (mean 252), PCL's unsigned sums do not (253).
    Against the small segment's 250 that is a colour difference of 3 * 2^2 = 12 < 25 (merge) or 3 * 3^2 = 27 (no merge)."""
    rng = np.random.default_rng(4)
    small = _blob(rng, (0.4, 0, 0), 80, 0.004)
    big = _blob(rng, (0, 0, 0), 72001, 0.03)
    pts = np.concatenate([small, big])
    rgb = np.concatenate([np.full((80, 3), 250), np.full((72001, 3), 253)]).astype(np.uint8)
    return pts, rgb


def test_segment_colours_are_summed_exactly_like_pcls_unsigned_vectors():
    pts, rgb = big_segment_scene()
    acc = np.cumsum(np.full(72001, 253, np.float32), dtype=np.float32)[-1]
    assert int(np.float32(acc) / np.float32(72001)) == 252, "the scene must be one a float sum gets wrong"
    labels, ncl = oracle.region_growing_rgb(pts, rgb, min_size=50)
    assert ncl == 2 and (labels[:80] == labels[0]).all() and (labels[80:] == labels[80]).all() and labels[0] != labels[80]
    # one grey level closer the exact means differ by 3 * 2^2 = 12 < 25 and the two do merge
    rgb[:80] = 251
    labels, ncl = oracle.region_growing_rgb(pts, rgb, min_size=50)
    assert ncl == 1 and (labels == 0).all()
