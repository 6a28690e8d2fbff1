import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle
from pointcloudcomparator_amd import capi
from test_normals_gpu import _room
for k in (3, 10, 50):
    pts = _room()
    ix = capi.Index(pts)
    got = ix.normals(k)
    nbr, _ = ix.knn(pts, k)
    want = oracle.normals(pts, k, neighbours=nbr)
    same = ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all(1)
    bad = np.nonzero(~same)[0]
    print("k", k, "same", same.mean(), "bad", len(bad))
    for i in bad[:8]:
        print(i, got[i], want[i], got[i].view(np.uint32) - want[i].view(np.uint32))
    if len(bad):
        d = np.abs((got[bad, :3] * want[bad, :3]).sum(1))
        print("min |dot|", d.min(), "max curv diff", np.abs(got[bad, 3] - want[bad, 3]).max())
