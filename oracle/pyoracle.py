"""ctypes loader of oracle/pcc_oracle.c plus a numpy definitional oracle.

TEST INFRASTRUCTURE ONLY / PARITY UNPINNED (see oracle/pcc_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libpcc_oracle.so"


def build(force: bool = False) -> Path:
    src = _HERE / "pcc_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        _SO.parent.mkdir(exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
                               "-pthread", "-o", str(_SO), str(src), "-lm"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(str(build()))
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        L.orc_nn1_exhaustive.argtypes = [vp, sz, sz, vp, sz, sz, vp, vp]
        L.orc_nn1_exhaustive.restype = None
        L.orc_knn_exhaustive.argtypes = [vp, sz, sz, vp, sz, sz, i32, vp, vp]
        L.orc_radius_count_exhaustive.argtypes = [vp, sz, sz, vp, sz, sz, C.c_float, vp]
        L.orc_radius_count_exhaustive.restype = None
        L.orc_kdtree_build.argtypes = [vp, sz, sz]
        L.orc_kdtree_build.restype = vp
        L.orc_kdtree_free.argtypes = [vp]
        L.orc_kdtree_free.restype = None
        L.orc_kdtree_size.argtypes = [vp]
        L.orc_kdtree_size.restype = sz
        L.orc_kdtree_knn.argtypes = [vp, vp, i32, vp, vp]
        L.orc_kdtree_radius.argtypes = [vp, vp, C.c_float, i32, vp, vp, i32]
        L.orc_kdtree_nn1_batch.argtypes = [vp, vp, sz, sz, vp, vp]
        L.orc_kdtree_nn1_batch.restype = None
        L.orc_kdtree_nn1_batch_mt.argtypes = [vp, vp, sz, sz, vp, vp, i32]
        L.orc_kdtree_nn1_batch_mt.restype = None
        L.orc_region_growing_rgb.argtypes = [vp, sz, sz, vp, vp, vp, i32, C.c_float, C.c_float, C.c_float, i32, i32, i32, i32, vp]
        L.orc_region_growing_rgb.restype = i32
        L.orc_normals.argtypes = [vp, sz, sz, i32, vp, vp]
        L.orc_normals.restype = None
        L.orc_normals_radius.argtypes = [vp, sz, sz, C.c_double, vp, vp]
        L.orc_normals_radius.restype = None
        L.orc_normals_from_neighbours.argtypes = [vp, sz, sz, vp, i32, vp, vp]
        L.orc_normals_from_neighbours.restype = None
        L.orc_region_growing.argtypes = [sz, vp, vp, i32, C.c_float, C.c_float, i32, i32, vp]
        L.orc_sac_plane.argtypes = [vp, sz, sz, i32, C.c_double, C.c_double, i32, vp, vp, vp]
        L.orc_sac_plane.restype = C.c_long
        L.orc_mt19937_raw.argtypes = [C.c_uint32, vp, sz]
        L.orc_mt19937_raw.restype = None
        L.orc_voxel_grid.argtypes = [vp, sz, sz, C.c_float, i32, vp, sz]
        L.orc_voxel_grid.restype = C.c_long
        L.orc_first_within.argtypes = [vp, sz, sz, vp, sz, sz, C.c_double, vp]
        L.orc_first_within.restype = None
        L.orc_match_rift_knn.argtypes = [vp, sz, vp, sz, sz, vp]
        L.orc_euclidean_clusters.argtypes = [vp, sz, sz, C.c_float, C.c_uint32, C.c_uint32, vp, vp, i32]
        L.orc_sor.argtypes = [vp, sz, sz, i32, C.c_double, vp, vp, C.POINTER(C.c_double)]
        L.orc_sor.restype = sz
        L.orc_icp.argtypes = [vp, sz, sz, vp, sz, sz, i32, i32, vp, C.POINTER(C.c_double), vp, vp]
        L.orc_icp_step_sums.argtypes = [vp, vp, sz, vp, sz, sz, vp, vp, vp]
        L.orc_icp_step_sums.restype = None
        L.orc_umeyama_from_sums.argtypes = [vp, vp]
        L.orc_transform.argtypes = [vp, vp, sz, sz, vp]
        L.orc_transform.restype = None
        _lib = L
    return _lib


def _f32(a):
    a = np.asarray(a)
    assert a.dtype == np.float32 and a.ndim == 2 and a.shape[1] >= 3 and a.strides[1] == 4
    return a, a.ctypes.data, a.shape[0], (a.strides[0] if a.shape[0] > 1 else a.shape[1] * 4)


# ---- numpy definitional oracle (SURVEY 8c golden vectors (1)) ---------------------
def nn1_numpy(ref: np.ndarray, qry: np.ndarray, chunk: int = 256):
    """float32 exhaustive: d=((dx*dx)+dy*dy)+dz*dz with each ufunc rounding separately
    (same bits as unfused C); argmin = lowest index; non-finite refs skipped."""
    ref = np.asarray(ref, dtype=np.float32)[:, :3]
    qry = np.asarray(qry, dtype=np.float32)[:, :3]
    valid = np.isfinite(ref).all(1)
    vidx = np.nonzero(valid)[0].astype(np.int32)
    r = ref[valid]
    n = len(qry)
    idx = np.full(n, -1, dtype=np.int32)
    d2 = np.full(n, np.inf, dtype=np.float32)
    if len(r) == 0:
        return idx, d2
    for s in range(0, n, chunk):
        q = qry[s:s + chunk]
        dx = q[:, None, 0] - r[None, :, 0]
        d = dx * dx
        dy = q[:, None, 1] - r[None, :, 1]
        d = d + dy * dy
        dz = q[:, None, 2] - r[None, :, 2]
        d = d + dz * dz
        a = np.argmin(d, axis=1)  # first occurrence = lowest index
        dm = d[np.arange(len(q)), a]
        # FLANN's result set starts with worst = FLT_MAX and rejects dist >= worst: an overflowed distance is no neighbour
        ok = np.isfinite(q).all(1) & (dm < np.finfo(np.float32).max)
        idx[s:s + chunk] = np.where(ok, vidx[a], -1)
        d2[s:s + chunk] = np.where(ok, dm, np.inf)
    return idx, d2


# ---- C restatement wrappers ---------------------------------------------------------
def nn1_exhaustive(ref, qry):
    r, rp, m, rs = _f32(ref)
    q, qp, n, qs = _f32(qry)
    idx = np.empty(n, np.int32)
    d2 = np.empty(n, np.float32)
    lib().orc_nn1_exhaustive(rp, m, rs, qp, n, qs, idx.ctypes.data, d2.ctypes.data)
    return idx, d2


def knn_exhaustive(ref, qry, k):
    r, rp, m, rs = _f32(ref)
    q, qp, n, qs = _f32(qry)
    idx = np.empty((n, k), np.int32)
    d2 = np.empty((n, k), np.float32)
    lib().orc_knn_exhaustive(rp, m, rs, qp, n, qs, k, idx.ctypes.data, d2.ctypes.data)
    return idx, d2


def radius_count_exhaustive(ref, qry, radius):
    r, rp, m, rs = _f32(ref)
    q, qp, n, qs = _f32(qry)
    r2 = np.float32(np.float64(radius) * np.float64(radius))
    cnt = np.empty(n, np.int32)
    lib().orc_radius_count_exhaustive(rp, m, rs, qp, n, qs, r2, cnt.ctypes.data)
    return cnt


def set_split_rule(rule: int) -> None:
    """which FLANN split rule the trees built from now on use (0 middleSplit_, 1 middleSplit, 2 middleSplit_ with the
    loop variable): shapes the tree, hence tie order, never a distance"""
    lib().orc_set_split_rule(int(rule))


def get_split_rule() -> int:
    return int(lib().orc_get_split_rule())


class KdTree:
    """FLANN KDTreeSingleIndex restatement behind pcl::KdTreeFLANN semantics."""

    def __init__(self, pts):
        self._pts, p, m, s = _f32(pts)
        self._h = lib().orc_kdtree_build(p, m, s)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kdtree_free(self._h)
            self._h = None

    @property
    def size(self):
        return lib().orc_kdtree_size(self._h) if self._h else 0

    def knn(self, q, k):
        q = np.ascontiguousarray(q, dtype=np.float32)
        idx = np.empty(k, np.int32)
        d2 = np.empty(k, np.float32)
        c = lib().orc_kdtree_knn(self._h, q.ctypes.data, k, idx.ctypes.data, d2.ctypes.data)
        return idx[:c], d2[:c]

    def radius(self, q, radius, sorted=True, cap=4096):
        q = np.ascontiguousarray(q, dtype=np.float32)
        r2 = np.float32(np.float64(radius) * np.float64(radius))
        idx = np.empty(cap, np.int32)
        d2 = np.empty(cap, np.float32)
        c = lib().orc_kdtree_radius(self._h, q.ctypes.data, r2, int(sorted), idx.ctypes.data, d2.ctypes.data, cap)
        if c > cap:
            return self.radius(q, radius, sorted, cap=c)
        return idx[:c], d2[:c]

    def nn1_batch(self, qry, nthreads=1):
        q, qp, n, qs = _f32(qry)
        idx = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        if nthreads > 1:
            lib().orc_kdtree_nn1_batch_mt(self._h, qp, n, qs, idx.ctypes.data, d2.ctypes.data, nthreads)
        else:
            lib().orc_kdtree_nn1_batch(self._h, qp, n, qs, idx.ctypes.data, d2.ctypes.data)
        return idx, d2

    def icp_step_sums(self, tgt, src):
        t, tp, m, ts = _f32(tgt)
        s, sp, n, ss = _f32(src)
        idx = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        sums = np.zeros(17, np.float64)
        lib().orc_icp_step_sums(self._h, tp, ts, sp, n, ss, idx.ctypes.data, d2.ctypes.data, sums.ctypes.data)
        return idx, d2, sums


def normals(pts, k, viewpoint=(0.0, 0.0, 0.0), neighbours=None):
    a, ap, n, s1 = _f32(pts)
    vpt = np.asarray(viewpoint, np.float32)
    out = np.empty((n, 4), np.float32)
    if neighbours is None:
        lib().orc_normals(ap, n, s1, k, vpt.ctypes.data, out.ctypes.data)
    else:
        nb = np.ascontiguousarray(neighbours, dtype=np.int32)
        lib().orc_normals_from_neighbours(ap, n, s1, nb.ctypes.data, nb.shape[1], vpt.ctypes.data, out.ctypes.data)
    return out


def normals_radius(pts, radius, viewpoint=(0.0, 0.0, 0.0)):
    a, ap, n, s1 = _f32(pts)
    vpt = np.asarray(viewpoint, np.float32)
    out = np.empty((n, 4), np.float32)
    lib().orc_normals_radius(ap, n, s1, float(radius), vpt.ctypes.data, out.ctypes.data)
    return out


def region_growing(normals4, neighbours, smoothness, curvature_threshold, min_size, max_size):
    nm = np.ascontiguousarray(normals4, dtype=np.float32)
    nb = np.ascontiguousarray(neighbours, dtype=np.int32)
    labels = np.empty(len(nm), np.int32)
    ncl = lib().orc_region_growing(len(nm), nm.ctypes.data, nb.ctypes.data, nb.shape[1], np.float32(smoothness),
                                   np.float32(curvature_threshold), min_size, max_size, labels.ctypes.data)
    return labels, ncl


def region_growing_rgb(pts, rgb, neighbours=None, neighbour_d2=None, distance=10.0, point_colour=6.0, region_colour=5.0,
                       min_size=200, max_size=2**31 - 1, nn=30, region_nn=100):
    """(labels, number of clusters) of pcl::RegionGrowingRGB::extract as color_growing_segmentation sets it up; rows of
    neighbours (ascending, -1 padded) may be given, else they come from the oracle's kd-tree"""
    a, ap, n, s1 = _f32(pts)
    col = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
    labels = np.full(n, -1, np.int32)
    if neighbours is not None:
        nb = np.ascontiguousarray(neighbours, dtype=np.int32)
        nd = np.ascontiguousarray(neighbour_d2, dtype=np.float32)
        ncl = lib().orc_region_growing_rgb(ap, n, s1, col.ctypes.data, nb.ctypes.data, nd.ctypes.data, nb.shape[1], distance,
                                           point_colour, region_colour, min_size, max_size, nn, region_nn, labels.ctypes.data)
    else:
        ncl = lib().orc_region_growing_rgb(ap, n, s1, col.ctypes.data, None, None, 0, distance, point_colour, region_colour,
                                           min_size, max_size, nn, region_nn, labels.ctypes.data)
    return labels, ncl


def sac_plane(pts, max_iterations=100, threshold=0.02, probability=0.99, optimize=True):
    """(inlier indices, coefficients[4], iterations) of SACSegmentation(PLANE, RANSAC).segment"""
    a, ap, n, s1 = _f32(pts)
    if n == 0:
        return np.empty(0, np.int32), np.zeros(4, np.float32), 0
    inl = np.empty(max(n, 1), np.int32)
    coeff = np.zeros(4, np.float32)
    its = C.c_int(0)
    m = lib().orc_sac_plane(ap, n, s1, max_iterations, threshold, probability, int(optimize), inl.ctypes.data,
                            coeff.ctypes.data, C.byref(its))
    return inl[:m].copy(), coeff, its.value


def mt19937_raw(seed, n):
    out = np.empty(n, np.uint32)
    lib().orc_mt19937_raw(seed, out.ctypes.data, n)
    return out


def voxel_grid(pts, leaf, has_rgb=False):
    a, ap, m, s1 = _f32(pts)
    out = np.zeros_like(a)
    nv = lib().orc_voxel_grid(ap, m, s1, np.float32(leaf), int(has_rgb), out.ctypes.data, s1)
    return out[:max(nv, 0)], nv


def first_within(pts, qry, radius):
    a, ap, m, s1 = _f32(pts)
    b, bp, n, s2 = _f32(qry)
    idx = np.empty(n, np.int32)
    lib().orc_first_within(ap, m, s1, bp, n, s2, float(radius), idx.ctypes.data)
    return idx


def match_rift_knn(des1, des2):
    a, ap, n1, s1 = _f32(des1)
    b, bp, n2, s2 = _f32(des2)
    assert s1 == s2
    out = np.empty(n2 + 1, np.int32)
    c = lib().orc_match_rift_knn(ap, n1, bp, n2, s1, out.ctypes.data)
    return out[:c]


def euclidean_clusters(pts, tolerance, min_size, max_size, max_clusters=65536):
    a, p, m, s = _f32(pts)
    labels = np.empty(m, np.int32)
    sizes = np.zeros(max_clusters, np.int32)
    n = lib().orc_euclidean_clusters(p, m, s, np.float32(tolerance), min_size, max_size, labels.ctypes.data,
                                     sizes.ctypes.data, max_clusters)
    return labels, n, sizes[:min(n, max_clusters)]


def sor(pts, mean_k=50, stddev_mult=1.5):
    a, p, n, s = _f32(pts)
    md = np.empty(n, np.float32)
    inl = np.empty(n, np.uint8)
    thr = C.c_double(0)
    kept = lib().orc_sor(p, n, s, mean_k, stddev_mult, md.ctypes.data, inl.ctypes.data, C.byref(thr))
    return md, inl, thr.value, kept


def umeyama_from_sums(sums):
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    T = np.zeros(16, np.float32)
    rc = lib().orc_umeyama_from_sums(sums.ctypes.data, T.ctypes.data)
    return rc, T.reshape(4, 4)


def transform(T, src):
    s, sp, n, ss = _f32(src)
    Tm = np.ascontiguousarray(np.asarray(T, np.float32).reshape(16))
    out = np.empty((n, 3), np.float32)
    lib().orc_transform(Tm.ctypes.data, sp, n, ss, out.ctypes.data)
    return out


def icp(src, tgt, max_iter=20, fixed=False):
    s, sp, n, ss = _f32(src)
    t, tp, m, ts = _f32(tgt)
    T = np.zeros(16, np.float32)
    fit = C.c_double(0)
    corr = np.empty(n, np.int32)
    mse = np.zeros(max_iter, np.float64)
    it = lib().orc_icp(sp, n, ss, tp, m, ts, max_iter, int(fixed), T.ctypes.data, C.byref(fit), corr.ctypes.data,
                       mse.ctypes.data)
    return T.reshape(4, 4), fit.value, it, corr, mse[:it]
