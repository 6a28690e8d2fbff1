"""ctypes binding of the libpcc_nn C-ABI (include/pcc_nn.h).

Thin by design: every method maps 1:1 onto one exported function; numpy arrays
are passed as host pointers, torch CUDA tensors as device pointers.  No search
is ever computed in Python -- a missing library is a hard error.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

MEM_HOST, MEM_DEVICE = 0, 1
TIES_LOWEST_INDEX, TIES_FLANN = 0, 1
ENGINE_AUTO, ENGINE_BRUTE, ENGINE_GRID = 0, 1, 2
KNN_MAX_K = 65536
# enum pcc_option
(OPT_GRID_PPC, OPT_GRID_TRIM, OPT_FAR_MODE, OPT_ICP_WARM, OPT_ICP_DEVICE_LOOP, OPT_EC_CELLS, OPT_SORT_MP_MIN,
 OPT_SORT_MP_MIN_Q, OPT_NN1_KERNEL, OPT_FLANN_SPLIT, OPT_NN1_DENSE_MIN, OPT_KNN_KERNEL, OPT_KNN_CACHE_K, OPT_NN1_OPEN_FLAT, OPT_SORT_STAGE1,
 OPT_ICP_SORTED, OPT_OVERLAP_PREP, OPT_GRID_AXES, OPT_XCD_RUN, OPT_FUSE_PARAMS, OPT_HOST_PIPE, OPT_SCAN_CHAINED, OPT_KNN_RUN) = range(1, 24)

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("PCC_LIB", _HERE / "lib" / "libpcc_nn.so"))

# every symbol include/pcc_nn.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "pcc_version", "pcc_last_error", "pcc_device_count",
    "pcc_index_create", "pcc_index_destroy", "pcc_index_size", "pcc_index_set_stream",
    "pcc_index_sync", "pcc_index_engine", "pcc_index_set_engine",
    "pcc_nn1", "pcc_knn", "pcc_radius_count", "pcc_radius_fill", "pcc_radius_count_max", "pcc_radius_fill_max",
    "pcc_euclidean_clusters", "pcc_sor", "pcc_icp_step", "pcc_transform", "pcc_icp_align",
    "pcc_match_knn", "pcc_index_stats", "pcc_index_set_input", "pcc_index_enable_timing",
    "pcc_index_timing", "pcc_first_within", "pcc_voxel_grid",
    "pcc_normals", "pcc_region_growing", "pcc_sac_plane", "pcc_rigid_from_sums",
    "pcc_rigid_from_sums_about", "pcc_icp_step_about",
    "pcc_normals_radius", "pcc_index_wait_stream", "pcc_stream_wait_index", "pcc_index_clone_to_device", "pcc_index_set_tie_order",
    "pcc_index_set_option", "pcc_index_get_option", "pcc_index_clone_to_devices", "pcc_counts_pairs", "pcc_index_sor_on_device",
    "pcc_debug_fail_alloc",
    "pcc_comm_unique_id", "pcc_comm_create_rank", "pcc_comm_create_local", "pcc_comm_destroy", "pcc_comm_info",
    "pcc_index_create_broadcast", "pcc_icp_align_sharded", "pcc_sor_partial", "pcc_sor_threshold", "pcc_sor_sharded",
]


class PccError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"libpcc_nn status {status}: {msg}")
        self.status = status


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  torch wheels ship their own libamdhip64.so (SONAME
    libamdhip64.so.7, same as /opt/rocm's).  If libpcc_nn pulled in /opt/rocm's copy first and
    torch were imported later, two runtimes would fight over the device ("No HIP GPUs are
    available").  Loading torch's copy first (without importing torch) makes both resolve to it."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.exists():
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def _load() -> C.CDLL:
    _preload_hip_runtime()
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make lib` (or __graft_entry__.build()). "
            "pointcloudcomparator_amd has no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    lib.pcc_last_error.restype = C.c_char_p
    lib.pcc_index_create.argtypes = [vp, sz, sz, i32, i32, i32, i32, C.POINTER(vp)]
    lib.pcc_index_destroy.argtypes = [vp]
    lib.pcc_index_set_input.argtypes = [vp, vp, sz, sz, i32, i32]
    lib.pcc_index_enable_timing.argtypes = [vp, i32]
    lib.pcc_index_timing.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pcc_index_size.argtypes = [vp, C.POINTER(sz)]
    lib.pcc_index_set_stream.argtypes = [vp, vp]
    lib.pcc_index_sync.argtypes = [vp]
    lib.pcc_index_set_tie_order.argtypes = [vp, i32]
    lib.pcc_index_clone_to_device.argtypes = [vp, i32, C.POINTER(vp)]
    lib.pcc_index_clone_to_devices.argtypes = [vp, C.POINTER(i32), i32, C.POINTER(vp)]
    lib.pcc_index_set_option.argtypes = [vp, i32, C.c_double]
    lib.pcc_index_get_option.argtypes = [vp, i32, C.POINTER(C.c_double)]
    lib.pcc_index_wait_stream.argtypes = [vp, vp]
    lib.pcc_stream_wait_index.argtypes = [vp, vp]
    lib.pcc_index_engine.argtypes = [vp, C.POINTER(i32)]
    lib.pcc_index_set_engine.argtypes = [vp, i32]
    lib.pcc_index_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.pcc_device_count.argtypes = [C.POINTER(i32)]
    lib.pcc_nn1.argtypes = [vp, vp, sz, sz, i32, vp, vp]
    lib.pcc_knn.argtypes = [vp, vp, sz, sz, i32, i32, vp, vp]
    lib.pcc_radius_count.argtypes = [vp, vp, sz, sz, i32, C.c_double, vp]
    lib.pcc_first_within.argtypes = [vp, vp, sz, sz, i32, C.c_double, vp]
    lib.pcc_rigid_from_sums.argtypes = [vp, vp]
    lib.pcc_rigid_from_sums_about.argtypes = [vp, vp, vp]
    lib.pcc_icp_step_about.argtypes = [vp, vp, sz, sz, i32, vp, vp, vp, C.POINTER(C.c_double)]
    lib.pcc_sac_plane.argtypes = [vp, vp, sz, sz, i32, i32, C.c_double, C.c_double, i32, vp, C.POINTER(sz), vp, vp]
    lib.pcc_normals.argtypes = [vp, i32, vp, i32, vp]
    lib.pcc_normals_radius.argtypes = [vp, C.c_double, vp, i32, vp]
    lib.pcc_region_growing.argtypes = [vp, vp, i32, i32, C.c_float, C.c_float, C.c_uint32, C.c_uint32, vp, vp]
    lib.pcc_voxel_grid.argtypes = [vp, vp, sz, sz, i32, C.c_float, i32, vp, sz, C.POINTER(sz)]
    lib.pcc_radius_fill.argtypes = [vp, vp, sz, sz, i32, C.c_double, i32, vp, vp, vp]
    lib.pcc_radius_count_max.argtypes = [vp, vp, sz, sz, i32, C.c_double, C.c_uint, vp]
    lib.pcc_radius_fill_max.argtypes = [vp, vp, sz, sz, i32, C.c_double, i32, C.c_uint, vp, vp, vp]
    lib.pcc_euclidean_clusters.argtypes = [vp, C.c_double, C.c_uint32, C.c_uint32, i32, vp,
                                           C.POINTER(C.c_int32), vp, i32]
    lib.pcc_sor.argtypes = [vp, i32, C.c_double, i32, vp, vp, C.POINTER(C.c_double), C.POINTER(sz)]
    lib.pcc_icp_step.argtypes = [vp, vp, sz, sz, i32, vp, vp, C.POINTER(C.c_double)]
    lib.pcc_transform.argtypes = [vp, C.POINTER(C.c_float), vp, sz, sz, vp, sz, i32]
    lib.pcc_icp_align.argtypes = [vp, vp, sz, sz, i32, i32, i32, C.POINTER(C.c_float),
                                  C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(i32)]
    lib.pcc_match_knn.argtypes = [vp, vp, sz, sz, i32, C.c_float, vp, C.POINTER(C.c_int32)]
    lib.pcc_comm_unique_id.argtypes = [vp, sz]
    lib.pcc_comm_create_rank.argtypes = [vp, sz, i32, i32, i32, C.POINTER(vp)]
    lib.pcc_comm_create_local.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    lib.pcc_comm_destroy.argtypes = [vp]
    lib.pcc_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.pcc_index_create_broadcast.argtypes = [vp, i32, vp, sz, sz, i32, i32, C.POINTER(vp), C.POINTER(sz)]
    lib.pcc_icp_align_sharded.argtypes = [vp, vp, vp, sz, sz, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_double),
                                          C.POINTER(i32), C.POINTER(i32)]
    lib.pcc_sor_partial.argtypes = [vp, sz, sz, i32, i32, vp, C.POINTER(C.c_double)]
    lib.pcc_sor_threshold.argtypes = [C.POINTER(C.c_double), C.c_uint64, i32, C.c_double, C.POINTER(C.c_double), C.POINTER(i32)]
    lib.pcc_sor_sharded.argtypes = [vp, vp, sz, sz, i32, C.c_double, i32, vp, vp, C.POINTER(C.c_double), C.POINTER(sz)]
    for name in SYMBOLS:
        fn = getattr(lib, name)  # raises AttributeError if the library lacks a declared symbol
        if name != "pcc_last_error":
            fn.restype = C.c_int
    return lib


LIB = _load()


def _check(status: int) -> None:
    if status != 0:
        raise PccError(status, (LIB.pcc_last_error() or b"").decode())


def device_count() -> int:
    """number of HIP devices; 0 when the runtime reports none (no exception)."""
    n = C.c_int(0)
    st = LIB.pcc_device_count(C.byref(n))
    return n.value if st == 0 else 0


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _points(x):
    """(pointer, n, stride_bytes, mem) of an (n, >=3) float32 numpy array or torch tensor."""
    if _is_torch(x):
        assert x.dtype.is_floating_point and x.element_size() == 4 and x.dim() == 2 and x.shape[1] >= 3
        assert x.stride(1) == 1
        mem = MEM_DEVICE if x.is_cuda else MEM_HOST
        return x.data_ptr(), x.shape[0], x.stride(0) * 4 if x.shape[0] > 1 else x.shape[1] * 4, mem
    a = x
    assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 2 and a.shape[1] >= 3
    assert a.shape[0] == 0 or a.strides[1] == 4
    stride = a.strides[0] if a.shape[0] > 1 else a.shape[1] * 4
    return a.ctypes.data, a.shape[0], stride, MEM_HOST


def _out(like, shape, dtype):
    """allocate an output in the memory space of `like`."""
    if _is_torch(like) and like.is_cuda:
        import torch
        tdt = {np.int32: torch.int32, np.float32: torch.float32, np.uint8: torch.uint8,
               np.int64: torch.int64}[dtype]
        t = torch.empty(shape, dtype=tdt, device=like.device)
        return t, t.data_ptr()
    a = np.empty(shape, dtype=dtype)
    return a, a.ctypes.data


def rigid_from_sums(sums, center=None):
    """4x4 float32 rigid transform from the 17 ICP sums (pcc_rigid_from_sums[_about]; host arithmetic, no handle).
    center: the point the sums were taken about (Index.icp_step(..., center=...)); None = the origin."""
    sm = np.ascontiguousarray(sums, dtype=np.float64)
    assert sm.shape == (17,)
    T = np.zeros(16, dtype=np.float32)
    if center is None:
        _check(LIB.pcc_rigid_from_sums(sm.ctypes.data, T.ctypes.data))
    else:
        cc = np.ascontiguousarray(center, dtype=np.float64)
        assert cc.shape == (3,)
        _check(LIB.pcc_rigid_from_sums_about(sm.ctypes.data, cc.ctypes.data, T.ctypes.data))
    return T.reshape(4, 4)


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """the id rank 0 makes and hands to the other ranks (pcc_comm_unique_id)"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(LIB.pcc_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


class Comm:
    """one rank of an RCCL communicator (pcc_comm): one per (process, GPU)"""

    def __init__(self, handle):
        self._c = handle

    @classmethod
    def from_id(cls, uid: bytes, world: int, rank: int, device: int):
        h = C.c_void_p()
        _check(LIB.pcc_comm_create_rank(uid, len(uid), world, rank, device, C.byref(h)))
        return cls(h)

    @classmethod
    def local(cls, devices):
        """communicators for several GPUs driven by this process: [rank 0 on devices[0], ...]"""
        arr = (C.c_int * len(devices))(*devices)
        hs = (C.c_void_p * len(devices))()
        _check(LIB.pcc_comm_create_local(arr, len(devices), hs))
        return [cls(C.c_void_p(h)) for h in hs]

    def info(self):
        r, w, d = C.c_int(0), C.c_int(0), C.c_int(0)
        _check(LIB.pcc_comm_info(self._c, C.byref(r), C.byref(w), C.byref(d)))
        return r.value, w.value, d.value

    def close(self):
        if self._c:
            LIB.pcc_comm_destroy(self._c)
            self._c = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def sor_threshold(sums, n_valid: int, mean_k: int = 50, stddev_mult: float = 1.5):
    """PCL's threshold from the combined (+, +, min, min) sums of all shards: (threshold, exact)"""
    sm = (C.c_double * 4)(*[float(v) for v in sums])
    thr, ex = C.c_double(0), C.c_int(0)
    _check(LIB.pcc_sor_threshold(sm, int(n_valid), mean_k, float(stddev_mult), C.byref(thr), C.byref(ex)))
    return thr.value, bool(ex.value)


class Index:
    """Owner of one pcc_index handle (the role pcl::KdTreeFLANN plays in the reference).

    auto_sync (default on): every call on torch CUDA tensors is bracketed with pcc_index_wait_stream / pcc_stream_wait_index
    against torch's current stream, so `ix.nn1(x * s)` reads finished data and `idx.cpu()` sees the result.  The bracket AFTER
    set_input makes torch's stream wait for the whole build, and the next call's bracket makes the library wait for torch's
    stream again: with auto_sync on, the query staging beside the build (PCC_OPT_OVERLAP_PREP) therefore hides nothing -- correct,
    only serial.  Callers that time or pipeline (bench.py) pass auto_sync=False and order the streams themselves (sync(),
    wait_stream(), stream_wait())."""

    @classmethod
    def broadcast(cls, comm: "Comm", root: int, points=None, engine: int = ENGINE_AUTO, auto_sync: bool = True):
        """the reference cloud of rank `root` indexed on every rank of `comm` (pcc_index_create_broadcast; collective).
        points: read on the root only."""
        ptr, n, stride, mem = _points(points) if points is not None else (None, 0, 12, MEM_HOST)
        if points is not None and _is_torch(points) and points.is_cuda:
            import torch
            torch.cuda.current_stream(points.device).synchronize()
        h = C.c_void_p()
        n_all = C.c_size_t(0)
        _check(LIB.pcc_index_create_broadcast(comm._c, root, ptr, n, stride, mem, engine, C.byref(h), C.byref(n_all)))
        self = cls.__new__(cls)
        self._h = h
        self.auto_sync = auto_sync
        self.n_original = n_all.value
        return self

    def __init__(self, points, engine: int = ENGINE_AUTO, device: int = 0, auto_sync: bool = True):
        """auto_sync: calls that take torch CUDA tensors are ordered against torch's current stream on both
        sides (inputs produced by torch are complete before the library reads them; torch work issued after the
        call sees the outputs) with pcc_index_wait_stream / pcc_stream_wait_index -- two event records per call,
        no host wait.  A caller that brackets its calls with sync() itself (bench.py's timed region) turns it off."""
        ptr, n, stride, mem = _points(points)
        if _is_torch(points) and points.is_cuda:
            device = points.device.index or 0
            import torch
            torch.cuda.current_stream(points.device).synchronize()  # create is synchronous anyway
        h = C.c_void_p()
        _check(LIB.pcc_index_create(ptr, n, stride, 3, mem, device, engine, C.byref(h)))
        self._h = h
        self.n_original = n
        self.auto_sync = auto_sync

    def _torch_stream(self, *tensors):
        """hipStream_t of torch's current stream on the device of the first CUDA tensor argument (None: no
        device tensor involved, or auto_sync off)."""
        if not self.auto_sync:
            return None
        for t in tensors:
            if t is not None and _is_torch(t) and t.is_cuda:
                import torch
                return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
        return None

    def _before(self, *tensors):
        st = self._torch_stream(*tensors)
        if st is not None:
            _check(LIB.pcc_index_wait_stream(self._h, st))
        return st

    def _after(self, st):
        if st is not None:
            _check(LIB.pcc_stream_wait_index(self._h, st))

    def clone_to_device(self, device: int) -> "Index":
        """a second handle over the same cloud on `device` (packed cloud copied device to device, index rebuilt there)"""
        h = C.c_void_p()
        _check(LIB.pcc_index_clone_to_device(self._h, device, C.byref(h)))
        other = Index.__new__(Index)
        other._h, other.n_original, other.auto_sync = h, self.n_original, self.auto_sync
        return other

    def clone_to_devices(self, devices):
        """one more handle over the same cloud per entry of `devices`: all peer copies in flight together, then the builds"""
        n = len(devices)
        arr = (C.c_int * n)(*devices)
        hs = (C.c_void_p * n)()
        _check(LIB.pcc_index_clone_to_devices(self._h, arr, n, hs))
        res = []
        for k in range(n):
            other = Index.__new__(Index)
            other._h, other.n_original, other.auto_sync = C.c_void_p(hs[k]), self.n_original, self.auto_sync
            res.append(other)
        return res

    def set_input(self, points):
        """pcl::KdTreeFLANN::setInputCloud on an existing object: rebuild over a new cloud,
        reusing the device allocations."""
        ptr, n, stride, mem = _points(points)
        st = self._before(points)
        _check(LIB.pcc_index_set_input(self._h, ptr, n, stride, 3, mem))
        # the pack kernel reads the caller's tensor on the library's stream after this returns: torch work issued
        # from here on (an overwrite, the allocator reusing the block) waits for it
        self._after(st)
        self.n_original = n

    def set_option(self, option: int, value: float):
        """pcc_index_set_option (OPT_*): implementation choices of this handle; no result bit depends on them"""
        _check(LIB.pcc_index_set_option(self._h, option, float(value)))

    def get_option(self, option: int) -> float:
        v = C.c_double(0)
        _check(LIB.pcc_index_get_option(self._h, option, C.byref(v)))
        return v.value

    def enable_timing(self, level=2):
        """0/False off, 1 main kernel only (cheap enough for a timed region), 2/True full breakdown"""
        level = 2 if level is True else int(level)
        _check(LIB.pcc_index_enable_timing(self._h, level))

    def timing(self):
        """ms of the instrumented kernels of the last call (see pcc_index_timing)."""
        t = (C.c_float * 8)()
        _check(LIB.pcc_index_timing(self._h, t))
        return list(t)

    def close(self):
        if getattr(self, "_h", None) and LIB is not None:  # (LIB is gone when the interpreter tears the module down)
            LIB.pcc_index_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def size(self) -> int:
        n = C.c_size_t(0)
        _check(LIB.pcc_index_size(self._h, C.byref(n)))
        return n.value

    @property
    def engine(self) -> int:
        e = C.c_int(0)
        _check(LIB.pcc_index_engine(self._h, C.byref(e)))
        return e.value

    def set_engine(self, engine: int):
        _check(LIB.pcc_index_set_engine(self._h, engine))

    def set_tie_order(self, ties: int):
        """TIES_LOWEST_INDEX (default) or TIES_FLANN: which of several equally near references nn1 / match_knn name"""
        _check(LIB.pcc_index_set_tie_order(self._h, ties))

    def set_stream(self, stream_ptr: int):
        _check(LIB.pcc_index_set_stream(self._h, C.c_void_p(stream_ptr)))

    def wait_stream(self, stream_ptr: int):
        """pcc_index_wait_stream: calls made after this one start only when everything submitted so far to the stream
        `stream_ptr` (a hipStream_t as an integer, e.g. torch.cuda.Stream().cuda_stream) has finished"""
        _check(LIB.pcc_index_wait_stream(self._h, C.c_void_p(stream_ptr)))

    def stream_wait(self, stream_ptr: int):
        """pcc_stream_wait_index: work submitted to `stream_ptr` from now on waits for what the index has been asked so far"""
        _check(LIB.pcc_stream_wait_index(self._h, C.c_void_p(stream_ptr)))

    def sync(self):
        _check(LIB.pcc_index_sync(self._h))

    def stats(self):
        s = (C.c_uint64 * 8)()
        _check(LIB.pcc_index_stats(self._h, s))
        return list(s)

    # -- searches ------------------------------------------------------------
    def nn1(self, queries, out_idx=None, out_d2=None):
        ptr, n, stride, mem = _points(queries)
        if out_idx is None:
            out_idx, pi = _out(queries, (n,), np.int32)
        else:
            pi = out_idx.data_ptr() if _is_torch(out_idx) else out_idx.ctypes.data
        if out_d2 is None:
            out_d2, pd = _out(queries, (n,), np.float32)
        else:
            pd = out_d2.data_ptr() if _is_torch(out_d2) else out_d2.ctypes.data
        st = self._before(queries, out_idx, out_d2)
        _check(LIB.pcc_nn1(self._h, ptr, n, stride, mem, pi, pd))
        self._after(st)
        return out_idx, out_d2

    def knn(self, queries, k: int):
        ptr, n, stride, mem = _points(queries)
        idx, pi = _out(queries, (n, k), np.int32)
        d2, pd = _out(queries, (n, k), np.float32)
        st = self._before(queries)
        _check(LIB.pcc_knn(self._h, ptr, n, stride, mem, k, pi, pd))
        self._after(st)
        return idx, d2

    def radius_count(self, queries, radius: float, max_nn: int = 0):
        ptr, n, stride, mem = _points(queries)
        cnt, pc = _out(queries, (n,), np.int32)
        st = self._before(queries)
        _check(LIB.pcc_radius_count_max(self._h, ptr, n, stride, mem, float(radius), int(max_nn), pc))
        self._after(st)
        return cnt

    def voxel_grid(self, points, leaf: float, has_rgb: bool = False):
        """pcl::VoxelGrid centroids of `points` (host array (n, >=3) float32; rgb word in column 4)."""
        ptr, n, stride, mem = _points(points)
        assert mem == MEM_HOST
        out = np.zeros((n, points.shape[1]), dtype=np.float32)
        cnt = C.c_size_t(0)
        _check(LIB.pcc_voxel_grid(self._h, ptr, n, stride, mem, np.float32(leaf), int(has_rgb), out.ctypes.data,
                                  out.strides[0] if n > 1 else points.shape[1] * 4, C.byref(cnt)))
        return out[:cnt.value]

    def first_within(self, queries, radius: float):
        """lowest index of a reference within `radius` (double-precision test), -1 if none"""
        ptr, n, stride, mem = _points(queries)
        idx, pi = _out(queries, (n,), np.int32)
        st = self._before(queries)
        _check(LIB.pcc_first_within(self._h, ptr, n, stride, mem, float(radius), pi))
        self._after(st)
        return idx

    def radius_search(self, queries, radius: float, sorted: bool = True, max_nn: int = 0):
        """CSR (offsets, idx, d2) of all neighbours with d2 < float(radius^2); max_nn > 0: the max_nn nearest of them."""
        ptr, n, stride, mem = _points(queries)
        cnt = self.radius_count(queries, radius, max_nn)
        if _is_torch(cnt):
            import torch
            if not self.auto_sync:
                self.sync()  # device results are written on the library's stream, torch works on its own
            offs = torch.zeros(n + 1, dtype=torch.int64, device=cnt.device)
            offs[1:] = torch.cumsum(cnt.to(torch.int64), 0)
            total = int(offs[-1].item())
            po = offs.data_ptr()
        else:
            offs = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(cnt, out=offs[1:])
            total = int(offs[-1])
            po = offs.ctypes.data
        idx, pi = _out(queries, (max(total, 1),), np.int32)
        d2, pd = _out(queries, (max(total, 1),), np.float32)
        st = self._before(queries, offs)
        _check(LIB.pcc_radius_fill_max(self._h, ptr, n, stride, mem, float(radius), int(sorted), int(max_nn), po, pi, pd))
        self._after(st)
        return offs, idx[:total], d2[:total]

    def radius_fill(self, queries, radius: float, offsets, sorted: bool = True, max_nn: int = 0):
        """the fill pass alone, for CSR offsets the caller already holds (host queries, numpy int64 offsets[nq + 1])"""
        ptr, n, stride, mem = _points(queries)
        assert mem == MEM_HOST and len(offsets) == n + 1
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        total = int(offs[-1])
        idx = np.empty(max(total, 1), np.int32)
        d2 = np.empty(max(total, 1), np.float32)
        _check(LIB.pcc_radius_fill_max(self._h, ptr, n, stride, mem, float(radius), int(sorted), int(max_nn), offs.ctypes.data,
                                       idx.ctypes.data, d2.ctypes.data))
        return idx[:total], d2[:total]

    def euclidean_clusters(self, tolerance: float, min_size: int, max_size: int, device_out=None,
                           max_sizes: int = 65536):
        if device_out is not None:
            labels, pl = _out(device_out, (self.n_original,), np.int32)
            mem = MEM_DEVICE
        else:
            labels = np.empty(self.n_original, dtype=np.int32)
            pl, mem = labels.ctypes.data, MEM_HOST
        ncl = C.c_int32(0)
        sizes = np.zeros(max_sizes, dtype=np.int32)
        st = self._before(device_out)
        _check(LIB.pcc_euclidean_clusters(self._h, float(tolerance), min_size, max_size, mem, pl,
                                          C.byref(ncl), sizes.ctypes.data, max_sizes))
        self._after(st)
        return labels, ncl.value, sizes[:min(ncl.value, max_sizes)]

    def sor(self, mean_k: int = 50, stddev_mult: float = 1.5, device=None):
        """pcl::StatisticalOutlierRemoval over the indexed cloud: (mean distances, inlier mask, threshold, kept).
        device: a torch device -> the two arrays stay in HBM (torch tensors), nothing cloud-sized crosses PCIe."""
        thr = C.c_double(0)
        kept = C.c_size_t(0)
        if device is not None:
            import torch
            md = torch.empty(self.n_original, dtype=torch.float32, device=device)
            inl = torch.empty(self.n_original, dtype=torch.uint8, device=device)
            st = self._before(md)
            _check(LIB.pcc_sor(self._h, mean_k, float(stddev_mult), MEM_DEVICE, md.data_ptr(), inl.data_ptr(), C.byref(thr), C.byref(kept)))
            self._after(st)
            return md, inl, thr.value, kept.value
        md = np.empty(self.n_original, dtype=np.float32)
        inl = np.empty(self.n_original, dtype=np.uint8)
        _check(LIB.pcc_sor(self._h, mean_k, float(stddev_mult), MEM_HOST, md.ctypes.data, inl.ctypes.data,
                           C.byref(thr), C.byref(kept)))
        return md, inl, thr.value, kept.value

    def sor_on_device(self) -> bool:
        """whether the last sor() took its sums, threshold and mask on the device (else: the in-order host loop)"""
        f = C.c_int(0)
        _check(LIB.pcc_index_sor_on_device(self._h, C.byref(f)))
        return bool(f.value)

    def sor_partial(self, start: int, count: int, mean_k: int = 50):
        """mean distances of the points [start, start + count) and the shard's share of the statistics (pcc_sor_partial)"""
        md = np.empty(count, dtype=np.float32)
        sums = (C.c_double * 4)()
        _check(LIB.pcc_sor_partial(self._h, start, count, mean_k, MEM_HOST, md.ctypes.data, sums))
        return md, np.array(list(sums))

    def sor_sharded(self, comm: "Comm", start: int, count: int, mean_k: int = 50, stddev_mult: float = 1.5):
        md = np.empty(count, dtype=np.float32)
        inl = np.empty(count, dtype=np.uint8)
        thr, kept = C.c_double(0), C.c_size_t(0)
        _check(LIB.pcc_sor_sharded(self._h, comm._c, start, count, mean_k, float(stddev_mult), MEM_HOST, md.ctypes.data,
                                   inl.ctypes.data, C.byref(thr), C.byref(kept)))
        return md, inl, thr.value, kept.value

    def icp_align_sharded(self, comm: "Comm", source_shard, max_iter: int = 20, fixed: bool = False):
        """pcc_icp_align_sharded: (T 4x4, fitness over all shards, iterations, converged)"""
        ptr, n, stride, mem = _points(source_shard)
        T = (C.c_float * 16)()
        fit, it, conv = C.c_double(0), C.c_int(0), C.c_int(0)
        st = self._before(source_shard)
        _check(LIB.pcc_icp_align_sharded(self._h, comm._c, ptr, n, stride, mem, max_iter, int(fixed), T, C.byref(fit),
                                         C.byref(it), C.byref(conv)))
        self._after(st)
        return np.array(list(T), dtype=np.float32).reshape(4, 4), fit.value, it.value, bool(conv.value)

    def sac_plane(self, points, max_iterations: int = 100, threshold: float = 0.02, probability: float = 0.99,
                  optimize: bool = True):
        """pcl::SACSegmentation(PLANE, RANSAC).segment on `points`: (inlier indices, coefficients[4], iterations)."""
        ptr, n, stride, mem = _points(points)
        inl, pi = _out(points, (max(n, 1),), np.int32)
        cnt = C.c_size_t(0)
        its = C.c_int(0)
        coeff = np.zeros(4, dtype=np.float32)
        st = self._before(points)
        _check(LIB.pcc_sac_plane(self._h, ptr, n, stride, mem, max_iterations, float(threshold), float(probability),
                                 int(optimize), pi, C.byref(cnt), coeff.ctypes.data, C.byref(its)))
        self._after(st)
        return inl[:cnt.value], coeff, its.value

    def normals(self, k: int = 50, viewpoint=None, device=None):
        """pcl::NormalEstimation over the index's own points: (n, 4) = nx, ny, nz, curvature.
        device: a torch cuda device to get the result as a device tensor."""
        vp = None
        if viewpoint is not None:
            vpa = np.ascontiguousarray(viewpoint, dtype=np.float32)
            vp = vpa.ctypes.data
        if device is not None:
            import torch
            out = torch.empty((self.n_original, 4), dtype=torch.float32, device=device)
            st = self._before(out)
            _check(LIB.pcc_normals(self._h, k, vp, MEM_DEVICE, out.data_ptr()))
            self._after(st)
            if st is None:
                self.sync()
            return out
        out = np.empty((self.n_original, 4), dtype=np.float32)
        _check(LIB.pcc_normals(self._h, k, vp, MEM_HOST, out.ctypes.data))
        return out

    def normals_radius(self, radius: float, viewpoint=None):
        """pcl::NormalEstimation with setRadiusSearch(radius): (n, 4) = nx, ny, nz, curvature (host array)."""
        vp = None
        if viewpoint is not None:
            vpa = np.ascontiguousarray(viewpoint, dtype=np.float32)
            vp = vpa.ctypes.data
        out = np.empty((self.n_original, 4), dtype=np.float32)
        _check(LIB.pcc_normals_radius(self._h, float(radius), vp, MEM_HOST, out.ctypes.data))
        return out

    def region_growing(self, normals, k: int = 100, smoothness: float = 3.0 / 180.0 * np.pi,
                       curvature_threshold: float = 1.0, min_size: int = 50, max_size: int = 1000000):
        """pcl::RegionGrowing::extract over the index's own points: (labels, n_clusters)."""
        ncl = C.c_int32(0)
        if _is_torch(normals) and normals.is_cuda:
            import torch
            assert normals.dtype == torch.float32 and normals.is_contiguous()
            labels = torch.empty(self.n_original, dtype=torch.int32, device=normals.device)
            st = self._before(normals)
            if st is None:
                torch.cuda.current_stream(normals.device).synchronize()
            _check(LIB.pcc_region_growing(self._h, normals.data_ptr(), MEM_DEVICE, k, np.float32(smoothness),
                                          np.float32(curvature_threshold), min_size, max_size, labels.data_ptr(),
                                          C.byref(ncl)))
            self._after(st)
            return labels, ncl.value
        nm = np.ascontiguousarray(normals, dtype=np.float32)
        assert nm.shape == (self.n_original, 4)
        labels = np.empty(self.n_original, dtype=np.int32)
        _check(LIB.pcc_region_growing(self._h, nm.ctypes.data, MEM_HOST, k, np.float32(smoothness),
                                      np.float32(curvature_threshold), min_size, max_size, labels.ctypes.data,
                                      C.byref(ncl)))
        return labels, ncl.value

    def icp_step(self, src, want_corr: bool = True, center=None):
        """correspondences of `src` against the index + the 17 Umeyama sums; center: take the sums about this point
        (every rank of a sharded loop the SAME one) -- needed for small clouds at large coordinates"""
        ptr, n, stride, mem = _points(src)
        sums = (C.c_double * 17)()
        if want_corr:
            idx, pi = _out(src, (n,), np.int32)
            d2, pd = _out(src, (n,), np.float32)
        else:
            idx = d2 = None
            pi = pd = None
        st = self._before(src)
        if center is None:
            _check(LIB.pcc_icp_step(self._h, ptr, n, stride, mem, pi, pd, sums))
        else:
            cc = np.ascontiguousarray(center, dtype=np.float64)
            assert cc.shape == (3,)
            _check(LIB.pcc_icp_step_about(self._h, ptr, n, stride, mem, cc.ctypes.data, pi, pd, sums))
        self._after(st)
        return idx, d2, np.array(list(sums), dtype=np.float64)

    def transform(self, T, src, dst=None):
        ptr, n, stride, mem = _points(src)
        Tm = np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(16))
        if dst is None:
            if _is_torch(src):
                import torch
                dst = torch.empty((n, 3), dtype=torch.float32, device=src.device)
            else:
                dst = np.empty((n, 3), dtype=np.float32)
        dptr, dn, dstride, dmem = _points(dst)
        assert dn == n and dmem == mem
        st = self._before(src, dst)
        _check(LIB.pcc_transform(self._h, Tm.ctypes.data_as(C.POINTER(C.c_float)), ptr, n, stride, dptr,
                                 dstride, mem))
        self._after(st)
        return dst

    def icp_align(self, src, max_iter: int = 20, fixed: bool = False):
        ptr, n, stride, mem = _points(src)
        T = (C.c_float * 16)()
        fit = C.c_double(0)
        it = C.c_int(0)
        conv = C.c_int(0)
        self._before(src)  # (the call synchronises the index's stream before it returns)
        _check(LIB.pcc_icp_align(self._h, ptr, n, stride, mem, max_iter, int(fixed), T, C.byref(fit),
                                 C.byref(it), C.byref(conv)))
        return np.array(list(T), dtype=np.float32).reshape(4, 4), fit.value, it.value, bool(conv.value)

    def match_knn(self, des2, threshold: float = 0.05):
        ptr, n, stride, mem = _points(des2)
        out = np.empty(n + 1, dtype=np.int32)
        sz = C.c_int32(0)
        self._before(des2)
        _check(LIB.pcc_match_knn(self._h, ptr, n, stride, mem, np.float32(threshold), out.ctypes.data,
                                 C.byref(sz)))
        return out[:sz.value]
