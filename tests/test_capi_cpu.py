"""CPU suite: the C-ABI library loads, exports every symbol include/pcc_nn.h declares, and
fails loudly (never falls back to a CPU path) when no GPU is present."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "pcc_nn.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from pointcloudcomparator_amd import capi
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(capi.LIB, name), f"{name} declared in pcc_nn.h but not exported"
    assert sorted(capi.SYMBOLS) == declared


def test_option_constants_of_the_binding_match_the_header():
    """enum pcc_option in include/pcc_nn.h <-> capi.OPT_*: same names, same values, none missing on either side"""
    from pointcloudcomparator_amd import capi
    text = (ROOT / "include" / "pcc_nn.h").read_text()
    body = re.search(r"enum pcc_option\s*\{(.*?)\};", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    header = {m.group(1): int(m.group(2)) for m in re.finditer(r"PCC_(OPT_[A-Z0-9_]+)\s*=\s*(\d+)", body)}
    binding = {k: v for k, v in vars(capi).items() if k.startswith("OPT_")}
    assert len(header) >= 13 and header == binding


def test_version_and_error_string():
    from pointcloudcomparator_amd import capi
    assert capi.LIB.pcc_version() == 100
    assert isinstance(capi.LIB.pcc_last_error(), bytes)


def test_argument_validation_needs_no_gpu():
    from pointcloudcomparator_amd import capi
    pts = np.zeros((4, 3), np.float32)
    h = ctypes.c_void_p()
    L = capi.LIB
    assert L.pcc_index_create(pts.ctypes.data, 4, 12, 5, 0, 0, 0, ctypes.byref(h)) == -5   # dim != 3
    assert L.pcc_index_create(pts.ctypes.data, 4, 10, 3, 0, 0, 0, ctypes.byref(h)) == -1   # bad stride
    assert L.pcc_index_create(pts.ctypes.data, 0, 12, 3, 0, 0, 0, ctypes.byref(h)) == -2   # empty cloud
    assert L.pcc_index_create(None, 4, 12, 3, 0, 0, 0, ctypes.byref(h)) == -1              # null points
    assert b"empty input cloud" in L.pcc_last_error() or b"null" in L.pcc_last_error()
    assert L.pcc_nn1(None, pts.ctypes.data, 4, 12, 0, None, None) == -1                     # null index
    # every entry point refuses a null handle before touching anything else
    out = np.zeros(64, np.float32)
    cnt = ctypes.c_size_t(0)
    n32 = ctypes.c_int32(0)
    P = pts.ctypes.data
    assert L.pcc_knn(None, P, 4, 12, 0, 1, None, None) == -1
    assert L.pcc_radius_count(None, P, 4, 12, 0, ctypes.c_double(0.1), out.ctypes.data) == -1
    assert L.pcc_euclidean_clusters(None, ctypes.c_double(0.1), 1, 10, 0, out.ctypes.data, ctypes.byref(n32), None, 0) == -1
    assert L.pcc_sor(None, 5, ctypes.c_double(1.0), 0, None, None, None, None) == -1
    assert L.pcc_first_within(None, P, 4, 12, 0, ctypes.c_double(0.1), out.ctypes.data) == -1
    assert L.pcc_voxel_grid(None, P, 4, 12, 0, ctypes.c_float(0.1), 0, out.ctypes.data, 12, ctypes.byref(cnt)) == -1
    assert L.pcc_normals(None, 5, None, 0, out.ctypes.data) == -1
    assert L.pcc_region_growing(None, out.ctypes.data, 0, 5, ctypes.c_float(0.1), ctypes.c_float(1.0), 1, 10,
                                out.ctypes.data, ctypes.byref(n32)) == -1
    assert L.pcc_sac_plane(None, P, 4, 12, 0, 10, ctypes.c_double(0.02), ctypes.c_double(0.99), 1, out.ctypes.data,
                           ctypes.byref(cnt), out.ctypes.data, None) == -1
    assert L.pcc_index_destroy(None) == 0                                                    # like free(NULL)


def test_no_cpu_fallback_without_device():
    from pointcloudcomparator_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.PccError) as e:
        capi.Index(np.zeros((10, 3), np.float32))
    assert e.value.status == -3 and "no CPU path" in str(e.value)


def test_product_never_imports_the_oracle():
    for f in (ROOT / "pointcloudcomparator_amd").rglob("*"):
        if f.suffix in {".py", ".hip", ".hpp", ".cpp", ".h"}:
            t = f.read_text()
            assert "import oracle" not in t and "pcc_oracle" not in t and "orc_" not in t, f
    for f in (ROOT / "include").rglob("*"):
        if f.is_file():
            assert "orc_" not in f.read_text(), f


def test_rigid_from_sums_about_a_centre():
    """pcc_rigid_from_sums_about (host arithmetic, no GPU): a small cloud at geo-referenced coordinates, rotated by 2
    degrees about its own centre.  Sums about the origin lose the rotation in cancellation (4e11 against 0.3), sums
    about a point of the cloud give it back; the two agree for a cloud at the origin."""
    from pointcloudcomparator_amd import capi
    rng = np.random.default_rng(4)
    local = rng.random((200, 3)) * 0.1
    c_, s_ = np.cos(np.radians(2.0)), np.sin(np.radians(2.0))
    R = np.array([[c_, -s_, 0], [s_, c_, 0], [0, 0, 1]])
    t = np.array([0.01, -0.02, 0.005])

    def sums_of(p, q, c):
        P, Q = p - c, q - c
        return np.concatenate([P.sum(0), Q.sum(0), (Q.T @ P).ravel(), [((p - q) ** 2).sum()], [len(p)]])

    for offset in (np.zeros(3), np.array([1000.0, 100000.0, 100000.0])):
        p = local + offset
        q = (local - local.mean(0)) @ R.T + local.mean(0) + t + offset
        ctr = p[0]
        T = capi.rigid_from_sums(sums_of(p, q, ctr), center=ctr)
        assert np.allclose(T[:3, :3], R, atol=2e-6)
        moved = p @ T[:3, :3].astype(np.float64).T + T[:3, 3].astype(np.float64)
        assert np.abs(moved - q).max() < (1e-6 if offset[1] == 0 else 2e-2)   # (the float translation resolves 0.008 out there)
        T0 = capi.rigid_from_sums(sums_of(p, q, np.zeros(3)))
        if offset[1] == 0:
            assert np.allclose(T0, T, atol=1e-6)
        else:
            assert not np.allclose(T0[:3, :3], R, atol=2e-6)                  # the origin-centred sums lost it

