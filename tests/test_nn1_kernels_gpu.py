"""The pruned k = 1 kernel has two forms (PCC_OPT_NN1_KERNEL): 0 = one lane per query (k_grid_nn1), 1 = rows drained
flat with lanes over candidates (k_grid_nn1_flat2); 2 and 3 force how the flat form finishes the lanes its cube leaves
open (a second kernel over a compacted list / in place), which 1 chooses by the number of queries.  All of them are
exact: indices and d2 bits must equal the oracle's exhaustive scan -- reference src/comparator.cpp:571-577."""
import numpy as np
import pytest

import oracle
from pointcloudcomparator_amd import capi, synth

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def _scenes():
    rng = np.random.default_rng(11)
    out = []
    a = synth.corridor_cloud(60000, synth.SEED_A)
    b = synth.corridor_cloud(20000, synth.SEED_B)
    out.append(("corridor", a, b))
    out.append(("objects", synth.corridor_cloud(60000, synth.SEED_A, layer="objects"),
                synth.corridor_cloud(20000, synth.SEED_B, layer="objects")))
    base = rng.random((3000, 3), dtype=np.float32)
    out.append(("duplicates", np.concatenate([base, base[::-1], base[:700]]),
                np.concatenate([base[rng.integers(0, 3000, 2000)], rng.random((1000, 3), dtype=np.float32)])))
    plane = rng.random((20000, 3), dtype=np.float32)
    plane[:, 2] = 0.25
    out.append(("plane", plane, rng.random((5000, 3), dtype=np.float32)))
    same = np.full((9000, 3), 0.5, np.float32)          # one cell holds everything: spans past the flat pass's capacity
    same[:100] = rng.random((100, 3), dtype=np.float32)
    out.append(("pile", same, rng.random((3000, 3), dtype=np.float32)))
    clumps = (rng.integers(0, 6, (30000, 3)) * np.float32(0.2) + rng.normal(0, 0.004, (30000, 3))).astype(np.float32)
    out.append(("clumps", clumps, (rng.random((8000, 3)) * 1.2 - 0.1).astype(np.float32)))
    far = synth.corridor_cloud(8000, synth.SEED_B)
    far[:3000] += np.float32([1.5, -2.0, 0.7])
    far[3000:4000] += np.float32(30.0)
    out.append(("far", a, far))
    nf = synth.corridor_cloud(30000, synth.SEED_A)
    nf[::97, 1] = np.nan
    qn = synth.corridor_cloud(9000, synth.SEED_B)
    qn[::53, 0] = np.inf
    out.append(("nonfinite", nf, qn))
    return out


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 12])
def test_nn1_kernel_forms_match_the_oracle(gpu, mode):
    """(12: form 2 with the listed open lanes finished one lane per query -- PCC_OPT_NN1_OPEN_FLAT = 0, round 3's kernel)"""
    open_flat, mode = (0, 2) if mode == 12 else (1, mode)
    for name, ref, qry in _scenes():
        oi, od = oracle.nn1_exhaustive(ref, qry)
        with capi.Index(ref, engine=capi.ENGINE_GRID) as ix:
            ix.set_option(capi.OPT_NN1_KERNEL, mode)
            ix.set_option(capi.OPT_NN1_OPEN_FLAT, open_flat)
            assert ix.get_option(capi.OPT_NN1_KERNEL) == mode and ix.get_option(capi.OPT_NN1_OPEN_FLAT) == open_flat
            for _ in range(2):  # (the second call takes the far route where the first had fallbacks)
                idx, d2 = ix.nn1(qry)
                assert (_bits(d2) == _bits(od)).all(), (name, mode, np.nonzero(_bits(d2) != _bits(od))[0][:5])
                assert (idx == oi).all(), (name, mode, np.nonzero(idx != oi)[0][:5])


def test_nn1_kernel_forms_agree_at_a_million(gpu):
    torch = pytest.importorskip("torch")
    n = 1_000_000
    a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A)).cuda()
    b = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B)).cuda()
    res = []
    with capi.Index(a, engine=capi.ENGINE_GRID) as ix:
        for mode in (0, 1, 2, 3):
            ix.set_option(capi.OPT_NN1_KERNEL, mode)
            idx, d2 = ix.nn1(b)
            res.append((idx.cpu().numpy(), d2.cpu().numpy()))
    with capi.Index(a, engine=capi.ENGINE_BRUTE) as ix:  # the exhaustive kernel on a sample
        bi, bd = ix.nn1(b[:20000])
        bi, bd = bi.cpu().numpy(), bd.cpu().numpy()
    for idx, d2 in res:
        assert (idx == res[0][0]).all() and (_bits(d2) == _bits(res[0][1])).all()
        assert (idx[:20000] == bi).all() and (_bits(d2[:20000]) == _bits(bd)).all()


@pytest.mark.parametrize("stage", [0, 1, 2])
def test_three_level_sort_forms_build_the_same_index(gpu, stage):
    """PCC_OPT_SORT_STAGE1: level 1 of the sort for clouds beyond the L2s with and without bucket-sorted LDS tiles (references /
    queries): 2M x 2M through the three-level sort on both sides (thresholds lowered), results = the default's, bit for bit"""
    torch = pytest.importorskip("torch")
    n = 2_000_000
    a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A)).cuda()
    b = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_B)).cuda()
    with capi.Index(a, engine=capi.ENGINE_GRID) as ix:
        i0, d0 = ix.nn1(b)
        ix.set_option(capi.OPT_SORT_STAGE1, stage)
        ix.set_option(capi.OPT_SORT_MP_MIN, 1000)       # the three-level sort for both clouds
        ix.set_option(capi.OPT_SORT_MP_MIN_Q, 1000)
        ix.set_input(a)
        i1, d1 = ix.nn1(b)
        assert (i0 == i1).all().item() and (d0.view(torch.int32) == d1.view(torch.int32)).all().item()
        assert ix.stats()[1] == 0
    with capi.Index(a[:50000], engine=capi.ENGINE_BRUTE) as bx:      # and against the exhaustive kernel on a slice
        bi, bd = bx.nn1(b[:3000])
    with capi.Index(a[:50000], engine=capi.ENGINE_GRID) as gx:
        gx.set_option(capi.OPT_SORT_STAGE1, stage)
        gx.set_option(capi.OPT_SORT_MP_MIN, 1000)
        gx.set_option(capi.OPT_SORT_MP_MIN_Q, 1000)
        gx.set_input(a[:50000])
        gi, gdd = gx.nn1(b[:3000])
    assert (bi == gi).all().item() and (bd.view(torch.int32) == gdd.view(torch.int32)).all().item()


def test_options_are_per_handle_and_validated(gpu):
    a = synth.corridor_cloud(5000, synth.SEED_A)
    with capi.Index(a) as ix, capi.Index(a) as iy:
        ix.set_option(capi.OPT_FAR_MODE, 1)
        assert ix.get_option(capi.OPT_FAR_MODE) == 1 and iy.get_option(capi.OPT_FAR_MODE) == -1
        for opt, bad in ((capi.OPT_GRID_PPC, 0.0), (capi.OPT_FAR_MODE, 2), (capi.OPT_NN1_KERNEL, 4), (99, 1), (capi.OPT_ICP_WARM, 0.5)):
            with pytest.raises(capi.PccError):
                ix.set_option(opt, bad)
