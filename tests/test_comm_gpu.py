"""The multi-GPU entry points of the C-ABI (RCCL: pcc_comm_*, pcc_index_create_broadcast, pcc_icp_align_sharded,
pcc_sor_partial / pcc_sor_threshold / pcc_sor_sharded) on the one-GPU box: a ONE-rank communicator runs every collective
(broadcast, all-reduce SUM / MIN) for real, and must reproduce the single-GPU calls bit for bit; the shard arithmetic
itself -- two shards of one cloud combined by hand -- is checked without a communicator.  Reference: the single tree /
ICP / SOR object per call site (src/comparator.cpp:564-577, 1089-1110, 1523-1541) that SURVEY.md 8e replicates per GPU."""
import numpy as np
import pytest

import oracle
from pointcloudcomparator_amd import capi, sharding, synth

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def test_one_rank_communicator_broadcast_icp_and_sor_match_the_single_gpu_calls(gpu):
    import torch
    tgt = synth.corridor_cloud(60000, synth.SEED_A)
    tgt[17] = np.nan                                            # a non-finite reference survives the broadcast as one
    src = synth.rigid_offset(synth.corridor_cloud(20000, synth.SEED_B), rot_deg=0.5, t=(0.01, -0.01, 0.005))
    with capi.Comm.from_id(capi.comm_unique_id(), 1, 0, 0) as comm:
        assert comm.info() == (0, 1, 0)
        with capi.Index.broadcast(comm, 0, tgt) as bx, capi.Index(tgt) as ix:
            assert bx.n_original == len(tgt) and bx.stats()[2] == len(tgt) - 1
            q = synth.corridor_cloud(5000, synth.SEED_B)
            bi, bd = bx.nn1(q)
            ii, id_ = ix.nn1(q)
            assert (bi == ii).all() and (_bits(bd) == _bits(id_)).all() and 17 not in bi
            # sharded ICP over one rank == pcc_icp_align: transform, fitness, iterations, verdict -- host and device source
            for s in (src, torch.from_numpy(src).cuda()):
                T1, f1, it1, c1 = ix.icp_align(s, max_iter=12, fixed=True)
                T2, f2, it2, c2 = bx.icp_align_sharded(comm, s, max_iter=12, fixed=True)
                assert (T1.view(np.uint32) == T2.view(np.uint32)).all() and f1 == f2 and it1 == it2 == 12 and c1 == c2
            T1, f1, it1, c1 = ix.icp_align(src, max_iter=30)
            T2, f2, it2, c2 = bx.icp_align_sharded(comm, src, max_iter=30)
            assert (T1.view(np.uint32) == T2.view(np.uint32)).all() and f1 == f2 and it1 == it2 and c1 == c2
            # sharded SOR over one rank (the whole cloud as its shard) == pcc_sor == the oracle
            md, inl, thr, kept = ix.sor(50, 1.5)
            smd, sinl, sthr, skept = bx.sor_sharded(comm, 0, len(tgt), 50, 1.5)
            assert (_bits(md) == _bits(smd)).all() and (inl == sinl).all() and thr == sthr and kept == skept
            omd, oinl, othr, okept = oracle.sor(tgt, 50, 1.5)
            assert sthr == othr and skept == okept


def test_local_communicator_rejects_a_device_listed_twice(gpu):
    with pytest.raises(capi.PccError):
        capi.Comm.local([0, 0])
    comms = capi.Comm.local([0])
    assert comms[0].info() == (0, 1, 0)
    comms[0].close()


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_sor_shards_combine_to_the_whole_clouds_statistics(gpu, parts):
    """the shard arithmetic without a communicator: partial sums of `parts` shards on one handle, combined with
    (+, +, min, min) as the all-reduces do, give PCL's threshold over the whole cloud -- the oracle's bits"""
    pts = synth.corridor_cloud(50000, synth.SEED_A)
    pts[123] = np.inf
    n = len(pts)
    omd, oinl, othr, okept = oracle.sor(pts, 50, 1.5)
    with capi.Index(pts) as ix:
        n_valid = ix.stats()[2]
        tot = np.array([0.0, 0.0, np.inf, np.inf])
        means, kept = [], 0
        for r in range(parts):
            start, count = sharding.shard_range(n, r, parts)
            md, sums = ix.sor_partial(start, count, 50)
            assert (_bits(md) == _bits(omd[start:start + count])).all()
            tot[:2] += sums[:2]
            tot[2:] = np.minimum(tot[2:], sums[2:])
            means.append(md)
        thr, exact = capi.sor_threshold(tot, n_valid, 50, 1.5)
        assert exact and thr == othr
        for md in means:
            kept += int((~(md.astype(np.float64) > thr)).sum())
        assert kept == okept
        # an empty shard contributes nothing
        md, sums = ix.sor_partial(n, 0, 50)
        assert len(md) == 0 and sums[0] == 0.0 and sums[1] == 0.0 and np.isinf(np.float32(0).view(np.float32) + np.array(sums[2], dtype=np.float64).astype(np.uint32).view(np.float32))
    # micrometre clumps next to decimetres: the combined sums are flagged inexact
    rng = np.random.default_rng(5)
    seeds = pts[:300]
    clumps = (seeds[:, None, :] + rng.normal(0, 2e-7, (300, 70, 3))).reshape(-1, 3).astype(np.float32)
    wide = np.concatenate([synth.corridor_cloud(60000, synth.SEED_B), clumps]).astype(np.float32)
    with capi.Index(wide) as ix:
        tot = np.array([0.0, 0.0, np.inf, np.inf])
        for r in range(parts):
            start, count = sharding.shard_range(len(wide), r, parts)
            _, sums = ix.sor_partial(start, count, 50)
            tot[:2] += sums[:2]
            tot[2:] = np.minimum(tot[2:], sums[2:])
        _, exact = capi.sor_threshold(tot, ix.stats()[2], 50, 1.5)
        assert not exact


def test_one_rank_communicator_on_a_small_cloud_and_edge_shards(gpu):
    """a cloud below the GRID engine's threshold (exhaustive engine; SOR builds the grid on demand), an empty SOR shard,
    a shard outside the cloud, an index on another handle's communicator arguments"""
    tgt = synth.corridor_cloud(900, synth.SEED_A)
    src = synth.rigid_offset(synth.corridor_cloud(400, synth.SEED_B), rot_deg=0.3, t=(0.005, 0.0, 0.0))
    with capi.Comm.from_id(capi.comm_unique_id(), 1, 0, 0) as comm:
        with capi.Index.broadcast(comm, 0, tgt) as bx, capi.Index(tgt) as ix:
            T1, f1, it1, c1 = ix.icp_align(src, max_iter=8)
            T2, f2, it2, c2 = bx.icp_align_sharded(comm, src, max_iter=8)
            assert (T1.view(np.uint32) == T2.view(np.uint32)).all() and f1 == f2 and it1 == it2 and c1 == c2
            md, inl, thr, kept = ix.sor(20, 1.0)
            smd, sinl, sthr, skept = bx.sor_sharded(comm, 0, len(tgt), 20, 1.0)
            assert (_bits(md) == _bits(smd)).all() and (inl == sinl).all() and thr == sthr and kept == skept
            # an empty shard still takes part in the collectives and learns the whole cloud's threshold
            emd, einl, ethr, ekept = bx.sor_sharded(comm, len(tgt), 0, 20, 1.0)
            assert len(emd) == 0 and len(einl) == 0 and ekept == 0   # (one rank: the sums of "all shards" are this empty one's)
            with pytest.raises(capi.PccError):
                bx.sor_sharded(comm, 10, len(tgt), 20, 1.0)                          # runs past the end of the cloud
            with pytest.raises(capi.PccError):
                bx.icp_align_sharded(comm, src[:0], max_iter=3)                       # every rank needs a non-empty shard
        with pytest.raises(capi.PccError):
            capi.Index.broadcast(comm, 3, tgt)                                        # root outside the communicator


def test_python_sharding_helper_over_the_real_partial_sums(gpu):
    """sharding.sor_sharded (what a torch.distributed job calls per rank) on top of pcc_sor_partial / pcc_sor_threshold:
    with one rank it reproduces pcc_sor; tests/test_sharding_cpu.py runs the same helper over two gloo ranks"""
    pts = synth.corridor_cloud(30000, synth.SEED_B)
    with capi.Index(pts) as ix:
        md, inl, thr, kept = ix.sor(50, 1.5)
        n_valid = ix.stats()[2]
        start, smd, sinl, sthr, skept, exact = sharding.sor_sharded(lambda s, c: ix.sor_partial(s, c, 50),
                                                                    lambda sums: capi.sor_threshold(sums, n_valid, 50, 1.5),
                                                                    len(pts), 0, 1)
    assert start == 0 and exact and sthr == thr and skept == kept
    assert (_bits(smd) == _bits(md)).all() and (sinl == inl.astype(bool)).all()


def test_a_rank_that_cannot_allocate_returns_nomem_and_the_communicator_stays_usable(gpu):
    """Collectives that cannot hang: every failure of one rank alone (here an injected allocation failure, hook
    pcc_debug_fail_alloc) ends in the status word the ranks agree on before the data collective, so the call returns the
    failing rank's code -- PCC_ERR_NOMEM on the spot -- and the SAME communicator and handles run the next call.  The
    reference maps every failure to a return code (src/comparator.cpp:1123,1134,1179)."""
    tgt = synth.corridor_cloud(40000, synth.SEED_A)
    src = synth.rigid_offset(synth.corridor_cloud(15000, synth.SEED_B), rot_deg=0.4, t=(0.01, 0.0, -0.005))
    L = capi.LIB
    with capi.Comm.from_id(capi.comm_unique_id(), 1, 0, 0) as comm:
        # pcc_index_create_broadcast: the root's first allocation is refused
        L.pcc_debug_fail_alloc(1)
        with pytest.raises(capi.PccError) as e:
            capi.Index.broadcast(comm, 0, tgt)
        L.pcc_debug_fail_alloc(0)
        assert e.value.status == -4 and "injected" in str(e.value)
        with capi.Index.broadcast(comm, 0, tgt) as bx, capi.Index(tgt) as ix:
            # pcc_sor_sharded on a fresh handle: the k-NN rows cannot be allocated
            L.pcc_debug_fail_alloc(1)
            with pytest.raises(capi.PccError) as e:
                bx.sor_sharded(comm, 0, len(tgt), 50, 1.5)
            L.pcc_debug_fail_alloc(0)
            assert e.value.status == -4
            md, inl, thr, kept = ix.sor(50, 1.5)
            smd, sinl, sthr, skept = bx.sor_sharded(comm, 0, len(tgt), 50, 1.5)
            assert (_bits(md) == _bits(smd)).all() and (inl == sinl).all() and thr == sthr and kept == skept
            # pcc_icp_align_sharded: the staging buffer of the source shard is refused
            L.pcc_debug_fail_alloc(1)
            with pytest.raises(capi.PccError) as e:
                bx.icp_align_sharded(comm, src, max_iter=5, fixed=True)
            L.pcc_debug_fail_alloc(0)
            assert e.value.status == -4
            T1, f1, it1, c1 = ix.icp_align(src, max_iter=5, fixed=True)
            T2, f2, it2, c2 = bx.icp_align_sharded(comm, src, max_iter=5, fixed=True)
            assert (T1.view(np.uint32) == T2.view(np.uint32)).all() and f1 == f2 and it1 == it2 == 5 and c1 == c2
    # pcc_comm_create_local: the scalar-exchange word of a rank is refused -> no communicator, an error, nothing leaked
    L.pcc_debug_fail_alloc(1)
    with pytest.raises(capi.PccError) as e:
        capi.Comm.local([0])
    L.pcc_debug_fail_alloc(0)
    assert e.value.status == -4
