"""CPU-side host logic: the report writer of the CLI (tests/cpp/test_report.cpp, no device call) and bench.py's own
rank launcher (the command it would start, and that it starts it before anything touches the GPU)."""
import subprocess
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_report_writer_sections_and_descriptor_files(tmp_path):
    exe = ROOT / "build" / "test_report"
    if not exe.exists():
        subprocess.check_call(["make", "build/test_report"], cwd=ROOT)
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "report ok" in r.stdout


def test_bench_launches_its_own_ranks(monkeypatch):
    sys.path.insert(0, str(ROOT))
    import bench
    calls = []
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    # more ranks than GPUs (none here) is refused with a message, before anything is launched ...
    monkeypatch.delenv("PCC_BENCH_SHARE_DEVICES", raising=False)
    with pytest.raises(SystemExit) as refused:
        bench.main()
    assert "one rank per GPU" in str(refused.value.code) and not calls
    # ... unless the shared-device rehearsal is asked for
    monkeypatch.setenv("PCC_BENCH_SHARE_DEVICES", "1")
    # main() must hand over to the child ranks before importing torch / the library
    imported_before = "pointcloudcomparator_amd.capi" in sys.modules
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert ("pointcloudcomparator_amd.capi" in sys.modules) == imported_before


def test_bench_configs_match_baseline_json():
    import json
    sys.path.insert(0, str(ROOT))
    import bench
    base = json.loads((ROOT / "BASELINE.json").read_text())
    assert len(base["configs"]) == 5
    assert bench.CONFIGS["c2"][:2] == (1_000_000, 1_000_000)
    assert bench.CONFIGS["c3"][:3] == (10_000_000, 10_000_000, 8)      # XYZRGB: 32-byte stride
    assert bench.CONFIGS["c4"][:2] == (2_000_000, 2_000_000)
    assert bench.CONFIGS["c5"][0] == 8_000_000 and bench.C5_TOTAL_QUERIES == 32_000_000
    assert bench.C5_TOTAL_QUERIES // 8 == bench.CONFIGS["c5"][1]       # 8 shards of 4M


def test_bench_line_helpers(monkeypatch):
    """what bench.py does to its own line and around the pair-counting child process, without a GPU: floats rounded to six
    digits (NaN / inf -> null: the line must stay valid JSON), the headline totals, the profiler guard"""
    import json
    import math
    sys.path.insert(0, str(ROOT))
    import bench
    out = bench.compact({"a": 1.23456789, "b": [float("nan"), float("inf"), 3, "x", {"c": 1e-12 / 3}], "n": 10_000_000, "t": True})
    assert out == {"a": 1.23457, "b": [None, None, 3, "x", {"c": 3.33333e-13}], "n": 10_000_000, "t": True}
    assert json.loads(json.dumps(out)) == out and not any(isinstance(v, float) and math.isnan(v) for v in out["b"][:2] if v is not None)
    assert bench.C3_TOTAL_QUERIES == bench.CONFIGS["c3"][1] == 10_000_000 and bench.CONFIGS["c5_shard"][:2] == (8_000_000, 4_000_000)
    assert bench.PROJECTION_G == (1, 2, 4, 8)
    for k in list(os.environ):
        if k.startswith(("ROCPROFILER", "ROCPROF_", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    assert not bench.under_profiler()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.under_profiler() and bench.count_pairs(["c2"]) == {}     # no child process under a profiler's preload
    monkeypatch.delenv("LD_PRELOAD")
    monkeypatch.setenv("ROCPROFILER_SOMETHING", "1")
    assert bench.under_profiler()


def _flann_tree_answers(tmp_path, pts, qs, rule, threads):
    """indices and d2 bits from the PRODUCT's tree (csrc/flann_tree.hpp: iterative build, flat nodes, explicit-stack
    walk) through its g++-built self-test program"""
    import subprocess
    exe = ROOT / "build" / "test_flann_tree"
    src = ROOT / "tests" / "cpp" / "test_flann_tree.cpp"
    hdr = ROOT / "pointcloudcomparator_amd" / "csrc" / "flann_tree.hpp"
    if not exe.exists() or exe.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        exe.parent.mkdir(exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-pthread",
                               f"-I{hdr.parent}", str(src), "-o", str(exe)])
    pf, qf = tmp_path / "p.bin", tmp_path / "q.bin"
    np.ascontiguousarray(pts, np.float32).tofile(pf)
    np.ascontiguousarray(qs, np.float32).tofile(qf)
    out = subprocess.check_output([str(exe), str(pf), str(qf), str(rule), str(threads)], text=True)
    rows = [l.split() for l in out.splitlines() if not l.startswith("#")]
    handed = [l for l in out.splitlines() if l.startswith("# short walks")][0].split()
    full, total = int(handed[-3]), int(handed[-1])
    idx = np.array([int(r[0]) for r in rows], np.int64)
    tied = np.array([int(r[2]) for r in rows], np.int64)
    # the short walk (minimum distance known: what the device runs for tied queries) names the same reference as the full one
    assert (tied == idx).all(), np.nonzero(tied != idx)[0][:5]
    assert full <= total // 50, (full, total)  # and it is the short walk that answers, not its fallback
    return idx, np.array([int(r[1]) for r in rows], np.uint32)


@pytest.mark.parametrize("rule", [0, 1, 2])
def test_flann_tree_of_the_product_matches_the_oracle_tree(tmp_path, rule):
    """PCC_TIES_FLANN: the product's kd-tree (iterative build over an explicit stack, flat depth-first node array, top
    levels forked over threads, explicit-stack walk shared with the device kernel) names the same reference as the
    oracle's recursive kd-tree for every query, under each of the three split rules -- on clouds made of exact ties
    (duplicates, lattices, a plane), with non-finite points, and with 1 and 8 build threads.  Two restatements of the
    same recollection of FLANN agreeing is consistency, not verification against FLANN."""
    import oracle
    rng = np.random.default_rng(40 + rule)
    base = rng.random((6000, 3), dtype=np.float32)
    dup = np.concatenate([base, base[::-1], base[:1500]])
    lattice = (rng.integers(0, 14, (20000, 3)) * np.float32(0.25)).astype(np.float32)
    plane = rng.random((15000, 3), dtype=np.float32)
    plane[:, 2] = 0.5
    plane[::3] = np.round(plane[::3] * 8) / 8
    holes = rng.random((9000, 3), dtype=np.float32)
    holes[::7, 1] = np.nan
    holes[5::11, 0] = np.inf
    skew = (rng.random((12000, 3)) ** 6 * 100).astype(np.float32)  # deep, lopsided trees
    tiny = rng.random((9, 3), dtype=np.float32)
    oracle.set_split_rule(rule)
    try:
        for name, pts in (("dup", dup), ("lattice", lattice), ("plane", plane), ("holes", holes), ("skew", skew), ("tiny", tiny)):
            qs = np.concatenate([pts[rng.integers(0, len(pts), 1500)], (rng.random((1500, 3)) * 1.2 - 0.1).astype(np.float32),
                                 (rng.integers(0, 28, (1000, 3)) * np.float32(0.125)).astype(np.float32)])
            qs = qs[np.isfinite(qs).all(1)]
            tree = oracle.KdTree(pts)
            oi, od = tree.nn1_batch(qs)
            for threads in (1, 8):
                pi, pd = _flann_tree_answers(tmp_path, pts, qs, rule, threads)
                assert (pd == od.view(np.uint32)).all(), (name, rule, threads)
                assert (pi == oi).all(), (name, rule, threads, np.nonzero(pi != oi)[0][:5])
    finally:
        oracle.set_split_rule(0)
