"""bench.py's N > 1 code on the one-GPU box: the rank code is started as a FRESH child process with a forced
single-rank RCCL group (WORLD_SIZE=1, backend nccl, the group formed even at world size 1), so that everything the
driver's 8-GPU launch will execute -- init_process_group("nccl", device_id=...), the RCCL broadcast of the
reference cloud, the timed step between barriers, all_reduce(MAX) of the time, the per-rank gather,
destroy_process_group -- has run on real hardware before an 8-GPU node sees it (reference:
the single tree per site of src/comparator.cpp:564-577 that every rank replicates)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_rank(extra_args, timeout=600, more_env=None):
    env = dict(os.environ)
    env.update(more_env or {})
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(_free_port()), "PCC_BENCH_BACKEND": "nccl", "PCC_BENCH_FORCE_GROUP": "1",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + extra_args, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_single_rank_rccl_group_runs_the_multi_gpu_path():
    out = _run_rank(["--config", "c2", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-exhaustive"], more_env={"PCC_BENCH_CABI_COMM": "1"})
    cabi = out["extra"]["c_abi_comm"]               # the library's own RCCL communicator next to torch's
    assert cabi["ranks"] == 1 and cabi["shard_results_equal_torch_broadcast_index"] and cabi["create_broadcast_ms"] > 0
    assert out["n_gpus"] == 1                      # the size RCCL's process group reported
    assert out["config"]["backend"] == "nccl"
    assert "RCCL" in out["config"]["parallelism"]
    assert out["broadcast_bytes"] == 1_000_000 * 12 and out["broadcast_ms"] > 0.0   # the broadcast really ran
    assert len(out["per_rank_ms_per_step"]) == 1 and out["per_rank_ms_per_step"][0] > 0.0   # all_reduce gather
    assert out["value"] > 1e8 and out["fallback_queries"] == 0


def test_single_rank_rccl_group_c5_sharding_arithmetic():
    """C5 through the same path: with one rank the shard is all 32M queries (strong scaling's N = 1 anchor)"""
    out = _run_rank(["--config", "c5", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-exhaustive"])
    assert out["scaling"] == "strong" and out["config"]["backend"] == "nccl"
    assert out["config"]["queries_total"] == 32_000_000 and out["config"]["queries_per_gpu"] == 32_000_000
    assert out["config"]["references"] == 8_000_000


def test_headline_leg_shards_the_ten_million_queries_and_ships_the_packed_cloud():
    """the N > 1 headline is north_star's partition: C3's 10M queries SHARDED over the ranks present (strong scaling; with the
    forced one-rank group the shard is all of them), the reference cloud broadcast at 16 B per point although the workload's
    stride is 32, and the line carries the terms a scaling curve has to be read with"""
    out = _run_rank(["--config", "c3", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-exhaustive", "--no-pairs"])
    assert out["scaling"] == "strong" and out["config"]["name"] == "c3"
    assert out["config"]["queries_total"] == 10_000_000 and out["config"]["queries_per_gpu"] == 10_000_000
    assert out["config"]["point_stride_bytes"] == 32 and out["broadcast_bytes"] == 10_000_000 * 16
    st = out["scaling_terms"]
    assert 0 < st["build_ms"] < st["step_ms"] and 0 < st["query_only_ms"] < st["step_ms"]
    assert out["fallback_queries"] == 0 and out["value"] > 1e9


def test_two_ranks_shard_the_headline_queries_between_them():
    """`bench.py --gpus 2` as the driver starts it, on the one-GPU box: two ranks share device 0 (PCC_BENCH_SHARE_DEVICES, gloo --
    RCCL refuses two ranks per device), launched by bench.py itself.  The headline leg must split C3's 10M queries 5M / 5M against
    the broadcast references (strong scaling) and report both ranks' step times; the times themselves mean nothing here."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"PCC_BENCH_SHARE_DEVICES": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--config", "c3", "--steps", "2", "--warmup", "1", "--no-cpu",
                        "--no-exhaustive", "--no-pairs"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["backend"] == "gloo"
    assert out["config"]["queries_total"] == 10_000_000 and out["config"]["queries_per_gpu"] == 5_000_000
    assert out["broadcast_bytes"] == 10_000_000 * 16 and len(out["per_rank_ms_per_step"]) == 2
    assert out["scaling_terms"]["queries_per_gpu"] == 5_000_000 and out["fallback_queries"] == 0


def test_more_ranks_than_gpus_is_refused_with_a_message():
    """`--gpus 8` on a box with fewer devices must fail at once and say why (no hang in the first collective)"""
    import torch
    ndev = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PCC_BENCH_SHARE_DEVICES")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(ndev + 7), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0
    assert f"this node has {ndev} GPU(s)" in p.stderr and "one rank per GPU" in p.stderr
    # and the same guard inside a launcher's environment (the driver's own torch.distributed.run)
    env.update({"WORLD_SIZE": str(ndev + 7), "RANK": "0", "LOCAL_RANK": "0"})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(ndev + 7)], env=env, capture_output=True,
                       text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0 and "one rank per GPU" in p.stderr
