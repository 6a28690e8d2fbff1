"""CPU-side host logic: the report writer of the CLI (tests/cpp/test_report.cpp, no device call) and bench.py's own
rank launcher (the command it would start, and that it starts it before anything touches the GPU)."""
import subprocess
import sys
from pathlib import Path

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
