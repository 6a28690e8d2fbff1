"""Runs the C++ host-mirror self test (tests/cpp/test_host_mirror.cpp, include/pcc/*.hpp):
the reference-shaped classes and functions on top of the C-ABI, used the way the reference's
call sites use PCL."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_cpp_host_mirror(gpu):
    exe = ROOT / "build" / "test_host_mirror"
    if not exe.exists():
        subprocess.check_call(["make", "hosttest"], cwd=ROOT)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host mirror ok" in r.stdout
    assert "has converged:1" in r.stdout  # performICP prints the reference's line


def test_lane_ops_on_the_device(gpu):
    """csrc/lane_ops.hpp: DPP / permlane exchanges and scans against their definitions, every lane"""
    exe = ROOT / "build" / "test_lane_ops"
    if not exe.exists():
        subprocess.check_call(["make", "hosttest"], cwd=ROOT)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "lane ops: 0 mismatches" in r.stdout
