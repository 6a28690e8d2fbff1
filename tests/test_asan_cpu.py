"""`make asan`: the host-side code -- oracle/pcc_oracle.c, csrc/flann_tree.hpp (the PCC_TIES_FLANN tree), rigid_solve.hpp,
plane_fit.hpp, host/ply_io.hpp -- under AddressSanitizer + UndefinedBehaviorSanitizer with a CPU-only driver
(tests/cpp/asan_driver.cpp: degenerate clouds, empty and deep trees, degenerate sums, truncated and corrupt PLY files).
CPU build only; sanitizers never run on the GPU box."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_make_asan_is_clean():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "asan"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "asan driver ok" in r.stdout and "asan: clean" in r.stdout
