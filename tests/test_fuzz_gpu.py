"""A short leg of the randomised parity campaign (tools/fuzz_gpu.py: scene families x query placements x operations,
every comparison bit-exact against the oracle).  The long runs are done by hand through gpurun (tests)."""
import importlib.util
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_campaign_short_leg(gpu, monkeypatch):
    path = Path(__file__).resolve().parent.parent / "tools" / "fuzz_gpu.py"
    spec = importlib.util.spec_from_file_location("fuzz_gpu", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["fuzz_gpu.py", "--seconds", "20", "--seed", "7", "--max-refs", "20000"])
    assert mod.main() == 0
