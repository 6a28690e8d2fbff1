"""A short leg of the randomised parity campaign (tools/fuzz_gpu.py: scene families x query placements x operations,
every comparison bit-exact against the oracle).  The long runs are done by hand through gpurun (tests)."""
import importlib.util
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def _run(tool, argv, monkeypatch):
    path = Path(__file__).resolve().parent.parent / "tools" / tool
    spec = importlib.util.spec_from_file_location(tool[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", [tool] + argv)
    return mod.main()


def test_fuzz_campaign_short_leg(gpu, monkeypatch):
    assert _run("fuzz_gpu.py", ["--seconds", "20", "--seed", "7", "--max-refs", "20000"], monkeypatch) == 0


def test_fuzz_sequences_short_leg(gpu, monkeypatch):
    """one long-lived handle, random rebuilds and searches of every kind in turn (tools/fuzz_seq_gpu.py)"""
    assert _run("fuzz_seq_gpu.py", ["--seconds", "15", "--seed", "11", "--max-refs", "20000", "--no-icp-align"], monkeypatch) == 0
