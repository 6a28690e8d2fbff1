#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left under gpurun_out/<tag>/summary/ into profiles/ (tracked, judged).
usage: collect_profiles.py <tag>"""
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = ROOT / "gpurun_out" / tag / "summary"
for f in sorted(src.iterdir()):
    shutil.copy(f, ROOT / "profiles" / f.name)
    print("profiles/" + f.name)
