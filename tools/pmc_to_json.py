#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, collected in separate runs with
--kernel-trace only) into profiles/<tag>_pmc_traffic.json, the file bench.py reads `traffic` from.

Units and corrections as MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of coalesced streaming reads, so the
read side is doubled; WRITE_SIZE is exact.  Calibrated in this repo's own access patterns on
kernels of known byte counts (k_pack: 12 MB read / 16 MB written per 1M points, k_unpack:
24 MB read / 8 MB written per 1M queries): both read sides report exactly 1/2, both write
sides exactly 1.  For the gather-heavy k_grid_nn1 the doubling is an upper bound (64-byte
requests would be counted in full); raw values are kept next to the corrected ones.

usage: pmc_to_json.py <tag> <workload-key> <fetch_dir> <write_dir>
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import source_digest  # noqa: E402  (the digest bench.py compares against: kernel sources this profile describes)


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("pcc::", "")
    return n.split("<")[0]


def load(d, counter):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, wl, fdir, wdir = sys.argv[1:5]
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    out_file = ROOT / "profiles" / f"{tag}_pmc_traffic.json"
    data = json.loads(out_file.read_text()) if out_file.exists() else {}
    rows = {}
    detail = {}
    for k in sorted(set(fetch) | set(write)):
        if not (k.startswith("k_")):
            continue
        f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
        rows[k] = (2.0 * f + w) * 1024.0  # bytes per launch, read side doubled (gfx950 correction)
        detail[k] = {"FETCH_SIZE_KiB_raw_avg": f, "WRITE_SIZE_KiB_avg": w, "launches_fetch_pass": len(fetch.get(k, [])),
                     "launches_write_pass": len(write.get(k, [])), "hbm_bytes_per_launch_corrected": rows[k],
                     "hbm_bytes_per_launch_uncorrected": (f + w) * 1024.0}
    data[wl] = rows
    data[wl + "_detail"] = detail
    data["source_digest"] = source_digest()
    out_file.write_text(json.dumps(data, indent=1))
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
