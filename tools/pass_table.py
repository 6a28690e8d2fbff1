#!/usr/bin/env python3
"""Per-pass table of the C3 step with the query staging BEHIND the build (PCC_OVERLAP_PREP = 0: no two kernels of the step
overlap, so a kernel's rocprofv3 duration is its own): average duration, HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE
passes (read side doubled as MI355X_MICROARCH.md prescribes for gfx950), achieved TB/s and its share of the 6.29 TB/s a copy
kernel reaches on this chip.
usage: pass_table.py <kernel_stats.csv> <fetch_dir> <write_dir> <out.txt>"""
import collections, csv, glob, sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("pcc::", "")
    return n.replace("HIP_vector_type<float, 4u>", "float4").replace("HIP_vector_type<unsigned int, 2u>", "uint2")


def pmc(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * 1024.0 for k, v in acc.items()}


stats, fdir, wdir, out = sys.argv[1:5]
fetch, write = pmc(fdir, "FETCH_SIZE"), pmc(wdir, "WRITE_SIZE")
rows = []
for r in csv.DictReader(open(stats)):
    k = short(r["Name"])
    if not k.startswith("k_"):
        continue
    us = float(r["AverageNs"]) / 1e3
    b = 2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)
    rows.append((us * int(r["Calls"]), k, int(r["Calls"]), us, fetch.get(k, 0.0) * 2, write.get(k, 0.0), b))
rows.sort(reverse=True)
with open(out, "w") as f:
    f.write("C3 step, one stream (PCC_OVERLAP_PREP = 0): rocprofv3 --kernel-trace --stats durations, --pmc FETCH_SIZE / WRITE_SIZE bytes per launch\n")
    f.write(f"{'kernel':58s} {'calls':>5s} {'avg us':>9s} {'read MB':>9s} {'write MB':>9s} {'TB/s':>6s} {'% of 6.29':>9s}\n")
    for _, k, calls, us, rd, wr, b in rows:
        tbs = b / (us * 1e-6) / 1e12 if us > 0 else 0.0
        f.write(f"{k:58s} {calls:5d} {us:9.1f} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {tbs:6.2f} {tbs / 6.29 * 100:8.1f}%\n")
print(open(out).read())
