# per-launch kernel trace of the C4 (ICP) configuration: which kernel costs what in which pass
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-c4trace}
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --config c4 --no-cpu --no-pairs > $O/bench.json 2> $O/err.log
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 - "$O" <<'PY'
import csv, sys, glob, collections
O = sys.argv[1]
f = [p for p in glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)][0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("pcc::", "").replace("void ", "")
    per[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(O + "/per_launch.txt", "w") as out:
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        last = v[-51:]
        out.write("%-60s n=%d total=%.0f us; last call's launches: %s\n" % (k[:60], len(v), sum(v), " ".join("%.0f" % x for x in last)))
PY
cat $O/per_launch.txt | cut -c1-700
