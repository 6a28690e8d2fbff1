"""dev helper: the last few steps of a rocprofv3 --kernel-trace csv as a timeline (start, duration, stream / queue, kernel) -- to see
which kernels of two streams actually ran side by side.  usage: trace_timeline.py <kernel_trace.csv> [n_last_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void pcc::", "").replace("pcc::", "")[:48]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q={r.get('Queue_Id', '?'):>3} st={r.get('Stream_Id', '?'):>3}  {'overlaps' if s < prev_end else '        '}  {name}")
    prev_end = max(prev_end, e)
