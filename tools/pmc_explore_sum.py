"""dev helper: average the counters tools/pmc_explore.sh collected for k_grid_nn1 (second half of the launches)."""
import csv, glob, collections, sys
for i in range(1, 11):
    fs = glob.glob(f'gpurun_out/pmcx/p{i}/*/*counter_collection.csv')
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if (sys.argv[1] if len(sys.argv) > 1 else 'k_grid_nn1') in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        v = v[len(v) // 2:]
        print(i, k, len(v), f"{sum(v) / len(v):.4g}")
