# dev helper (round 6): fabric traffic and L2 hit rate of the k = 1 search kernels under the two grid layouts
# (PCC_GRID_AXES=0: x / y / z as in rounds 1-5; -1: by extent).  One rocprofv3 --pmc pass per counter group, --kernel-trace only.
#   gpurun -- 'bash tools/exp_axes_pmc.sh [n] [layer] [xcd_run]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-1e7}; LAYER=${2:-both}; export PCC_XCD_RUN=${3:-32}
O=gpurun_out/axpmc; mkdir -p $O
for AX in 0 -1; do
  export PCC_GRID_AXES=$AX
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    rm -rf $O/ax${AX}_p$i
    timeout -k 10 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/ax${AX}_p$i -- python3 tools/exp_nn1.py $N $LAYER > $O/ax${AX}_p$i.log 2> $O/ax${AX}_p$i.err || { echo "pass $i axes $AX failed"; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, collections
O = "gpurun_out/axpmc"
for ax in ("0", "-1"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in glob.glob(f"{O}/ax{ax}_p*/"):
        for f in glob.glob(d + "**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pcc::", "").split("<")[0]
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== PCC_GRID_AXES={ax} ({'x / y / z (rounds 1-5)' if ax == '0' else 'by extent'})")
    for k in ("k_grid_nn1_flat2", "k_nn1_open_flat", "k_grid_knn_sel", "k_grid_radius_fill_wave"):
        if k not in agg:
            continue
        a = {c: sum(v) / len(v) for c, v in agg[k].items()}
        fetch, write = a.get("FETCH_SIZE", 0) * 1024 * 2, a.get("WRITE_SIZE", 0) * 1024
        hit, miss = a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 0)
        print(f"{k:24s} fetch (x2) {fetch/1e9:6.3f} GB  write {write/1e9:6.3f} GB  L2 hit {hit/(hit+miss+1e-9)*100:5.1f} %  "
              f"EA read requests {a.get('TCC_EA0_RDREQ_sum', 0)/1e6:7.2f} M (32 B: {a.get('TCC_EA0_RDREQ_32B_sum', 0)/1e6:6.2f} M)  "
              f"L1 accesses {a.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0)/1e6:7.1f} M  L1->L2 reads {a.get('TCP_TCC_READ_REQ_sum', 0)/1e6:7.1f} M")
PY
