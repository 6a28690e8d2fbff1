# dev helper: PMC passes over tools/exp_nn1.py for one nn1 kernel form: bash tools/pmc_flat.sh <mode> <n> <scene> <set...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MODE=$1; N=$2; SCENE=$3; shift 3
O=gpurun_out/pmcf_${MODE}_${SCENE}; mkdir -p $O
export PCC_NN1_KERNEL=$MODE
SETS=(""
 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES"
 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES"
 "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum")
for i in "$@"; do
  rm -rf $O/p$i
  timeout -k 10 200 rocprofv3 --pmc ${SETS[$i]} --kernel-trace --output-format csv -d $O/p$i -- python3 tools/exp_nn1.py $N $SCENE > $O/p$i.log 2> $O/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 12):
    fs = glob.glob(f'$O/p{i}/*/*counter_collection.csv')
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if '${PMC_KERNEL:-k_grid_nn1}' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        v = v[len(v) // 2:]
        print('mode $MODE $SCENE', k, len(v), f"{sum(v) / len(v):.5g}")
PY
