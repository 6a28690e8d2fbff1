# dev helper: PMC passes (one per counter group, --kernel-trace only) over any command, averaged for one kernel:
#   bash tools/pmc_cmd.sh <tag> <kernel substring> "<set numbers>" python3 tools/exp_radius.py 2e6 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; KERNEL=$2; PASSES=$3; shift 3
O=gpurun_out/pmc_${TAG}; mkdir -p $O
SETS=(""
 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES"
 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES"
 "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_64B_sum"
 "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum")
for i in $PASSES; do
  rm -rf $O/p$i
  timeout -k 10 300 rocprofv3 --pmc ${SETS[$i]} --kernel-trace --output-format csv -d $O/p$i -- "$@" > $O/p$i.log 2> $O/p$i.err || echo "pass $i failed"
done
python3 tools/nn1_counters.py $O/derived.json "$TAG=$O:$KERNEL"
