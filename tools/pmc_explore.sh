# dev helper: exploratory PMC passes over tools/exp_nn1.py (k_grid_nn1 at 1M x 1M); sets chosen by number
#   gpurun -- 'bash tools/pmc_explore.sh 4 5'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcx; mkdir -p $O
SETS=(""
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"
 "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
 "SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
 "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_IFETCH")
for i in "$@"; do
  rm -rf $O/p$i
  timeout -k 10 200 rocprofv3 --pmc ${SETS[$i]} --kernel-trace --output-format csv -d $O/p$i -- python3 tools/exp_nn1.py ${PMCX_N:-1e6} ${PMCX_SCENE:-both} > $O/p$i.log 2> $O/p$i.err || echo "pass $i failed"
  echo "pass $i done"
done
