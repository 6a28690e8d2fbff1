# Regenerates everything under profiles/ for one round on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02'
# then copy gpurun_out/<tag>/summary/* into profiles/ (tools/collect_profiles.py does that).
# One rocprofv3 pass per kind: --kernel-trace --stats for durations, --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate
# passes (MI355X_MICROARCH.md: they do not fit one pass), never combined with other trace domains.
set -e
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
S=$O/summary
mkdir -p $S
# PART=A: traces, PMC traffic, the one-stream pass table, the default bench; PART=B: operations, rooflines, counters (a gpurun call is
# limited to 20 minutes: the two halves are separate calls; default: both)
PART=${PART:-AB}
if [ "$PART" != B ]; then
for c in c2 c3 c4 c5 c5_shard; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -- python3 bench.py --config $c --no-cpu --no-pairs > $S/${TAG}_${c}_trace_bench.json 2> $O/trace_$c.err
  cp $(find $O/trace_$c -name "*kernel_stats.csv" | head -1) $S/${TAG}_${c}_kernel_stats.csv
  echo "trace $c done"
done
for c in c2 c3 c4 c5 c5_shard; do
  for k in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $O/pmc_${c}_$k -- python3 bench.py --config $c --steps 6 --warmup 2 --no-cpu --no-exhaustive --no-pairs > $O/pmc_${c}_$k.json 2> $O/pmc_${c}_$k.err
  done
  python3 tools/pmc_to_json.py $TAG $c $O/pmc_${c}_FETCH_SIZE $O/pmc_${c}_WRITE_SIZE > $O/pmc_$c.log
  echo "pmc $c done"
done
cp profiles/${TAG}_pmc_traffic.json $S/
# the C3 step on ONE stream (query staging behind the build): per-pass durations and bytes that can be ranked
export PCC_OVERLAP_PREP=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3_serial -- python3 bench.py --config c3 --no-cpu --no-pairs --no-exhaustive > $O/trace_c3_serial.json 2> $O/trace_c3_serial.err
cp $(find $O/trace_c3_serial -name "*kernel_stats.csv" | head -1) $S/${TAG}_c3_serial_kernel_stats.csv
for k in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $O/pmc_c3_serial_$k -- python3 bench.py --config c3 --steps 6 --warmup 2 --no-cpu --no-exhaustive --no-pairs > $O/pmc_c3_serial_$k.json 2> $O/pmc_c3_serial_$k.err
done
unset PCC_OVERLAP_PREP
python3 tools/pass_table.py $S/${TAG}_c3_serial_kernel_stats.csv $O/pmc_c3_serial_FETCH_SIZE $O/pmc_c3_serial_WRITE_SIZE $S/${TAG}_c3_serial_pass_table.txt > $O/pass_table.log
echo "serial trace done"
# the unprofiled default run last: its roofline.traffic is read from the PMC file written just above
python3 bench.py > $S/${TAG}_bench_default.json 2> $O/bench_default.err
echo "default bench done"
fi
[ "$PART" = A ] && exit 0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ops -- python3 tools/profile_ops.py > $S/${TAG}_ops_wallclock.txt 2> $O/ops.err
cp $(find $O/ops -name "*kernel_stats.csv" | head -1) $S/${TAG}_ops_kernel_stats.csv
python3 tools/exp_ties.py 1e6 2>/dev/null | tail -1 >> $S/${TAG}_ops_wallclock.txt
cat $S/${TAG}_ops_wallclock.txt
# per-operation rooflines (algorithmic bytes of SURVEY 8d against the HBM roof, rocprofv3 averages per kernel)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/opsr -- python3 tools/ops_roofline.py run $O/ops_run.json > $O/opsr.log 2> $O/opsr.err
# pair counts of the same operations by the profiling build (VALU column of the table)
PCC_LIB=$GRAFT_REPO_ROOT/pointcloudcomparator_amd/lib/libpcc_nn_prof.so python3 tools/ops_roofline.py run $S/${TAG}_ops_pairs.json > $O/opsp.log 2> $O/opsp.err
python3 tools/ops_roofline.py merge $O/ops_run.json $(find $O/opsr -name "*kernel_stats.csv" | head -1) $S/${TAG}_ops_roofline.json $S/${TAG}_ops_pairs.json
# counters of the k = 1 kernels (one --pmc pass per group), 10M x 10M and 1M x 1M corridor scene
rm -rf gpurun_out/pmcf_1_both
bash tools/pmc_flat.sh 1 1e7 both 1 2 3 4 > $O/pmcf_c3.log 2>&1
mv gpurun_out/pmcf_1_both $O/pmcf_c3
bash tools/pmc_flat.sh 1 1e6 both 1 2 3 4 > $O/pmcf_c2.log 2>&1
mv gpurun_out/pmcf_1_both $O/pmcf_c2
bash tools/pmc_cmd.sh knn51 k_grid_knn_sel "1 2 3 4" python3 tools/exp_knn.py 1e6 51 > $O/pmc_knn.log 2>&1
cp gpurun_out/pmc_knn51/derived.json $S/${TAG}_knn_counters.json
python3 tools/nn1_counters.py $S/${TAG}_nn1_counters.json c3_flat2=$O/pmcf_c3:k_grid_nn1_flat2 c3_open=$O/pmcf_c3:k_nn1_open c2_flat2=$O/pmcf_c2:k_grid_nn1_flat2 > $O/nn1_counters.log
