# The per-operation part of tools/profile_round.sh alone (kernels outside the k = 1 path changed; the C2-C5 traces and
# the PMC traffic passes stand): bash tools/profile_ops_only.sh r03, then python3 tools/collect_profiles.py r03
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
S=$O/summary
mkdir -p $S
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ops -- python3 tools/profile_ops.py > $S/${TAG}_ops_wallclock.txt 2> $O/ops.err
cp $(find $O/ops -name "*kernel_stats.csv" | head -1) $S/${TAG}_ops_kernel_stats.csv
python3 tools/exp_ties.py 1e6 2>/dev/null | tail -1 >> $S/${TAG}_ops_wallclock.txt
cat $S/${TAG}_ops_wallclock.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/opsr -- python3 tools/ops_roofline.py run $O/ops_run.json > $O/opsr.log 2> $O/opsr.err
python3 tools/ops_roofline.py merge $O/ops_run.json $(find $O/opsr -name "*kernel_stats.csv" | head -1) $S/${TAG}_ops_roofline.json
