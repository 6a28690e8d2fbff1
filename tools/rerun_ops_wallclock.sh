# dev helper: only the traced operation run of tools/profile_round.sh part B (ops wall clock + kernel stats)
set -e
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; S=$O/summary; mkdir -p $S
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ops2 -- python3 tools/profile_ops.py > $S/${TAG}_ops_wallclock.txt 2> $O/ops.err
cp $(find $O/ops2 -name "*kernel_stats.csv" | head -1) $S/${TAG}_ops_kernel_stats.csv
python3 tools/exp_ties.py 1e6 2>/dev/null | tail -1 >> $S/${TAG}_ops_wallclock.txt
cat $S/${TAG}_ops_wallclock.txt
