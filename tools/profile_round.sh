set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01h   # remove a previous local gpurun_out/r01h first: gpurun merges, it does not replace
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1 && tail -2 $O/gpu_tests.log
python bench.py --steps 20 --warmup 3 > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $O/trace_bench.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 8 --warmup 2 --no-cpu > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 8 --warmup 2 --no-cpu > $O/pmc_write_bench.json 2> $O/pmc_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ops -- python3 tools/profile_ops.py > $O/ops.log 2> $O/ops.err
cat $O/ops.log
