set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d
mkdir -p $O
python -m pytest tests/test_nn1_gpu.py tests/test_nn1_kernels_gpu.py tests/test_sac_gpu.py tests/test_normals_gpu.py tests/test_golden_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python -m pytest tests/test_search_gpu.py -x -q -m gpu -k "voxel or cluster or sor" >> $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
OPTS="sort_stage1=1;sort_stage1=0" python tools/exp_r5.py 1e7 > $O/c3.log 2>&1; grep "^n=" $O/c3.log
python bench.py --config c3 --no-cpu --no-pairs --no-exhaustive > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; d=json.load(open('$O/bench_c3.json')); print('ms_per_step', d['ms_per_step'], 'build', d['build_ms'], 'sort', d['query_sort_ms'], 'call', d['search_call_ms'])"
