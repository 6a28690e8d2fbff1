set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e
mkdir -p $O
python -m pytest tests/test_normals_gpu.py tests/test_golden_gpu.py tests/test_sac_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "normals or ransac" >> $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python tools/profile_ops.py > $O/ops_wallclock.txt 2>&1; cat $O/ops_wallclock.txt
