set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a
mkdir -p $O
python -m pytest tests/test_comm_gpu.py tests/test_bench_gpu.py tests/test_rgb_gpu.py tests/test_nn1_kernels_gpu.py tests/test_nn1_gpu.py tests/test_golden_gpu.py -x -q -m gpu > $O/pytest_a.log 2>&1 || { tail -30 $O/pytest_a.log; exit 1; }
tail -3 $O/pytest_a.log
python -m pytest tests/test_search_gpu.py -x -q -m gpu -k "radius or icp" > $O/pytest_b.log 2>&1 || { tail -30 $O/pytest_b.log; exit 1; }
tail -3 $O/pytest_b.log
OPTS="nn1_single_span=0;nn1_single_span=8;nn1_single_span=16;nn1_single_span=24;nn1_single_span=48;nn1_single_span=128" python tools/exp_r5.py 1e7 1e6 > $O/single.log 2>&1
cat $O/single.log
for a in 1 3 4 8 16; do
  PCC_LIB=$GRAFT_REPO_ROOT/pointcloudcomparator_amd/lib/libpcc_nn_abl$a.so OPTS="nn1_single_span=0;nn1_single_span=24" python tools/exp_r5.py 1e7 > $O/abl$a.log 2>&1
  echo "abl$a"; cat $O/abl$a.log
done
PCC_NN1_SINGLE_SPAN=0 python tools/exp_icp.py > $O/icp_span0.log 2>&1; tail -1 $O/icp_span0.log
python tools/exp_icp.py > $O/icp_span24.log 2>&1; tail -1 $O/icp_span24.log
