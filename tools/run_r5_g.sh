set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g
mkdir -p $O
python -m pytest tests/test_search_gpu.py tests/test_properties_gpu.py tests/test_outliers_gpu.py tests/test_golden_gpu.py tests/test_stress_gpu.py -x -q -m gpu -k "cluster or euclid or properties or golden or stress" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python tools/exp_clusters.py 5e6 3,1 > $O/clusters.log 2>&1; cat $O/clusters.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/profile_ops.py > $O/prof.log 2>&1
grep -h "k_ecc\|k_mp_fine\|k_cs_fine" $(find $O/prof -name "*kernel_stats.csv" | head -1) | awk -F'","' '{print substr($1,1,50), $2, $4}'
