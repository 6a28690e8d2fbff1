cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
python -m pytest tests/test_search_gpu.py tests/test_properties_gpu.py tests/test_outliers_gpu.py tests/test_golden_gpu.py tests/test_stress_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "cluster or euclid or properties or golden or stress or room" 2>&1 | tail -3

