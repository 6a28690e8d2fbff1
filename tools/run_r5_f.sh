set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f
mkdir -p $O
PCC_KNN_RUN=8 python -m pytest tests/test_search_gpu.py tests/test_normals_gpu.py tests/test_properties_gpu.py -x -q -m gpu -k "knn or sor or normals or region or properties" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for scene in corridor room; do
 for r in 0 4 8 16 0 8; do
  SCENE=$scene PCC_KNN_RUN=$r python tools/exp_knn.py 1e6 51 100 2>&1 | grep "K=" | sed "s/^/run=$r /"
 done
done | tee $O/knn_run.log
