# dev helper: the four k-NN reference points of DESIGN 4.4 (1M self query, corridor / room scan, K = 51 / 100), run 1 against run 16
for sc in corridor room; do for K in 51 100; do ROUNDS=${ROUNDS:-3} python tools/exp_knn_ab.py $sc 1e6 $K "knn_run=1" "knn_run=16" 2>&1 | grep -v amdgpu.ids; done; done
