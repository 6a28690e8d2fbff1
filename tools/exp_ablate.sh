#!/bin/bash
# development: builds of the library with parts of k_grid_nn1_flat2 compiled out (PCC_ABLATE bits) -- wrong results, the kernel time
# says what the part costs (profiles/r05_nn1_ablation.txt).  The ablation branches are NOT in the product source: they are
# tools/exp_ablate.patch, applied here to a scratch copy of csrc/grid.hip.
#   bits: 1 drops pass 1, 2 pass 0, 4 every drain loop, 8 the open-lane listing (every lane counts as resolved), 32 stores the key at
#   out[t] instead of out[order[t]], 64 reads q[t] instead of q[order[t]], 128 drops the row-bound gathers.
# usage: tools/exp_ablate.sh 1 3 4 8 ...  -> build/abl<N>/libpcc_nn_abl<N>.so (select with PCC_LIB)
set -e
for a in "$@"; do
  mkdir -p build/abl$a
  cp pointcloudcomparator_amd/csrc/grid.hip build/abl$a/grid.hip
  patch -s build/abl$a/grid.hip tools/exp_ablate.patch
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -Iinclude -Ipointcloudcomparator_amd/csrc \
     -DPCC_ABLATE=$a $ABL_FLAGS -c build/abl$a/grid.hip -o build/abl$a/grid.o
  OBJS=$(ls build/*.o | grep -v "/grid.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl$a/libpcc_nn_abl$a.so build/abl$a/grid.o $OBJS -ldl
  echo "built build/abl$a/libpcc_nn_abl$a.so"
done
