#!/bin/bash
# development: builds of the library with parts of k_grid_nn1_flat2 compiled out (PCC_ABLATE bits, grid.hip) -- wrong results,
# the kernel time says what the part costs.  usage: tools/exp_ablate.sh 1 3 4 8 ...  -> pointcloudcomparator_amd/lib/libpcc_nn_abl<N>.so
set -e
for a in "$@"; do
  mkdir -p build/abl$a
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -Iinclude -Ipointcloudcomparator_amd/csrc \
     -DPCC_ABLATE=$a $ABL_FLAGS -c pointcloudcomparator_amd/csrc/grid.hip -o build/abl$a/grid.o
  OBJS=$(ls build/*.o | grep -v "/grid.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pointcloudcomparator_amd/lib/libpcc_nn_abl$a.so build/abl$a/grid.o $OBJS -ldl
  echo "built abl$a"
done
