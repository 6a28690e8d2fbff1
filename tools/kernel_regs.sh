#!/bin/bash
# registers / scratch / LDS of the kernels of one HIP source, from the compiler's own remarks (no GPU needed):
#   tools/kernel_regs.sh grid.hip [name-filter]
SRC=pointcloudcomparator_amd/csrc/$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -Iinclude -Ipointcloudcomparator_amd/csrc \
  $EXTRA_HIPFLAGS -Rpass-analysis=kernel-resource-usage -c $SRC -o /dev/null 2>&1 | grep "remark:" | \
  awk '/Function Name:/{n=$(NF-1)} / VGPRs:/{v=$(NF-1)} /ScratchSize/{s=$(NF-1)} /Occupancy/{o=$(NF-1)} /LDS Size/{print n, "vgpr="v, "scratch="s, "occ="o, "lds="$(NF-1)}' | c++filt | sed 's/(.*) / /' | grep -E "${2:-.}"
