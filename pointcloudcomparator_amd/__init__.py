"""pointcloudcomparator_amd -- MI355X-native nearest-neighbour engine behind the
PointCloudComparator match/score path.

The product is the C-ABI shared library ``libpcc_nn.so`` (``include/pcc_nn.h``):
hand-written gfx950 HIP kernels for the k-NN / radius searches the reference runs
through ``pcl::KdTreeFLANN`` (reference ``src/comparator.cpp:560-588, 1089-1110,
1520-1549``; ``src/segmentation.cpp:119-131``).  The C++ host mirror of the
reference's functions lives in ``include/pcc/*.hpp``.  This Python package is
plumbing for tests and ``bench.py``: a ctypes binding of the C-ABI
(:mod:`.capi`) and the synthetic cloud generator (:mod:`.synth`).

There is no CPU compute path in here: if ``libpcc_nn.so`` is missing the import
of :mod:`.capi` raises.
"""
from . import capi, synth  # noqa: F401
from .capi import (  # noqa: F401
    Index, PccError, ENGINE_AUTO, ENGINE_BRUTE, ENGINE_GRID, MEM_HOST, MEM_DEVICE,
)
