"""oracle -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's NN path.

PARITY UNPINNED: the reference ships no tests or golden vectors and its arithmetic
lives in PCL 1.7 / FLANN 1.8.4, which are absent here.  Only tests/,
bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this package; the
product (pointcloudcomparator_amd, libpcc_nn) never does.
"""
from .pyoracle import *  # noqa: F401,F403
