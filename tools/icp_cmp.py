"""dev helper: bits of the ICP result (compare runs with PCC_ICP_DEVICE_LOOP / PCC_ICP_WARM = 0 and 1)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from pointcloudcomparator_amd import capi, synth
M = 500_000
ref = synth.corridor_cloud(M, synth.SEED_A)
src = synth.corridor_cloud(M, synth.SEED_B)
c, s_ = np.cos(np.radians(2.0)), np.sin(np.radians(2.0))
R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], dtype=np.float32)
src = (src @ R.T + np.array([0.03, -0.02, 0.01], dtype=np.float32)).astype(np.float32)
ix = capi.Index(torch.from_numpy(ref).cuda())
fixed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
r = ix.icp_align(torch.from_numpy(src).cuda(), max_iter=iters, fixed=fixed)
T, fit, it, conv = r
print(it, conv, repr(fit))
print(T.view(np.uint32).ravel().tolist())
