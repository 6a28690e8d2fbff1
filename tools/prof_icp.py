import sys; sys.path.insert(0,'.')
import torch
from pointcloudcomparator_amd import capi, synth
m = 2_000_000
tgt = torch.from_numpy(synth.corridor_cloud(m, synth.SEED_A)).cuda()
src = torch.from_numpy(synth.rigid_offset(synth.corridor_cloud(m, synth.SEED_B))).cuda()
ix = capi.Index(tgt)
for _ in range(2): r = ix.icp_align(src, max_iter=50, fixed=True)
print(r[2], ix.stats())
