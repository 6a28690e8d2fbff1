import sys; sys.path.insert(0,'.')
import torch
from pointcloudcomparator_amd import capi, synth
obj = synth.corridor_cloud(5_000_000, synth.SEED_A, layer="objects")
ix = capi.Index(torch.from_numpy(obj).cuda())
for _ in range(3): r = ix.euclidean_clusters(0.05, 100, 250000)
print(r[1])
