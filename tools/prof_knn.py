import sys; sys.path.insert(0,'.')
import torch
from pointcloudcomparator_amd import capi, synth
n = 1_000_000
ta = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A)).cuda()
ix = capi.Index(ta)
for _ in range(3):
    ix.knn(ta, 51); ix.sync()
for _ in range(2):
    ix.knn(ta, 100); ix.sync()
