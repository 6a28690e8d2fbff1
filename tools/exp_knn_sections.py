"""dev helper (round 6): where the waves of k_grid_knn_sel spend their time.  Needs the library built with
tools/exp_knn_sections.patch applied (tools/exp_knn_sections.sh does that into lib/libpcc_nn_sect.so and restores the source):
s_memtime deltas per section, summed over the waves.  Shares of WAVE time (stalls included), not of instructions.
usage: PCC_LIB=.../libpcc_nn_sect.so exp_knn_sections.py <corridor|room> <n> <K>"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

scene, n, K = sys.argv[1], int(float(sys.argv[2])), int(sys.argv[3])
a = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A) if scene == "corridor" else synth.room_cloud(n, synth.SEED_A)).cuda()
ix = capi.Index(a, auto_sync=False)
lib = ctypes.CDLL(os.environ["PCC_LIB"])
buf = (ctypes.c_ulonglong * 32)()
NAMES = {0: "header (query, frame, cell)", 10: "bound: separation, box", 1: "bound: ball table", 2: "bound: walk, keep below",
         3: "bound: accept / second stage", 4: "full: sizing rounds", 5: "full: pass-1 table, walk, buckets",
         6: "full: K-th bucket, compaction", 7: "full: pass 2 (outside the cube)", 8: "sort", 9: "emit, K-th"}
for _ in range(2):
    ix.knn(a, K)
ix.sync()
lib.pcc_debug_knn_sections(buf)
ix.knn(a, K)
ix.sync()
assert lib.pcc_debug_knn_sections(buf) == 0
cyc, cnt = list(buf)[:12], list(buf)[16:28]
tot = sum(cyc)
print(f"{scene} {n} K={K}: wave time by section (one call; {cnt[0]} queries)")
for i in (0, 10, 1, 2, 3, 4, 5, 6, 7, 8, 9):
    if cnt[i]:
        print(f"  {NAMES[i]:36s} {100.0 * cyc[i] / tot:5.1f} %   entered {cnt[i]:8d} times ({100.0 * cnt[i] / max(cnt[0], 1):5.1f} % of queries)   {cyc[i] / cnt[i]:8.0f} ticks each")
if cnt[4]:
    print(f"  full path: cube candidates {cyc[11] / cnt[4]:6.1f} per query, final half-width {cnt[11] / cnt[4]:4.2f} on average")
