"""Distances evaluated per call by the pruned kernels, counted by the PROFILING build (libpcc_nn_prof.so, `make prof`:
pcc_index_stats[4]).  The counts are properties of (workload, kernel form), not of the run: bench.py starts this as a
child process (PCC_LIB pointing at the profiling library) outside its timed region and prices the timed kernels of the
normal library with them: pairs/s, and 9 VALU operations per pair against the non-FMA fp32 roof.
usage: PCC_LIB=pointcloudcomparator_amd/lib/libpcc_nn_prof.so python3 tools/count_pairs.py c2|c3|c4|c5|c5_shard [...]
prints one JSON object {config: {...}}"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

SIZES = {"c1": (10_000, 10_000), "c2": (1_000_000, 1_000_000), "c3": (10_000_000, 10_000_000), "c4": (2_000_000, 2_000_000),
         "c5_shard": (8_000_000, 4_000_000), "c5": (8_000_000, 32_000_000)}


def cloud(n, seed, chunk=4_000_000):
    parts = [synth.corridor_cloud(min(chunk, n - o), seed, start=o) for o in range(0, n, chunk)]
    return parts[0] if len(parts) == 1 else np.concatenate(parts)


def main():
    assert capi.LIB.pcc_counts_pairs() == 1, "needs the profiling build: PCC_LIB=.../libpcc_nn_prof.so"
    out = {}
    for cfg in sys.argv[1:]:
        M, N = SIZES[cfg]
        ref = torch.from_numpy(cloud(M, synth.SEED_A)).cuda()
        q = cloud(N, synth.SEED_B)
        if cfg == "c4":
            q = synth.rigid_offset(q)
        q = torch.from_numpy(q).cuda()
        with capi.Index(ref, engine=capi.ENGINE_GRID, auto_sync=False) as ix:
            ix.stats()                     # clears the counters
            if cfg == "c4":
                ix.icp_align(q, max_iter=50, fixed=True)
                st = ix.stats()
                out[cfg] = {"pairs_per_call": int(st[4]), "passes": 51, "pairs_per_pass": st[4] / 51.0, "queries": N, "references": M}
            else:
                idx = torch.empty(N, dtype=torch.int32, device="cuda")
                d2 = torch.empty(N, dtype=torch.float32, device="cuda")
                ix.nn1(q, idx, d2)
                st = ix.stats()
                out[cfg] = {"pairs_per_call": int(st[4]), "pairs_per_query": st[4] / float(N), "queries": N, "references": M,
                            "open_lanes": int(st[7]), "fallback_queries": int(st[1])}
        del ref, q
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
