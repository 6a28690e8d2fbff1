#!/usr/bin/env python3
"""bench.py -- headline benchmark of libpcc_nn: exact k=1 NN queries/s on N x N clouds.

One "step" = what the reference does per cloud pair on this path (src/comparator.cpp:
564-577): build the search index over the reference cloud, then one nearest-neighbour
query per point of the query cloud.  Inputs (raw AoS clouds) are resident in HBM
when the timed region starts; outputs (idx int32, d2 float32) stay in HBM.

Single GPU:   python bench.py [--steps K --warmup W --config c2|c3|c4|c5]
Multi GPU:    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
              one rank per GPU; the reference cloud is broadcast once over RCCL/xGMI,
              every rank searches its own shard of queries (weak scaling, no collective
              inside the timed region).

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit...,
plus "roofline" (dominant kernel of the measured path, HIP-event timed on the library's
stream), "cpu_baseline" (the oracle's FLANN-restatement kd-tree on this host, reported
only) and "exhaustive" (the north-star tiled brute-force kernel on the same workload,
priced against the non-FMA fp32 VALU peak that bounds it).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch  # noqa: E402  (import before the C-ABI so one HIP runtime is shared)

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from pointcloudcomparator_amd import capi, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_NOFMA_PEAK_TOPS = 78.65   # 157.3 TFLOP/s fp32 vector peak / 2 (no FMA allowed: bit parity with FLANN)
OPS_PER_PAIR = 9               # 3 sub + 3 mul + 2 add + 1 min

CONFIGS = {
    # name: (reference points M, query points N per GPU, point stride in floats, description)
    "c1": (10_000, 10_000, 3, "C1: 10k x 10k XYZ, k=1"),
    "c2": (1_000_000, 1_000_000, 3, "C2: 1M x 1M XYZ, k=1 NN"),
    "c3": (10_000_000, 10_000_000, 8, "C3: 10M x 10M XYZRGB (32-B stride), k=1 NN"),
    "c4": (2_000_000, 2_000_000, 3, "C4: 2M x 2M XYZ, k=1 NN (one ICP correspondence pass)"),
    "c5": (8_000_000, 4_000_000, 3, "C5 shard: 4M queries per GPU vs 8M references"),
}


def make_cloud(n, seed, floats, start=0):
    pts = synth.corridor_cloud(n, seed, start=start)
    return synth.with_rgb_stride(pts) if floats == 8 else pts


def load_pmc_traffic(kernel_key, workload_key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/*.json)."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*pmc_traffic*.json")):
        try:
            d = json.loads(f.read_text())
        except Exception:
            continue
        v = d.get(workload_key, {}).get(kernel_key)
        if v is not None:
            best = v
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--engine", default="auto", choices=["auto", "grid", "brute"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-exhaustive", action="store_true", help="skip the exhaustive-kernel leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # one rank per GPU.  Rehearsals on a box with fewer GPUs than ranks share devices; RCCL refuses two
    # ranks on one device, so that case (never the driver's) falls back to gloo and says so.
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev
    backend = os.environ.get("PCC_BENCH_BACKEND", "nccl" if world <= ndev else "gloo")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
            if rank == 0:
                print(f"[bench] note: {world} ranks on {ndev} GPU(s): backend {backend}, devices shared", file=sys.stderr)
    else:
        torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    n_gpus = world
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    M, N, floats, desc = CONFIGS[args.config]
    K, W = args.steps, args.warmup

    # ---- inputs: reference cloud on rank 0, broadcast once over RCCL; queries sharded ----------
    ref_host = make_cloud(M, synth.SEED_A, floats) if rank == 0 else None
    ref = torch.empty((M, floats), dtype=torch.float32, device=dev)
    if rank == 0:
        ref.copy_(torch.from_numpy(ref_host))
    bcast_ms = 0.0
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        dist.broadcast(ref, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    qry_host = make_cloud(N, synth.SEED_B, floats, start=rank * N)  # this rank's shard
    qry = torch.from_numpy(qry_host).to(dev)
    idx = torch.empty(N, dtype=torch.int32, device=dev)
    d2 = torch.empty(N, dtype=torch.float32, device=dev)

    engine = {"auto": capi.ENGINE_AUTO, "grid": capi.ENGINE_GRID, "brute": capi.ENGINE_BRUTE}[args.engine]
    ix = capi.Index(ref, engine=engine)
    engine_name = {capi.ENGINE_GRID: "grid", capi.ENGINE_BRUTE: "brute"}[ix.engine]

    def step():
        ix.set_input(ref)       # index build over the resident reference cloud
        ix.nn1(qry, idx, d2)    # N queries; asynchronous on the library's stream

    def timed(fn, k):
        ix.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        ix.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(W):
        step()
    # timed region: only the dominant kernel is bracketed by HIP events (2 per step, recorded by
    # the library on its own stream into a ring, no sync); a full 10-event breakdown costs
    # ~48 us per 0.4 ms step, so it runs in a second, untimed pass
    ix.enable_timing(1)
    dt = timed(step, K)
    tm_main = ix.timing()     # tm_main[0] = average k_grid_nn1 / k_nn1_brute duration over the K timed steps
    ix.enable_timing(2)
    timed(step, min(K, 10))
    tm = ix.timing()
    tm[0] = tm_main[0]
    ix.enable_timing(0)
    stats = ix.stats()
    ms_per_step = dt / K * 1e3
    bcast_name = "RCCL" if backend == "nccl" else backend
    value = N * n_gpus / (dt / K)

    # query-only rate (index kept, as inside ICP where the target tree is built once)
    dtq = timed(lambda: ix.nn1(qry, idx, d2), K)
    query_only = N * n_gpus / (dtq / K)

    out = {
        "metric": "nn_queries_per_sec",
        "value": value,
        "unit": "queries/s",
        "n_gpus": n_gpus,
        "steps": K,
        "warmup": W,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{desc}; step = index build + {N} queries per GPU vs {M} references",
            "engine": engine_name,
            "references": M,
            "queries_per_gpu": N,
            "point_stride_bytes": floats * 4,
            "parallelism": f"query-sharded x{n_gpus}, reference cloud replicated ({bcast_name} broadcast {bcast_ms:.2f} ms, outside the timed region)",
        },
        "query_only_queries_per_sec": query_only,
        "build_ms": tm[3],
        "search_call_ms": tm[2],
        "fallback_queries": stats[1] if engine_name == "grid" else 0,
    }

    if rank == 0:
        res_idx = idx.cpu().numpy()
        res_d2 = d2.cpu().numpy()
        wl_key = f"{args.config}"
        # ---- roofline of the dominant kernel of the measured path ---------------------------
        if engine_name == "grid":
            # k_grid_nn1: every reference point (16 B packed) has to be read at least once,
            # every query read once (16 B packed + 4 B order) and its result written (8 B)
            alg_bytes = 16.0 * M + 28.0 * N
            ach = alg_bytes / (tm[0] * 1e-3) / 1e9 if tm[0] > 0 else 0.0
            out["roofline"] = {
                "kernel": "k_grid_nn1", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": load_pmc_traffic("k_grid_nn1", wl_key),
                "kernel_ms": tm[0], "algorithmic_bytes": alg_bytes,
                "note": "pruned exact search: ~80 % of its time is L1 line lookups and L2->L1 fills of per-lane 16-byte gathers (csrc/ubench_gather.hip prices them; DESIGN.md 4.2); compulsory HBM bytes are a few % of the roof by construction",
            }
        else:
            pairs = float(M) * N
            ach = pairs * OPS_PER_PAIR / (tm[0] * 1e-3) / 1e12 if tm[0] > 0 else 0.0
            out["roofline"] = {
                "kernel": "k_nn1_brute", "bound": "valu", "achieved": ach, "peak": VALU_NOFMA_PEAK_TOPS,
                "unit": "Top/s", "frac": ach / VALU_NOFMA_PEAK_TOPS,
                "traffic": load_pmc_traffic("k_nn1_brute", wl_key), "kernel_ms": tm[0],
            }

    # ---- exhaustive (north-star) kernel on the same workload, N=1 only -------------------------
    if n_gpus == 1 and not args.no_exhaustive and engine_name != "brute" and M * N <= 4e12:
        ix.set_engine(capi.ENGINE_BRUTE)
        idx_b = torch.empty_like(idx)
        d2_b = torch.empty_like(d2)
        ix.nn1(qry, idx_b, d2_b)
        ix.enable_timing(1)
        kb = 3
        dtb = timed(lambda: ix.nn1(qry, idx_b, d2_b), kb)
        tb = ix.timing()
        ix.enable_timing(0)
        pairs = float(M) * N
        ach = pairs * OPS_PER_PAIR / (tb[0] * 1e-3) / 1e12
        same = bool((idx_b == idx).all().item() and (d2_b.view(torch.int32) == d2.view(torch.int32)).all().item())
        out["exhaustive"] = {
            "kernel": "k_nn1_brute", "value": N / (dtb / kb), "unit": "queries/s", "ms_per_pass": dtb / kb * 1e3,
            "pairs_per_sec": pairs / (tb[0] * 1e-3),
            "roofline": {"bound": "valu", "achieved": ach, "peak": VALU_NOFMA_PEAK_TOPS, "unit": "Top/s",
                         "frac": ach / VALU_NOFMA_PEAK_TOPS, "kernel_ms": tb[0],
                         "traffic": load_pmc_traffic("k_nn1_brute", args.config),
                         "ops_per_pair": OPS_PER_PAIR},
            "bit_identical_to_grid": same,
        }
        ix.set_engine(engine)

    # ---- CPU baseline: the oracle's kd-tree restatement on this host (reported only) -----------
    if rank == 0 and n_gpus == 1 and not args.no_cpu:
        import oracle
        ncores = len(os.sched_getaffinity(0))
        sample = min(N, 1_000_000)
        t0 = time.perf_counter()
        kd = oracle.KdTree(ref_host)
        tb_cpu = time.perf_counter() - t0
        t0 = time.perf_counter()
        ci, cd = kd.nn1_batch(qry_host[:sample])
        tq_cpu = time.perf_counter() - t0
        # whole-job rate of the same step (build + all N queries), queries extrapolated from the sample
        cpu_rate = N / (tb_cpu + tq_cpu * (N / sample))
        t0 = time.perf_counter()
        kd.nn1_batch(qry_host[:sample], nthreads=ncores)
        tq_mt = time.perf_counter() - t0
        d2_equal = bool((cd.view(np.uint32) == res_d2[:sample].view(np.uint32)).all())
        idx_diff = int((ci != res_idx[:sample]).sum())
        out["cpu_baseline"] = {
            "value": cpu_rate, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": f"kd-tree build over all {M} references ({tb_cpu:.3f} s) + first {sample} of {N} queries "
                      f"({tq_cpu:.3f} s), one thread, oracle/pcc_oracle.c (FLANN KDTreeSingleIndex restatement)",
            "all_cores": {"cores": ncores, "query_only_queries_per_sec": sample / tq_mt},
            "query_only_queries_per_sec": sample / tq_cpu,
            "gpu_matches_cpu": {"d2_bits_equal": d2_equal, "index_mismatches": idx_diff},
        }

    ix.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
