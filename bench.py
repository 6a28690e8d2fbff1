#!/usr/bin/env python3
"""bench.py -- headline benchmark of libpcc_nn: exact k=1 NN queries/s on N x N clouds.

One "step" = what the reference does per cloud pair on this path (src/comparator.cpp:
564-577): build the search index over the reference cloud, then one nearest-neighbour
query per point of the query cloud.  Inputs (raw AoS clouds) are resident in HBM
when the timed region starts; outputs (idx int32, d2 float32) stay in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]

Workload (BASELINE.json configs):
  default       C3, 10M x 10M XYZRGB (north_star's target size).  N = 1: the whole configuration on one GPU.  N > 1: the SAME
                10M queries sharded over the ranks against the replicated 10M references -- north_star's partition, STRONG
                scaling: `value` = 10M / (slowest rank's step).  The index build (0.4 ms at 10M) is replicated on every rank,
                so the whole step cannot scale linearly (Amdahl); `scaling_terms` carries step / query-only / build times so
                the curve can be read both ways, `extra.c3_weak` is the weak form (every rank its own 10M queries).
                `extra` adds, at N = 1: C2 (1M x 1M, with the exhaustive north-star kernel on the same data), the C3
                radius-0.05 clustering leg (-e path, 5M object points), C4 (50 fixed ICP iterations, 2M x 2M), the C5 shard
                and `scaling_projection` (one GPU doing a G-th of the queries, G = 1, 2, 4, 8: a PROJECTION of the
                strong-scaling curve from one device, never a measurement of it); at every N: C5 (32M queries sharded over
                the ranks vs 8M references, BASELINE configs[4]).
  --config cX   only that configuration as the measured workload, no extras (profiling); c5_shard = what one of eight GPUs
                does in C5 (4M queries vs 8M references).
Multi GPU: one rank per GPU over RCCL.  `--gpus N` without a torch.distributed environment launches
  the N ranks itself (python -m torch.distributed.run ... as a CHILD process, before this process
  touches the GPU); under the driver's own launcher the ranks are used as they come.  The reference
  cloud is broadcast once (outside the timed region; 16 B per point -- the packed form, SURVEY.md 8e -- whatever the
  workload's stride), queries are sharded, no collective inside a step.

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit..., "roofline"
(dominant kernel of the measured path, HIP-event timed on the library's stream) and "cpu_baseline"
(the oracle's FLANN-restatement kd-tree on this host, reported only).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_NOFMA_PEAK_TOPS = 78.65   # 157.3 TFLOP/s fp32 vector peak / 2 (no FMA allowed: bit parity with FLANN)
OPS_PER_PAIR = 9               # 3 sub + 3 mul + 2 add + 1 min

CONFIGS = {
    # name: (reference points M, query points N per GPU, point stride in floats, description)
    "c1": (10_000, 10_000, 3, "C1: 10k x 10k XYZ, k=1"),
    "c2": (1_000_000, 1_000_000, 3, "C2: 1M x 1M XYZ, k=1 NN"),
    "c3": (10_000_000, 10_000_000, 8, "C3: 10M x 10M XYZRGB (32-B stride), k=1 NN"),
    "c4": (2_000_000, 2_000_000, 3, "C4: 2M x 2M XYZ, -i ICP, 50 fixed iterations"),
    "c5": (8_000_000, 4_000_000, 3, "C5: 32M queries in total, sharded over the ranks present, vs 8M references"),
    "c5_shard": (8_000_000, 4_000_000, 3, "C5 shard: what ONE of eight GPUs does in C5 -- 4M queries vs 8M references"),
}
C5_TOTAL_QUERIES = 32_000_000
C3_TOTAL_QUERIES = 10_000_000
PROJECTION_G = (1, 2, 4, 8)
# sources whose kernels the committed PMC passes describe; a profile taken from other sources is flagged stale
PROFILED_SOURCES = ["grid.hip", "grid_device.hpp", "cellsort.hip", "cellsort_mp.hip", "nn1_brute.hip", "pack.hip"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="auto", choices=["auto"] + sorted(CONFIGS))
    ap.add_argument("--engine", default="auto", choices=["auto", "grid", "brute"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-exhaustive", action="store_true", help="skip the exhaustive-kernel leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra configurations of the default run")
    ap.add_argument("--no-pairs", action="store_true", help="skip the pair count by the profiling build (roofline then carries the HBM figure only)")
    return ap.parse_args()


def check_device_count(ranks: int) -> None:
    """one rank per GPU: a node with fewer devices than ranks is refused before anything is launched or any
    process group is formed (a hang at the first collective otherwise).  torch.cuda.device_count() does not
    initialise the GPU on this image.  PCC_BENCH_SHARE_DEVICES=1 keeps the one-GPU rehearsal (ranks share
    devices, gloo instead of RCCL) possible; it is never the driver's case."""
    import torch
    ndev = torch.cuda.device_count()
    if ranks > ndev and os.environ.get("PCC_BENCH_SHARE_DEVICES", "0") != "1":
        sys.exit(f"bench.py: {ranks} ranks requested but this node has {ndev} GPU(s); one rank per GPU is required "
                 "(set PCC_BENCH_SHARE_DEVICES=1 for a shared-device rehearsal over gloo)")


def launch_ranks(args) -> int:
    """`bench.py --gpus N` outside a torch.distributed environment: start the N ranks as a child process group.
    Nothing here has touched the GPU (no torch.cuda call, libpcc_nn not loaded): the child is a fresh program."""
    check_device_count(args.gpus)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def under_profiler() -> bool:
    """rocprofv3 preloads its tool library into this process (and would into a child): no child process then -- on this pool
    a program started from a GPU-initialised process is refused, and a profiled run does not need the pair count"""
    if any(k.startswith(("ROCPROFILER", "ROCPROF_", "ROCP_")) for k in os.environ):
        return True
    return "rocprof" in os.environ.get("LD_PRELOAD", "").lower()


def count_pairs(cfgs):
    """distances evaluated per call, counted by the PROFILING build of the library (libpcc_nn_prof.so, pcc_index_stats[4])
    in a child process -- the timed library carries no counter.  {} when the profiling build is missing or fails."""
    prof = ROOT / "pointcloudcomparator_amd" / "lib" / "libpcc_nn_prof.so"
    if not prof.exists() or not cfgs or under_profiler():
        return {}
    env = dict(os.environ)
    env["PCC_LIB"] = str(prof)
    try:
        p = subprocess.run([sys.executable, str(ROOT / "tools" / "count_pairs.py")] + list(cfgs), env=env, capture_output=True,
                           text=True, timeout=300)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        return json.loads(line[-1]) if p.returncode == 0 and line else {}
    except Exception:
        return {}


def load_nn1_counters(label):
    """derived PMC metrics of the k = 1 kernels from the committed profiles/*_nn1_counters.json (VALU busy share, lanes per
    instruction ...): what the counters say bounds the kernel"""
    for f in sorted((ROOT / "profiles").glob("*nn1_counters.json"), reverse=True):
        try:
            d = json.loads(f.read_text()).get(label, {}).get("derived")
        except Exception:
            d = None
        if d:
            return {"source": "profiles/" + f.name, "valu_busy_fraction": next((v for k, v in d.items() if k.startswith("valu_busy")), None),
                    "active_lanes_per_valu_instruction": d.get("active_lanes_per_valu_instruction"),
                    "valu_instructions_per_wave": d.get("valu_instructions_per_wave"),
                    "wave_time_waiting_fraction": next((v for k, v in d.items() if k.startswith("wave_time_waiting")), None)}
    return None


def source_digest() -> str:
    h = hashlib.sha256()
    for name in PROFILED_SOURCES:
        p = ROOT / "pointcloudcomparator_amd" / "csrc" / name
        h.update(p.read_bytes() if p.exists() else b"")
    return h.hexdigest()[:16]


def pcl_present() -> bool:
    """is a PCL kd-tree library installed on this box (pkg-config / ldconfig / the usual include directories)?  If it ever is,
    a C++ harness against the real pcl::KdTreeFLANN is what can pin parity; until then cpu_baseline.kind = port."""
    try:
        for cmd in (["pkg-config", "--exists", "pcl_kdtree"], ["pkg-config", "--exists", "pcl_kdtree-1.7"]):
            if subprocess.run(cmd, capture_output=True, timeout=10).returncode == 0:
                return True
    except Exception:
        pass
    try:
        if "pcl_kdtree" in subprocess.run(["ldconfig", "-p"], capture_output=True, text=True, timeout=10).stdout:
            return True
    except Exception:
        pass
    return any(any(Path(d).glob("pcl*/pcl/kdtree/kdtree_flann.h")) for d in ("/usr/include", "/usr/local/include", "/opt/include"))


def load_pmc_traffic(kernel_key, workload_key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/*pmc_traffic*.json), and whether
    that profile was taken from the kernel sources in this tree."""
    best, stale = None, None
    for f in sorted((ROOT / "profiles").glob("*pmc_traffic*.json")):
        try:
            d = json.loads(f.read_text())
        except Exception:
            continue
        v = d.get(workload_key, {}).get(kernel_key) if workload_key else None
        if v is not None:
            best = v
            stale = d.get("source_digest") != source_digest()
    return best, stale


def compact(x):
    """floats to six significant digits: the line stays within what a log tail keeps"""
    if isinstance(x, float):
        return float(f"{x:.6g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: compact(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [compact(v) for v in x]
    return x


def short_line(out):
    """the one JSON line: contract keys + roofline / cpu_baseline as flat scalars + every extra leg as {ms, value, ...}"""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "query_only_queries_per_sec", "build_ms", "search_call_ms", "query_sort_ms", "fallback_queries",
            "broadcast_ms", "broadcast_bytes", "per_rank_ms_per_step", "icp")
    line = {k: out[k] for k in keep if k in out}
    if "scaling_terms" in out:
        line["scaling_terms"] = {k: v for k, v in out["scaling_terms"].items() if k != "note"}
    rf = out.get("roofline")
    if rf:
        hbm = rf.get("hbm") or {}
        flat = {"kernel": rf.get("kernel"), "bound": "hbm", "achieved": hbm.get("achieved", rf.get("achieved")), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": hbm.get("frac", rf.get("frac")), "traffic": rf.get("traffic"), "traffic_stale": rf.get("traffic_stale"),
                "kernel_ms": rf.get("kernel_ms"), "algorithmic_bytes": hbm.get("algorithmic_bytes")}
        if rf.get("bound") == "valu" and "hbm" in rf:   # pruned search: the pair arithmetic against the non-FMA fp32 roof beside it
            flat.update({"pairs_per_launch": rf.get("pairs_per_launch"), "valu_achieved_tops": rf.get("achieved"), "valu_frac": rf.get("frac")})
        elif rf.get("bound") == "valu":                 # exhaustive engine measured directly: VALU-bound by SURVEY 8d
            flat.update({"bound": "valu", "achieved": rf.get("achieved"), "peak": VALU_NOFMA_PEAK_TOPS, "unit": "Top/s", "frac": rf.get("frac")})
        ex = rf.get("exhaustive") or (out.get("exhaustive") or {}).get("roofline")
        if ex:  # the kernel north_star names (k_nn1_brute), on C2's data or on the measured config
            flat.update({"exhaustive_kernel": "k_nn1_brute", "exhaustive_kernel_ms": ex.get("kernel_ms"), "exhaustive_achieved_tops": ex.get("achieved"),
                         "exhaustive_frac": ex.get("frac"), "exhaustive_bit_identical": ex.get("bit_identical_to_grid", (out.get("exhaustive") or {}).get("bit_identical_to_grid"))})
        cnt = rf.get("counters") or {}
        for k in ("valu_busy_fraction", "active_lanes_per_valu_instruction", "valu_instructions_per_wave"):
            if cnt.get(k) is not None:
                flat["pmc_" + k] = cnt[k]
        line["roofline"] = flat
    if "cpu_baseline" in out:
        line["cpu_baseline"] = out["cpu_baseline"]
    ex = out.get("extra")
    if ex:
        e = {}
        for name in ("c2", "c5_shard", "c5", "c3_weak"):
            if name in ex:
                r = ex[name]
                e[name] = {"ms": r["ms_per_step"], "value": r["value"], "build_ms": r["build_ms"], "kernel_ms": r["roofline"].get("kernel_ms"),
                           "frac": (r["roofline"].get("hbm") or r["roofline"]).get("frac")}
        if "c3_clusters" in ex:
            r = ex["c3_clusters"]
            e["c3_clusters"] = {"ms": r["ms"], "value": r["points_per_sec"], "clusters": r["clusters"], "ok": r["all_points_clustered"] and r["clusters"] == r["clusters_expected"]}
        if "c4_icp" in ex:
            r = ex["c4_icp"]
            e["c4_icp"] = {"ms": r["ms"], "value": r["nn_queries_per_sec"], "ms_per_pass": r["ms_per_pass"], "nn_kernel_ms_per_pass": r["split_ms_per_pass"]["nn_kernel"],
                           "frac": ((r.get("roofline") or {}).get("hbm") or {}).get("frac")}
        if "room" in ex:
            sc, big = ex["room"]["scan"], ex["room"]["10M"]
            e["room_scan"] = {"points": sc["points"], "ms": sc["nn1_step_ms"], "value": sc["nn1_queries_per_sec"], "sor_k51_ms": sc.get("sor_k51_ms"),
                              "clusters_ms": sc.get("clusters_ms"), "clusters_ok": sc.get("clusters") == sc.get("clusters_expected")}
            e["room_10M"] = {"ms": big["nn1_step_ms"], "value": big["nn1_queries_per_sec"], "kernel_ms": big["nn1_kernel_ms"]}
        sp = ex.get("scaling_projection")
        if sp:
            e["scaling_projection"] = {"kind": "PROJECTION from one GPU (this GPU doing one rank's share), not a multi-GPU measurement", "gpus": list(PROJECTION_G)}
            for name in ("c3", "c5"):
                if sp.get(name):
                    e["scaling_projection"][name] = {"step": [r["projected_speedup_step"] for r in sp[name]["rows"]],
                                                     "query_only": [r["projected_speedup_query_only"] for r in sp[name]["rows"]],
                                                     "query_only_ms": [r["query_only_ms"] for r in sp[name]["rows"]]}
        for name in ("small_calls", "host_path", "c_abi_comm"):
            if name in ex:
                e[name] = ex[name]
        line["extra"] = e
    line["note"] = ("pruned exact k = 1 search (k_grid_nn1_flat2 + k_nn1_open_flat): roofline = SURVEY 8d bytes (12 B per reference and query, 8 B "
                    "per result) over the kernels' HIP-event time; `traffic` = FETCH_SIZE x 2 + WRITE_SIZE per call from the committed PMC pass; "
                    "exhaustive_* = the tiled brute-force kernel north_star names, on C2, against the non-FMA fp32 roof.  Full record: "
                    "gpurun_out/bench_full.json")
    return line


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    # Pair counts of the workloads this run measures: the PROFILING build of the library in a CHILD process -- started here,
    # before this process initialises the GPU or joins a process group (a child forked later would inherit a profiler's
    # preload and keep the other ranks waiting in their first collective).  One-GPU runs only: the counts price the
    # single-GPU rooflines; under rocprofv3 the child is skipped altogether.
    primary_cfg = "c3" if args.config == "auto" else args.config
    pairs = {}
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_pairs:
        want = [primary_cfg] if primary_cfg != "c1" else []
        if args.config == "auto" and not args.no_extra:
            want += ["c2", "c4", "c5_shard", "c5"]
        pairs.update(count_pairs([c for c in dict.fromkeys(want) if c in ("c2", "c3", "c4", "c5", "c5_shard")]))

    import numpy as np
    import torch  # (import before the C-ABI so one HIP runtime is shared)
    sys.path.insert(0, str(ROOT))
    from pointcloudcomparator_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # one rank per GPU.  Rehearsals on a box with fewer GPUs than ranks share devices; RCCL refuses two
    # ranks on one device, so that case (never the driver's) falls back to gloo and says so.
    check_device_count(world)
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev
    backend = os.environ.get("PCC_BENCH_BACKEND", "nccl" if world <= ndev else "gloo")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # PCC_BENCH_FORCE_GROUP=1 forms the process group at world size 1 too: every collective of the N > 1 path
    # (RCCL broadcast, all_reduce(MAX), per-rank gather, destroy) then runs on a one-GPU box (tests/test_bench_gpu.py)
    if world > 1 or os.environ.get("PCC_BENCH_FORCE_GROUP", "0") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
            if rank == 0:
                print(f"[bench] note: {world} ranks on {ndev} GPU(s): backend {backend}, devices shared", file=sys.stderr)
    n_gpus = dist.get_world_size() if dist is not None else 1  # the ranks the process group actually holds
    backend_name = dist.get_backend() if dist is not None else "none"
    if args.gpus != n_gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but the process group has {n_gpus} rank(s); using {n_gpus}", file=sys.stderr)
    bcast_name = "RCCL" if backend == "nccl" else backend
    engine = {"auto": capi.ENGINE_AUTO, "grid": capi.ENGINE_GRID, "brute": capi.ENGINE_BRUTE}[args.engine]
    K, W = args.steps, args.warmup
    from pointcloudcomparator_amd.sharding import shard_range

    def make_cloud(n, seed, floats, start=0, chunk=4_000_000):
        parts = [synth.corridor_cloud(min(chunk, n - o), seed, start=start + o) for o in range(0, n, chunk)]
        pts = parts[0] if len(parts) == 1 else np.concatenate(parts)
        return synth.with_rgb_stride(pts) if floats == 8 else pts

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(ix, fn, k):
        """K calls bracketed by barrier + synchronize on both sides; seconds, max over ranks"""
        ix.sync()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        ix.sync()
        torch.cuda.synchronize()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    def broadcast_refs(M, floats):
        """the reference cloud: generated on rank 0, ONE broadcast to every rank (RCCL over xGMI) of 16 B per point -- the
        packed (x, y, z, pad) form SURVEY.md 8e and the library's own pcc_index_create_broadcast ship --, whatever the
        workload's stride; each rank lays the points back into that stride (the colour words of XYZRGB are not on the
        path), so the timed step reads the same 32-byte records on every rank"""
        ref_host = make_cloud(M, synth.SEED_A, floats) if rank == 0 else None
        ref = torch.zeros((M, floats), dtype=torch.float32, device=dev)
        if rank == 0:
            ref.copy_(torch.from_numpy(ref_host))
        ms, nbytes = 0.0, 0
        if dist is not None:
            wire = ref if floats <= 4 else ref[:, :4].contiguous()
            if floats > 4:
                wire[:, 3] = 0.0
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            dist.broadcast(wire, src=0)
            torch.cuda.synchronize()
            ms = max_over_ranks((time.perf_counter() - t0) * 1e3)
            nbytes = wire.numel() * 4
            if floats > 4 and rank != 0:
                ref[:, :3] = wire[:, :3]
            del wire
        return ref, ref_host, ms, nbytes

    # ---- k = 1 NN step of one configuration -----------------------------------------------------------------
    def run_nn(cfg, n_per_rank=None, total_queries=None, steps=K, warmup=W, with_exhaustive=False, with_cpu=False, project=False):
        """total_queries: that many queries sharded over the ranks (strong scaling); else n_per_rank (or the config's size)
        on every rank (weak).  project (one GPU only): the same step with a G-th of the queries, G in PROJECTION_G."""
        M, N, floats, desc = CONFIGS[cfg]
        if total_queries is not None:
            q_start, N = shard_range(total_queries, rank, n_gpus)
            q_total = total_queries
        else:
            N = n_per_rank if n_per_rank is not None else N
            q_start, q_total = rank * N, N * n_gpus
        ref, ref_host, bcast_ms, bcast_bytes = broadcast_refs(M, floats)
        qry_host = make_cloud(N, synth.SEED_B, floats, start=q_start)  # this rank's shard of cloud B
        qry = torch.from_numpy(qry_host).to(dev)
        idx = torch.empty(N, dtype=torch.int32, device=dev)
        d2 = torch.empty(N, dtype=torch.float32, device=dev)
        # auto_sync off: the timed region brackets its calls with ix.sync() itself (no per-call event records)
        ix = capi.Index(ref, engine=engine, auto_sync=False)
        engine_name = {capi.ENGINE_GRID: "grid", capi.ENGINE_BRUTE: "brute"}[ix.engine]

        def step():
            ix.set_input(ref)       # index build over the resident reference cloud
            ix.nn1(qry, idx, d2)    # N queries; asynchronous on the library's stream

        for _ in range(warmup):
            step()
        # timed region: only the dominant kernel is bracketed by HIP events (2 per step, recorded by the library
        # on its own stream into a ring, no sync); the full breakdown costs ~48 us per step and runs untimed after
        ix.enable_timing(1)
        dt = timed(ix, step, steps)
        tm_main = ix.timing()
        t0 = time.perf_counter()
        step()
        ix.sync()
        rank_ms = (time.perf_counter() - t0) * 1e3   # this rank alone, no barrier
        ix.enable_timing(2)
        timed(ix, step, min(steps, 10))
        tm = ix.timing()
        tm[0] = tm_main[0]
        ix.enable_timing(0)
        stats = ix.stats()
        # (one untimed call each first: a search that does not follow a build stages its queries on the main stream, in scratch
        # buffers a step never grew there -- the first call allocates, 4 ms at 32M queries)
        ix.nn1(qry, idx, d2)
        dtq = timed(ix, lambda: ix.nn1(qry, idx, d2), steps)  # index kept (ICP builds the target tree once)
        ix.set_input(ref)
        dtb = timed(ix, lambda: ix.set_input(ref), steps)     # the build alone (inside a step the queries' staging runs beside it)
        ix.nn1(qry, idx, d2)
        per_rank_ms = None
        if dist is not None:
            t = torch.zeros(n_gpus, dtype=torch.float64, device=dev)
            t[rank] = rank_ms
            dist.all_reduce(t)
            per_rank_ms = [round(float(v), 4) for v in t.tolist()]
        r = {
            "workload": f"{desc}; step = index build over {M} references + {N} queries on each GPU "
                        f"({q_total} over {n_gpus} GPU{'s' if n_gpus > 1 else ''})",
            "value": q_total / (dt / steps), "unit": "queries/s", "ms_per_step": dt / steps * 1e3,
            "steps": steps, "engine": engine_name, "references": M, "queries_per_gpu": N,
            "queries_total": q_total, "point_stride_bytes": floats * 4,
            "query_only_queries_per_sec": q_total / (dtq / steps),
            "build_ms": dtb / steps * 1e3, "search_call_ms": tm[2], "query_sort_ms": tm[4], "main_kernel_ms": tm[0],
            "fallback_queries": stats[1] if engine_name == "grid" else 0,
            "broadcast_ms": bcast_ms, "broadcast_bytes": bcast_bytes,
            # the terms of the scaling curve: the whole step (max over ranks), the query half alone (index kept: ICP's case) and
            # the index build, which every rank repeats whatever its share of the queries -- the Amdahl term of query sharding
            "scaling_terms": {"step_ms": dt / steps * 1e3, "query_only_ms": dtq / steps * 1e3, "build_ms": dtb / steps * 1e3,
                              "build_ms_inside_step": tm[3], "queries_per_gpu": N,
                              "note": "build_ms (the build alone, its own loop) is replicated on every rank; only query_only_ms shrinks with "
                                      "the shard.  The two parts are looped separately and do not add up to step_ms exactly (a step also "
                                      "carries the hand-over between the two calls).  From 2M queries per GPU on the queries are packed and "
                                      "sorted on a second stream beside the build's sort (PCC_OPT_OVERLAP_PREP: 2 % of the step in a same-box "
                                      "A/B, DESIGN.md 4.3), so the build's events inside a step (build_ms_inside_step) span the staging's "
                                      "kernels too and are no measure of the build"},
        }
        if per_rank_ms is not None:
            r["per_rank_ms_per_step"] = per_rank_ms
        if project and n_gpus == 1:
            # ONE GPU doing what each of G GPUs would do: the same index build + the first N / G queries.  With no collective
            # inside a step the slowest rank's time IS the job's time, so step(1) / step(G) projects the strong-scaling
            # speed-up -- a projection from one device (no xGMI, no launch jitter across ranks), never a measurement.
            rows = []
            for G in PROJECTION_G:
                n = N // G
                if G == 1:
                    sm, qm = dt / steps * 1e3, dtq / steps * 1e3
                else:
                    qv, iv, dv = qry[:n], idx[:n], d2[:n]

                    def step_g():
                        ix.set_input(ref)
                        ix.nn1(qv, iv, dv)

                    for _ in range(2):
                        step_g()
                    kk = min(steps, 10)
                    sm = timed(ix, step_g, kk) / kk * 1e3
                    ix.nn1(qv, iv, dv)
                    qm = timed(ix, lambda: ix.nn1(qv, iv, dv), kk) / kk * 1e3
                rows.append({"gpus": G, "queries_per_gpu": n, "step_ms": sm, "query_only_ms": qm})
            for row in rows:
                row["projected_speedup_step"] = rows[0]["step_ms"] / row["step_ms"]
                row["projected_speedup_query_only"] = rows[0]["query_only_ms"] / row["query_only_ms"]
            r["projection"] = {"kind": "PROJECTION from one GPU (each row: this GPU doing one rank's share); not a multi-GPU measurement",
                               "build_ms_replicated": dtb / steps * 1e3, "rows": rows}
            ix.set_input(ref)
            ix.nn1(qry, idx, d2)  # (the full result again, for the checks below)
        # ---- roofline of the dominant kernel of the measured path -----------------------------------------
        if engine_name == "grid":
            # the pruned search (k_grid_nn1_flat2, plus k_nn1_open_flat for the lanes its cube leaves open from 2M queries on;
            # both inside the event bracket).  SURVEY.md 8d, k = 1 pruned: every reference and every query read once at 12 B,
            # 8 B of result per query.  What binds the kernel is not that stream but the instructions it issues; the pair
            # arithmetic among them is priced against the non-FMA fp32 roof from the profiling build's pair count
            alg = 12.0 * (M + N) + 8.0 * N
            ach = alg / (tm[0] * 1e-3) / 1e9 if tm[0] > 0 else 0.0
            # (profiles and pair counts are keyed by WORKLOAD, not by config name: the 32M-query C5 and its 4M shard differ)
            wl = cfg
            if cfg in ("c5", "c5_shard"):
                wl = "c5" if N == C5_TOTAL_QUERIES else ("c5_shard" if N == C5_TOTAL_QUERIES // 8 else None)
            elif N != CONFIGS[cfg][1]:
                wl = None
            t_main, stale = load_pmc_traffic("k_grid_nn1_flat2", wl)
            t_open, _ = load_pmc_traffic("k_nn1_open_flat", wl)
            if t_open is None:
                t_open, _ = load_pmc_traffic("k_nn1_open", wl)
            traffic = None if t_main is None else t_main + (t_open or 0.0)
            pk = pairs.get(wl)
            hbm = {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes": alg,
                   "bytes_per_unit": "12 B per reference + 12 B per query read, 8 B per query written (SURVEY.md 8d)"}
            r["roofline"] = {"kernel": "k_grid_nn1_flat2 (+ k_nn1_open_flat)", "kernel_ms": tm[0], "traffic": traffic, "traffic_stale": stale,
                             "hbm": hbm, "counters": load_nn1_counters("c3_flat2" if M >= 5_000_000 else "c2_flat2")}
            if pk and tm[0] > 0 and pk.get("queries") == N and pk.get("references") == M:
                pps = pk["pairs_per_call"] / (tm[0] * 1e-3)
                top = pps * OPS_PER_PAIR / 1e12
                r["roofline"].update({"bound": "valu", "achieved": top, "peak": VALU_NOFMA_PEAK_TOPS, "unit": "Top/s",
                                      "frac": top / VALU_NOFMA_PEAK_TOPS, "pairs_per_launch": pk["pairs_per_call"],
                                      "pairs_per_query": pk["pairs_per_call"] / float(N), "pairs_per_s": pps, "ops_per_pair": OPS_PER_PAIR,
                                      "valu_fraction": top / VALU_NOFMA_PEAK_TOPS, "hbm_fraction": hbm["frac"]})
                r["roofline"]["note"] = ("pruned exact search: VALU-issue / latency bound (see counters: the VALUs are busy most of the "
                                         "time, the pair arithmetic -- 9 operations x pairs_per_launch, counted by libpcc_nn_prof.so -- is "
                                         "a small share of what they issue; the rest is per-(query, row) set-up, DESIGN.md 4.2).  "
                                         "The HBM figure on SURVEY 8d's bytes is beside it under `hbm`")
            else:  # no profiling build at hand: the schema's HBM form
                r["roofline"].update({"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                      "note": "pair count unavailable (libpcc_nn_prof.so missing): HBM figure only"})
        else:
            ach = float(M) * N * OPS_PER_PAIR / (tm[0] * 1e-3) / 1e12 if tm[0] > 0 else 0.0
            traffic, stale = load_pmc_traffic("k_nn1_brute", cfg if N == CONFIGS[cfg][1] else None)
            r["roofline"] = {"kernel": "k_nn1_brute", "bound": "valu", "achieved": ach, "peak": VALU_NOFMA_PEAK_TOPS,
                             "unit": "Top/s", "frac": ach / VALU_NOFMA_PEAK_TOPS, "traffic": traffic,
                             "traffic_stale": stale, "kernel_ms": tm[0]}
        res_idx = res_d2 = None
        if rank == 0 and (with_cpu or with_exhaustive):
            res_idx, res_d2 = idx.cpu().numpy(), d2.cpu().numpy()
        # ---- exhaustive (north-star) kernel on the same workload ---------------------------------------------
        if with_exhaustive and n_gpus == 1 and engine_name != "brute" and float(M) * N <= 4e12:
            ix.set_engine(capi.ENGINE_BRUTE)
            idx_b, d2_b = torch.empty_like(idx), torch.empty_like(d2)
            ix.nn1(qry, idx_b, d2_b)
            ix.enable_timing(1)
            kb = 3
            dtb = timed(ix, lambda: ix.nn1(qry, idx_b, d2_b), kb)
            tb = ix.timing()
            ix.enable_timing(0)
            n_pairs = float(M) * N
            ach = n_pairs * OPS_PER_PAIR / (tb[0] * 1e-3) / 1e12
            same = bool((idx_b == idx).all().item() and (d2_b.view(torch.int32) == d2.view(torch.int32)).all().item())
            traffic, stale = load_pmc_traffic("k_nn1_brute", cfg)
            r["exhaustive"] = {
                "kernel": "k_nn1_brute", "value": N / (dtb / kb), "unit": "queries/s", "ms_per_pass": dtb / kb * 1e3,
                "pairs_per_sec": n_pairs / (tb[0] * 1e-3),
                "roofline": {"bound": "valu", "achieved": ach, "peak": VALU_NOFMA_PEAK_TOPS, "unit": "Top/s",
                             "frac": ach / VALU_NOFMA_PEAK_TOPS, "kernel_ms": tb[0], "traffic": traffic,
                             "traffic_stale": stale, "ops_per_pair": OPS_PER_PAIR},
                "bit_identical_to_grid": same}
            ix.set_engine(engine)
        # ---- CPU baseline: the oracle's kd-tree restatement on this host (reported only) ---------------------
        if with_cpu and rank == 0 and n_gpus == 1:
            import oracle
            ncores = len(os.sched_getaffinity(0))
            sample = min(N, 1_000_000)
            t0 = time.perf_counter()
            kd = oracle.KdTree(ref_host)
            tb_cpu = time.perf_counter() - t0
            t0 = time.perf_counter()
            ci, cd = kd.nn1_batch(qry_host[:sample])
            tq_cpu = time.perf_counter() - t0
            cpu_rate = N / (tb_cpu + tq_cpu * (N / sample))  # the same step, queries extrapolated from the sample
            t0 = time.perf_counter()
            kd.nn1_batch(qry_host[:sample], nthreads=ncores)
            tq_mt = time.perf_counter() - t0
            r["cpu_baseline"] = {
                "value": cpu_rate, "unit": "queries/s", "cores": 1, "kind": "port",
                "sample": f"kd-tree build over all {M} references ({tb_cpu:.3f} s) + first {sample} of {N} queries "
                          f"({tq_cpu:.3f} s), one thread, oracle/pcc_oracle.c (FLANN KDTreeSingleIndex restatement)",
                "query_only_queries_per_sec": sample / tq_cpu,
                "all_cores": ncores, "all_cores_query_only_qps": sample / tq_mt,
                "d2_bits_equal": bool((cd.view(np.uint32) == res_d2[:sample].view(np.uint32)).all()),
                "index_mismatches": int((ci != res_idx[:sample]).sum()),
                # real PCL on the GPU box would make this `reference`; probed, never assumed
                "pcl_found": pcl_present()}
            del kd
        ix.close()
        del ref, qry, idx, d2
        torch.cuda.empty_cache()
        return r

    # ---- C3's second half: radius-0.05 Euclidean clustering of the object layer (-e path) -------------------
    def run_clusters(n=5_000_000, reps=3):
        obj = torch.from_numpy(synth.corridor_cloud(n, synth.SEED_A, layer="objects")).to(dev)
        ix = capi.Index(obj, auto_sync=False)
        labels = torch.empty(n, dtype=torch.int32, device=dev)
        out = ix.euclidean_clusters(0.05, 100, 250000, device_out=labels)  # warm-up + result
        ix.sync()
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix.set_input(obj)  # the tree EuclideanClusterExtraction is handed (src/segmentation.cpp:120-122)
            out = ix.euclidean_clusters(0.05, 100, 250000, device_out=labels)
            ix.sync()
            best = min(best, time.perf_counter() - t0)
        lab, ncl, sizes = out
        ix.close()
        return {"workload": f"C3 -e leg: index build + EuclideanClusterExtraction(0.05, 100, 250000) over the {n} object-layer "
                            "points of cloud A (known partition: the 256 balls)",
                "ms": best * 1e3, "points_per_sec": n / best, "clusters": int(ncl),
                "clusters_expected": synth.N_BALLS, "points_clustered": int(np.asarray(sizes).sum()),
                "all_points_clustered": bool(int(np.asarray(sizes).sum()) == n)}

    # ---- a surface-sampled scene next to the volumetric corridor: two scans of a room ------------------------------
    def run_room(n, reps=3, full=True):
        """points on 2-D surfaces (synth.room_cloud; sizes from the reference's own result file): the k = 1 step, the -n
        noise pass (SOR, 51-NN) and the -e clustering of the furniture.  Reported beside C3, never instead of it."""
        a_host, b_host = synth.room_cloud(n, synth.SEED_A), synth.room_cloud(n, synth.SEED_B)
        a, b = torch.from_numpy(a_host).to(dev), torch.from_numpy(b_host).to(dev)
        idx = torch.empty(n, dtype=torch.int32, device=dev)
        d2 = torch.empty(n, dtype=torch.float32, device=dev)
        ix = capi.Index(a, auto_sync=False)

        def step():
            ix.set_input(a)
            ix.nn1(b, idx, d2)

        for _ in range(3):
            step()
        dt = timed(ix, step, 10) / 10
        ix.enable_timing(2)
        timed(ix, step, 5)
        tm = ix.timing()
        ix.enable_timing(0)
        r = {"workload": f"room scan, {n} x {n} points on the surfaces of a 6 x 5 x 2.5 m room with furniture (synth.room_cloud)",
             "points": n, "nn1_step_ms": dt * 1e3, "nn1_queries_per_sec": n / dt, "build_ms": tm[3], "query_sort_ms": tm[4],
             "nn1_kernel_ms": tm[0], "cells": int(ix.stats()[3]), "fallback_queries": int(ix.stats()[1])}
        if full:
            best = 1e9
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                md, inl, thr, kept = ix.sor(50, 1.5, device=dev)
                best = min(best, time.perf_counter() - t0)
            r["sor_k51_ms"] = best * 1e3
            r["sor_kept"] = int(kept)
            m = max(n // 4, 100_000)
            f = torch.from_numpy(synth.room_cloud(m, synth.SEED_A, part="furniture")).to(dev)
            labels = torch.empty(m, dtype=torch.int32, device=dev)
            fx = capi.Index(f, auto_sync=False)
            best = 1e9
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fx.set_input(f)
                lab, ncl, sizes = fx.euclidean_clusters(0.05, 100, 2_000_000_000, device_out=labels)
                fx.sync()
                best = min(best, time.perf_counter() - t0)
            r["clusters_ms"] = best * 1e3
            r["clusters_points"] = m
            r["clusters"] = int(ncl)
            r["clusters_expected"] = len(synth._ROOM_BOXES)
            fx.close()
        ix.close()
        del a, b, idx, d2
        torch.cuda.empty_cache()
        return r

    # ---- C4: -i ICP, 50 fixed iterations, 2M x 2M -------------------------------------------------------------
    def run_icp(reps=2, iters=50):
        M, N, floats, desc = CONFIGS["c4"]
        tgt = torch.from_numpy(make_cloud(M, synth.SEED_A, floats)).to(dev)
        src = torch.from_numpy(synth.rigid_offset(make_cloud(N, synth.SEED_B, floats))).to(dev)
        ix = capi.Index(tgt, auto_sync=False)
        ix.icp_align(src, max_iter=iters, fixed=True)
        ix.enable_timing(2)
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            T, fit, it, conv = ix.icp_align(src, max_iter=iters, fixed=True)
            best = min(best, time.perf_counter() - t0)
        tm = ix.timing()  # averages over the passes of the last calls: [0] NN kernel, [1] far/fallback, [2] whole pass
        ix.enable_timing(0)
        ix.close()
        passes = iters + 1  # + getFitnessScore
        pk = pairs.get("c4") if iters == 50 else None
        roof = None
        if pk:  # the NN kernels of all passes against the non-FMA fp32 roof, and the 8d bytes of a pass against the HBM roof
            nn_s = tm[0] * 1e-3 * passes
            top = pk["pairs_per_call"] * OPS_PER_PAIR / nn_s / 1e12 if nn_s > 0 else 0.0
            alg = passes * (12.0 * (M + N) + 8.0 * N + 24.0 * N)  # NN as k = 1 + transform 12 B read + 12 B write per source point
            roof = {"bound": "valu", "achieved": top, "peak": VALU_NOFMA_PEAK_TOPS, "unit": "Top/s", "frac": top / VALU_NOFMA_PEAK_TOPS,
                    "pairs_per_call": pk["pairs_per_call"], "pairs_per_pass": pk["pairs_per_call"] / passes,
                    "hbm": {"algorithmic_bytes": alg, "achieved": alg / best / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": alg / best / 1e9 / HBM_PEAK_GBS}}
        return {"roofline": roof,"workload": f"{desc}: pcc_icp_align(max_iter={iters}, fixed=1) = {iters} x (NN of every source point + "
                            "Umeyama sums + transform) + the fitness pass, source resident in HBM",
                "ms": best * 1e3, "iterations": int(it), "converged": bool(conv), "fitness": fit,
                "ms_per_pass": best * 1e3 / passes, "nn_queries_per_sec": N * passes / best,
                "split_ms_per_pass": {"nn_kernel": tm[0], "far_and_fallback": tm[1], "pass_total_events": tm[2],
                                      "sums_reduce_transform": max(tm[2] - tm[0] - tm[1], 0.0)}}

    # ---- the small-call regime: matchRIFTFeaturesKnn, call site #1 (src/comparator.cpp:560-588) ----------------------------
    def run_small_calls(sizes=(4, 100, 1000, 18381), c1=10_000):
        """What the reference does up to 3 x ~100 times per comparison: a tree over `n` descriptors of 128-byte stride in HOST
        memory, one k = 1 query per descriptor of a second cloud of the same size, matches below 0.05 returned -- through the
        C-ABI exactly as include/pcc/comparator_nn.hpp calls it (one handle, re-pointed per call: pcc_index_set_input +
        pcc_match_knn), in both tie modes, microseconds per call; the CPU oracle's per-call time beside it (tree build + queries,
        one thread).  n = 18 381 is the largest descriptor cloud in the reference's own result file (build/results.txt:15).
        C1 (10k x 10k XYZ from host memory, pcc_nn1) the same way."""
        import oracle
        rng = np.random.default_rng(0x51FF)
        rows = []
        ix = capi.Index(np.zeros((4, 32), np.float32), auto_sync=False)

        def per_call(fn, budget_s=0.25, kmin=5, kmax=2000):
            fn()
            t0 = time.perf_counter()
            fn()
            one = time.perf_counter() - t0
            k = int(max(kmin, min(kmax, budget_s / max(one, 1e-7))))
            t0 = time.perf_counter()
            for _ in range(k):
                fn()
            return (time.perf_counter() - t0) / k * 1e6

        for n in sizes:
            # descriptor clouds: normalised histograms (the searched first three bins lie in [0, 1], many of them equal)
            d1 = np.round(rng.random((n, 32), dtype=np.float32) * 16) / np.float32(64)
            d2_ = d1[rng.permutation(n)] + (rng.random((n, 32), dtype=np.float32) < 0.3) * np.float32(1.0 / 64)
            d2_ = np.ascontiguousarray(d2_, dtype=np.float32)
            row = {"n": n}
            want = oracle.match_rift_knn(d1, d2_)
            for mode, key in ((capi.TIES_LOWEST_INDEX, "gpu_us"), (capi.TIES_FLANN, "gpu_flann_ties_us")):
                ix.set_tie_order(mode)

                def call():
                    ix.set_input(d1)
                    return ix.match_knn(d2_)
                got = call()
                if mode == capi.TIES_FLANN:
                    row["matches_equal_oracle"] = bool(len(got) == len(want) and (got == want).all())
                row[key] = per_call(call)
            row["cpu_oracle_us"] = per_call(lambda: oracle.match_rift_knn(d1, d2_))
            row["gpu_over_cpu"] = row["gpu_flann_ties_us"] / row["cpu_oracle_us"]
            # the same in FLANN's tie order on descriptors WITHOUT exact ties (continuous values): no tied query, no tie replay
            c1_ = rng.random((n, 32), dtype=np.float32)
            c2_ = np.ascontiguousarray(c1_[rng.permutation(n)] + rng.random((n, 32), dtype=np.float32) * np.float32(0.01), dtype=np.float32)

            def call_c():
                ix.set_input(c1_)
                return ix.match_knn(c2_)
            got_c, want_c = call_c(), oracle.match_rift_knn(c1_, c2_)
            row["tie_free_matches_equal_oracle"] = bool(len(got_c) == len(want_c) and (got_c == want_c).all())
            row["gpu_flann_tie_free_us"] = per_call(call_c)
            rows.append(row)
        ix.close()
        # C1: 10k x 10k XYZ, k = 1, host memory in and out
        a, b = make_cloud(c1, synth.SEED_A, 3), make_cloud(c1, synth.SEED_B, 3)
        ix = capi.Index(a, auto_sync=False)

        def c1_call():
            ix.set_input(a)
            return ix.nn1(b)
        gi, gd = c1_call()
        kd = oracle.KdTree(a)
        ci, cd = kd.nn1_batch(b)
        c1_row = {"n": c1, "gpu_us": per_call(c1_call),
                  "cpu_oracle_us": per_call(lambda: oracle.KdTree(a).nn1_batch(b)),
                  "d2_bits_equal": bool((gd.view(np.uint32) == cd.view(np.uint32)).all()), "index_mismatches": int((gi != ci).sum())}
        ix.close()
        return {"what": "us per call from HOST memory through the C-ABI, handle reused (set_input + match_knn / nn1); CPU = oracle "
                        "kd-tree build + queries, 1 thread", "match_knn_128B": rows, "c1_nn1_10k": c1_row}

    # ---- the path a ./comparator user takes: clouds in HOST memory, results back to HOST memory ---------------------------
    def run_host_path(cfg="c3", reps=3):
        """The step of `cfg` with PCC_MEM_HOST inputs and outputs (pageable numpy arrays, as the reference's std::vectors are:
        src/comparator.cpp:1119,1130 load to host, :571-577 hand vectors over): wall clock per step, beside the device-resident
        step and the PCIe time of the bytes that have to cross (measured with a pinned copy of the same size in this run)."""
        M, N, floats, desc = CONFIGS[cfg]
        ref_h, qry_h = make_cloud(M, synth.SEED_A, floats), make_cloud(N, synth.SEED_B, floats)
        idx_h, d2_h = np.empty(N, np.int32), np.empty(N, np.float32)
        ix = capi.Index(ref_h, auto_sync=False)

        def step():
            ix.set_input(ref_h)
            ix.nn1(qry_h, idx_h, d2_h)
        step()
        best, build, query = 1e9, 1e9, 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            ix.set_input(ref_h)
            ix.sync()
            t1 = time.perf_counter()
            ix.nn1(qry_h, idx_h, d2_h)
            ix.sync()
            t2 = time.perf_counter()
            build, query = min(build, t1 - t0), min(query, t2 - t1)
            t0 = time.perf_counter()
            step()
            ix.sync()
            best = min(best, time.perf_counter() - t0)
        # the same step as rounds 1-5 took it: one hipMemcpyAsync of the raw pageable array each way (PCC_OPT_HOST_PIPE = 0)
        ix.set_option(capi.OPT_HOST_PIPE, 0)
        step()
        ix.sync()
        t0 = time.perf_counter()
        step()
        ix.sync()
        plain = time.perf_counter() - t0
        ix.close()
        # PCIe: one pinned H2D of the raw cloud's size, one pinned D2H of the results' size
        pin = torch.empty(M * floats, dtype=torch.float32).pin_memory()
        dv = torch.empty(M * floats, dtype=torch.float32, device=dev)
        dv.copy_(pin, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        dv.copy_(pin, non_blocking=True); torch.cuda.synchronize()
        h2d = time.perf_counter() - t0
        pin2 = torch.empty(N * 2, dtype=torch.float32).pin_memory()
        t0 = time.perf_counter()
        pin2.copy_(dv[:N * 2], non_blocking=True); torch.cuda.synchronize()
        d2h = time.perf_counter() - t0
        raw = M * floats * 4
        del pin, dv, pin2
        torch.cuda.empty_cache()
        return {"workload": f"{desc} from and to HOST memory (pageable)", "step_ms": best * 1e3, "step_ms_plain_hipmemcpy": plain * 1e3,
                "host_threads": int(os.environ.get("PCC_HOST_THREADS", min(8, max(2, (os.cpu_count() or 4) // 2)))), "set_input_ms": build * 1e3,
                "nn1_ms": query * 1e3, "raw_bytes_per_cloud": raw, "pinned_h2d_GBps": raw / h2d / 1e9,
                "pcie_floor_ms_raw": (2 * h2d + d2h) * 1e3,
                "pcie_floor_ms_xyz_only": (2 * h2d * 12.0 / (floats * 4) + d2h) * 1e3}

    # ---- the measured workload ---------------------------------------------------------------------------------------
    # C3 / C5: a FIXED total of queries sharded over the ranks present (north_star's partition: strong scaling, N = 1 is the
    # whole configuration on one GPU); the other configurations keep their size on every rank (weak)
    strong_total = {"c3": C3_TOTAL_QUERIES, "c5": C5_TOTAL_QUERIES}.get(primary_cfg)
    if primary_cfg == "c4":
        prim = None
        icp = run_icp()
        value, ms_per_step, workload = icp["nn_queries_per_sec"], icp["ms"], icp["workload"]
    else:
        prim = run_nn(primary_cfg, total_queries=strong_total,
                      with_exhaustive=(not args.no_exhaustive and args.config != "auto"), with_cpu=not args.no_cpu,
                      project=(args.config == "auto" and not args.no_extra))
        value, ms_per_step, workload = prim["value"], prim["ms_per_step"], prim["workload"]

    out = {
        "metric": "nn_queries_per_sec", "value": value, "unit": "queries/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong_total else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "name": primary_cfg, "backend": backend_name,
                   "parallelism": f"query-sharded x{n_gpus}, reference cloud replicated "
                                  f"({bcast_name} broadcast of the packed cloud, outside the timed region)"},
    }
    if prim is not None:
        for k in ("engine", "references", "queries_per_gpu", "queries_total", "point_stride_bytes"):
            out["config"][k] = prim[k]
        for k in ("scaling_terms", "query_only_queries_per_sec", "build_ms", "search_call_ms", "query_sort_ms", "fallback_queries",
                  "broadcast_ms", "broadcast_bytes", "per_rank_ms_per_step", "roofline", "exhaustive", "cpu_baseline"):
            if k in prim:
                out[k] = prim[k]
    else:
        out["icp"] = icp

    if args.config == "auto" and not args.no_extra:
        extra = {}
        if n_gpus == 1:
            extra["c2"] = run_nn("c2", with_exhaustive=not args.no_exhaustive, with_cpu=False)
            ex = extra["c2"].get("exhaustive")
            if ex and "roofline" in out:
                # the kernel north_star describes (tiled exhaustive k = 1), on C2's data, beside the pruned search's figures:
                # top level, so that a truncated line still carries it
                out["roofline"]["exhaustive"] = {"kernel": "k_nn1_brute", "workload": "C2 1M x 1M", "bound": "valu",
                                                 "kernel_ms": ex["roofline"]["kernel_ms"], "achieved": ex["roofline"]["achieved"],
                                                 "peak": VALU_NOFMA_PEAK_TOPS, "unit": "Top/s", "frac": ex["roofline"]["frac"],
                                                 "pairs_per_sec": ex["pairs_per_sec"], "bit_identical_to_grid": ex["bit_identical_to_grid"]}
            extra["c3_clusters"] = run_clusters()
            extra["c4_icp"] = run_icp()
            extra["room"] = {"scan": run_room(synth.ROOM_SIZES[1]), "10M": run_room(synth.ROOM_SIZES[2], full=False)}
            extra["small_calls"] = run_small_calls()
            extra["host_path"] = {"c3": run_host_path("c3")}
            # what ONE of eight GPUs does in BASELINE configs[4] (a 4M-query shard vs the 8M references)
            shard = run_nn("c5_shard", steps=min(K, 10), warmup=2)
            shard["scaling"] = "one shard of the 8-GPU configuration, measured on one GPU"
            extra["c5_shard"] = shard
        else:
            # the weak form beside the headline: every rank its own 10M queries against the replicated references
            weak = run_nn("c3", steps=min(K, 10), warmup=2)
            weak["scaling"] = "weak (10M queries per GPU)"
            extra["c3_weak"] = weak
        # BASELINE configs[4]: 32M queries vs 8M references, sharded over the ranks present -- at EVERY N, one GPU
        # included (all 32M queries there), so that the strong-scaling curve has its N = 1 anchor
        c5 = run_nn("c5", total_queries=C5_TOTAL_QUERIES, steps=min(K, 10), warmup=2, project=True)
        c5["scaling"] = "strong (32M queries in total, whatever the rank count)"
        extra["c5"] = c5
        if n_gpus == 1:
            # what the one-GPU anchors say about 1 -> 8: the index build is replicated, so the whole step follows Amdahl while
            # the query half alone comes close to linear.  A projection (labelled so in every entry), never a measurement.
            extra["scaling_projection"] = {"c3": prim.get("projection") if prim else None, "c5": c5.pop("projection", None)}
            if prim:
                prim.pop("projection", None)
        out["extra"] = extra

    # the C-ABI's own RCCL path (pcc_comm_create_rank + pcc_index_create_broadcast: what a C++ host without torch uses),
    # beside torch.distributed's: the same cloud broadcast through libpcc_nn, one shard searched, bits compared with the
    # index built from torch's broadcast.  Opt-in (PCC_BENCH_CABI_COMM=1; tests/test_bench_gpu.py runs it with one rank):
    # a second communicator on an untried node is not worth risking the headline line for
    if dist is not None and os.environ.get("PCC_BENCH_CABI_COMM", "0") == "1" and backend == "nccl":
        M, N = 1_000_000, 250_000
        uid = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = capi.Comm.from_id(uid[0], n_gpus, rank, dev_index)
        ref_host = make_cloud(M, synth.SEED_A, 3) if rank == 0 else None
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        bx = capi.Index.broadcast(comm, 0, ref_host)
        ms = max_over_ranks((time.perf_counter() - t0) * 1e3)
        ref_t = torch.empty((M, 3), dtype=torch.float32, device=dev)
        if rank == 0:
            ref_t.copy_(torch.from_numpy(ref_host))
        dist.broadcast(ref_t, src=0)
        q = torch.from_numpy(make_cloud(N, synth.SEED_B, 3, start=rank * N)).to(dev)
        with capi.Index(ref_t) as tx:
            i1, d1 = tx.nn1(q)
            i2, d2_ = bx.nn1(q)
            same = bool((i1 == i2).all().item() and (d1.view(torch.int32) == d2_.view(torch.int32)).all().item())
        ok = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        out.setdefault("extra", {})["c_abi_comm"] = {"ranks": comm.info()[1], "references": M, "create_broadcast_ms": ms,
                                                       "shard_results_equal_torch_broadcast_index": bool(ok.item() == 1.0)}
        bx.close()
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the full record goes to a file (whoever wants every leg's detail reads it there); the LINE is what the driver keeps --
        # the contract's keys, flat scalars, one note: under 7 KB
        full = compact(out)
        for d in (ROOT / "gpurun_out", Path(os.environ.get("TMPDIR", "/tmp"))):
            try:
                d.mkdir(exist_ok=True)
                (d / "bench_full.json").write_text(json.dumps(full, indent=1))
                break
            except Exception:
                continue
        print(json.dumps(compact(short_line(out)), separators=(",", ":")))


if __name__ == "__main__":
    main()
