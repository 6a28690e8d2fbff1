"""dev helper (round 6): same-box A/B of option sets on the k = 1 step, interleaved round-robin so that drift of the box cancels.
usage: exp_ab.py <config: c2|c3|room> "grid_axes=0,fuse_params=0" "grid_axes=-1" ...   (option names as in capi.OPT_*)
Prints the median over ROUNDS (default 5) of: step, build alone, query-only (wall clock over 20 calls each)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudcomparator_amd import capi, synth

cfg = sys.argv[1]
sets = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in s.split(",") if kv) for s in sys.argv[2:]]
if ":" in cfg:  # "corr:2e6" / "room:3e6": that many points of the corridor / room scene, 12-byte stride
    kind, n, rgb = ("both" if cfg.split(":")[0] == "corr" else "room"), int(float(cfg.split(":")[1])), False
else:
    n, kind, rgb = {"c2": (1_000_000, "both", False), "c3": (10_000_000, "both", True), "room": (1_379_736, "room", False)}[cfg]
rounds = int(os.environ.get("ROUNDS", "5"))


def cloud(seed):
    gen = synth.room_cloud if kind == "room" else synth.corridor_cloud
    parts = [gen(min(4_000_000, n - o), seed, start=o) for o in range(0, n, 4_000_000)]
    pts = parts[0] if len(parts) == 1 else np.concatenate(parts)
    return torch.from_numpy(synth.with_rgb_stride(pts) if rgb else pts).cuda()


a, b = cloud(synth.SEED_A), cloud(synth.SEED_B)
idx = torch.empty(n, dtype=torch.int32, device="cuda")
d2 = torch.empty(n, dtype=torch.float32, device="cuda")
# ONE handle for every option set: its buffers only grow, so every set runs in the same memory (a handle per set puts each set's
# index at other addresses, and that alone moved steps by several percent -- round 6)
ix = capi.Index(a, engine=capi.ENGINE_GRID, auto_sync=False)
keys = sorted({k for st in sets for k in st})
defaults = {k: ix.get_option(getattr(capi, "OPT_" + k.upper())) for k in keys}


def apply(st):
    for k in keys:
        ix.set_option(getattr(capi, "OPT_" + k.upper()), st.get(k, defaults[k]))
    for _ in range(2):
        ix.set_input(a)
        ix.nn1(b, idx, d2)
    ix.sync()


def wall(fn, k=20):
    fn(); ix.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    ix.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


def step():
    ix.set_input(a)
    ix.nn1(b, idx, d2)


res = [([], [], []) for _ in sets]
for r in range(rounds):
    for i, st in enumerate(sets):
        apply(st)
        res[i][0].append(wall(step))
        res[i][1].append(wall(lambda: ix.set_input(a)))
        res[i][2].append(wall(lambda: ix.nn1(b, idx, d2)))
for st, (s_, b_, q_) in zip(sets, res):
    print(f"{cfg} {str(st):60s} step {statistics.median(s_):7.4f} ms (min {min(s_):7.4f})  build {statistics.median(b_):7.4f}  query-only {statistics.median(q_):7.4f}", flush=True)
