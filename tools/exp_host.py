"""dev helper (round 6): the C3 step from and to pageable HOST memory (set_input + nn1 with numpy arrays), wall clock per step, by
number of staging threads (PCC_HOST_THREADS is read per transfer) and with PCC_OPT_HOST_PIPE off.  usage: exp_host.py [n] [threads ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pointcloudcomparator_amd import capi, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
threads = [int(x) for x in sys.argv[2:]] or [4, 8, 12, 16]


def cloud(seed):
    parts = [synth.corridor_cloud(min(4_000_000, n - o), seed, start=o) for o in range(0, n, 4_000_000)]
    return synth.with_rgb_stride(np.concatenate(parts))


a, b = cloud(synth.SEED_A), cloud(synth.SEED_B)
idx, d2 = np.empty(n, np.int32), np.empty(n, np.float32)
ix = capi.Index(a[:4096], auto_sync=False)


def step():
    ix.set_input(a)
    ix.nn1(b, idx, d2)


def timed(k=3):
    step(); ix.sync()
    best = 1e9
    for _ in range(k):
        t0 = time.perf_counter(); ix.set_input(a); ix.sync(); t1 = time.perf_counter(); ix.nn1(b, idx, d2); ix.sync(); t2 = time.perf_counter()
        best = min(best, t2 - t0)
        parts = (t1 - t0, t2 - t1)
    return best, parts


for t in threads:
    os.environ["PCC_HOST_THREADS"] = str(t)
    best, parts = timed()
    print(f"host pipe, {t:2d} threads: step {best * 1e3:7.2f} ms  (set_input {parts[0] * 1e3:6.2f}, nn1 {parts[1] * 1e3:6.2f})", flush=True)
ix.set_option(capi.OPT_HOST_PIPE, 0)
best, parts = timed(1)
print(f"plain hipMemcpyAsync:   step {best * 1e3:7.2f} ms  (set_input {parts[0] * 1e3:6.2f}, nn1 {parts[1] * 1e3:6.2f})", flush=True)
