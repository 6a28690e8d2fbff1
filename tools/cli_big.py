import sys, time, subprocess, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from ply_util import write_ply
rng = np.random.default_rng(1)
def room(n):
    per = n // 5
    def wall(o, e1, e2):
        uv = rng.random((per, 2)) * 4
        return np.asarray(o) + uv[:, :1] * np.asarray(e1) + uv[:, 1:] * np.asarray(e2)
    parts = [wall((0,0,0),(1,0,0),(0,1,0)), wall((0,0,0),(1,0,0),(0,0,0.6)), wall((0,0,0),(0,1,0),(0,0,0.6))]
    for c in [(1,1,0.5),(2.5,1.5,0.6),(1.5,3,0.4),(3,3,0.7)]:
        d = rng.normal(size=(per // 2, 3)); parts.append(d / np.linalg.norm(d, axis=1, keepdims=True) * 0.3 + c)
    p = np.concatenate(parts) + rng.normal(0, 0.002, (sum(len(x) for x in parts), 3))
    return np.ascontiguousarray(p[rng.permutation(len(p))].astype(np.float32))
a = room(500000); b = room(500000) + np.float32([0.004, -0.003, 0.002])
write_ply("/tmp/a.ply", a, fmt="binary"); write_ply("/tmp/b.ply", b, fmt="binary")
for flags in (["-i", "-e", "-n"], ["-n"]):
    t = time.perf_counter()
    r = subprocess.run(["build/comparator", *flags, "/tmp/a.ply", "/tmp/b.ply", "--results", "/tmp/res.txt"], capture_output=True, text=True, timeout=600)
    dt = time.perf_counter() - t
    print(flags, f"{dt:.2f} s rc={r.returncode}")
    print("\n".join(l for l in r.stdout.splitlines() if any(k in l for k in ("filtering", "planar", "Cluster", "clusters", "converged", "Noise pass")))[:1200])
