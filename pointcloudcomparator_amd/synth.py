"""Deterministic synthetic clouds for the BASELINE.json configs.

Counter-based generator: value j of point i is splitmix64(seed, 8*i + j) mapped to a
24-bit-mantissa uniform float, so any (seed, index) is reproducible without state.
Scene: a "corridor" box (ranges taken from the reference's sample centroids,
build/results.txt:14-120) holding a uniform background layer (50 % of the points)
and an object layer (50 %) of 256 balls of radius 0.30 m centred on a 1 m lattice.
"""
from __future__ import annotations

import numpy as np

SEED_A = 0x5EED0001
SEED_B = 0x5EED0002
BOX_LO = np.array([-3.0, -28.0, -0.1])
BOX_HI = np.array([8.0, 4.0, 2.7])
N_BALLS = 256
BALL_R = 0.30

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, ctr: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (ctr.astype(np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform24(seed: int, ctr: np.ndarray) -> np.ndarray:
    """uniform in [0,1) with 24 random mantissa bits (exactly representable in fp32)."""
    return (splitmix64(seed, ctr) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def ball_centres() -> np.ndarray:
    """256 lattice sites (1 m pitch) whose 0.30 m balls lie inside the box; the choice of
    sites is seed-independent so clouds A and B sample the same scene."""
    xs = np.arange(-2.0, 7.01, 1.0)
    ys = np.arange(-27.0, 3.01, 1.0)
    zs = np.array([0.5, 1.5])
    g = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), -1).reshape(-1, 3)
    key = splitmix64(0xBA115, np.arange(len(g), dtype=np.uint64))
    pick = np.sort(np.argsort(key, kind="stable")[:N_BALLS])
    return g[pick]


def corridor_cloud(n: int, seed: int, layer: str = "both", start: int = 0) -> np.ndarray:
    """(n, 3) float32.  layer: 'both' | 'background' | 'objects'.  Even point indices are
    background, odd are object points, so any prefix keeps the 50/50 mix."""
    i = np.arange(start, start + n, dtype=np.uint64)
    u = [uniform24(seed, i * np.uint64(8) + np.uint64(j)) for j in range(4)]
    bg = BOX_LO + np.stack(u[:3], -1) * (BOX_HI - BOX_LO)
    c = ball_centres()
    ball = (splitmix64(seed ^ 0xB411, i) % np.uint64(N_BALLS)).astype(np.int64)
    r = BALL_R * np.cbrt(u[0])
    ct = 2.0 * u[1] - 1.0
    st = np.sqrt(np.maximum(0.0, 1.0 - ct * ct))
    ph = 2.0 * np.pi * u[2]
    obj = c[ball] + np.stack([r * st * np.cos(ph), r * st * np.sin(ph), r * ct], -1)
    if layer == "background":
        pts = bg
    elif layer == "objects":
        pts = obj
    else:
        pts = np.where(((i & np.uint64(1)) == 0)[:, None], bg, obj)
    return np.ascontiguousarray(pts.astype(np.float32))


def rigid_offset(points: np.ndarray, seed: int = 0x1C9, rot_deg: float = 2.0,
                 t=(0.03, -0.02, 0.01), jitter: float = 0.005) -> np.ndarray:
    """cloud B of the ICP config: rotate about z, translate, add N(0, jitter) noise."""
    a = np.deg2rad(rot_deg)
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    n = len(points)
    i = np.arange(n, dtype=np.uint64)
    u1 = np.maximum(uniform24(seed, i * np.uint64(8)), 2.0 ** -24)
    g = []
    for j in range(3):  # Box-Muller per axis
        u2 = uniform24(seed, i * np.uint64(8) + np.uint64(1 + j))
        uu = np.maximum(uniform24(seed, i * np.uint64(8) + np.uint64(4 + j)), 2.0 ** -24)
        g.append(np.sqrt(-2.0 * np.log(uu)) * np.cos(2.0 * np.pi * u2))
    del u1
    out = points.astype(np.float64) @ R.T + np.asarray(t) + jitter * np.stack(g, -1)
    return np.ascontiguousarray(out.astype(np.float32))


def with_rgb_stride(points: np.ndarray) -> np.ndarray:
    """(n, 8) float32 view-compatible with pcl::PointXYZRGB: x,y,z,pad,rgb,pad,pad,pad (32 B)."""
    n = len(points)
    out = np.zeros((n, 8), dtype=np.float32)
    out[:, :3] = points
    out[:, 3] = 1.0
    rgb = (splitmix64(0xC0104, np.arange(n, dtype=np.uint64)) & np.uint64(0xFFFFFF)).astype(np.uint32)
    out[:, 4] = rgb.view(np.float32)
    return out
