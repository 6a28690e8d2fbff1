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


# ---- a surface-sampled scene: what the reference's real inputs look like --------------------------------------------
# The reference compares two scans of a room (build/results.txt:4-8: 346 911 and 346 921 points in 74 / 65 clusters):
# points on 2-D surfaces, not in a volume.  A 6 x 5 x 2.5 m room (75 m^3; floor, ceiling, four walls = 115 m^2) with
# a few pieces of box-shaped furniture (five visible faces each), sampled uniformly by area, N(0, 2 mm) of jitter along
# the surface normal.  ROOM_SIZES: the two cloud sizes of build/results.txt:4-5 and north_star's 10M.
ROOM = np.array([6.0, 5.0, 2.5])
ROOM_JITTER = 0.002
ROOM_SIZES = (346_911, 1_379_736, 10_000_000)
_ROOM_BOXES = (  # (x0, y0, x1, y1, height): furniture standing on the floor
    (0.3, 0.3, 2.3, 1.2, 0.75), (3.0, 0.2, 3.6, 0.8, 1.9), (4.4, 3.6, 5.8, 4.8, 0.45), (0.2, 3.4, 0.8, 4.8, 2.0),
    (2.4, 2.0, 3.6, 2.9, 0.72), (5.2, 0.3, 5.8, 2.3, 0.9))


def _room_faces():
    """list of (origin, edge u, edge v, unit normal): rectangles origin + s u + t v, s, t in [0, 1)"""
    X, Y, Z = ROOM
    f = [((0, 0, 0), (X, 0, 0), (0, Y, 0), (0, 0, 1)), ((0, 0, Z), (X, 0, 0), (0, Y, 0), (0, 0, 1)),
         ((0, 0, 0), (X, 0, 0), (0, 0, Z), (0, 1, 0)), ((0, Y, 0), (X, 0, 0), (0, 0, Z), (0, 1, 0)),
         ((0, 0, 0), (0, Y, 0), (0, 0, Z), (1, 0, 0)), ((X, 0, 0), (0, Y, 0), (0, 0, Z), (1, 0, 0))]
    for x0, y0, x1, y1, h in _ROOM_BOXES:
        f += [((x0, y0, h), (x1 - x0, 0, 0), (0, y1 - y0, 0), (0, 0, 1)),
              ((x0, y0, 0), (x1 - x0, 0, 0), (0, 0, h), (0, 1, 0)), ((x0, y1, 0), (x1 - x0, 0, 0), (0, 0, h), (0, 1, 0)),
              ((x0, y0, 0), (0, y1 - y0, 0), (0, 0, h), (1, 0, 0)), ((x1, y0, 0), (0, y1 - y0, 0), (0, 0, h), (1, 0, 0))]
    return [tuple(np.asarray(v, np.float64) for v in face) for face in f]


def room_cloud(n: int, seed: int, start: int = 0, part: str = "all") -> np.ndarray:
    """(n, 3) float32: point i lies on the face its first random number picks (by area), at (s, t) from the next two,
    pushed along the normal by a Box-Muller normal from the last two.  Counter-based like corridor_cloud.
    part = 'furniture': only the box faces (what is left of a room once the planes are removed, the input of the
    reference's -e clustering, src/segmentation.cpp:79-131): len(_ROOM_BOXES) clusters at tolerance 0.05."""
    faces = _room_faces()
    if part == "furniture":
        faces = faces[6:]
    area = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v, _ in faces])
    cum = np.cumsum(area) / area.sum()
    i = np.arange(start, start + n, dtype=np.uint64)
    u = [uniform24(seed ^ 0x400F, i * np.uint64(8) + np.uint64(j)) for j in range(5)]
    k = np.minimum(np.searchsorted(cum, u[0], side="right"), len(faces) - 1)
    org = np.stack([f[0] for f in faces])[k]
    eu = np.stack([f[1] for f in faces])[k]
    ev = np.stack([f[2] for f in faces])[k]
    nr = np.stack([f[3] for f in faces])[k]
    g = np.sqrt(-2.0 * np.log(np.maximum(u[3], 2.0 ** -24))) * np.cos(2.0 * np.pi * u[4])
    pts = org + u[1][:, None] * eu + u[2][:, None] * ev + (ROOM_JITTER * g)[:, None] * nr
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
