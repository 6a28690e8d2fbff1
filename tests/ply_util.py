"""write small PLY files (ascii / binary_little_endian) for the CLI and reader tests"""
import numpy as np


def write_ply(path, pts, rgb=None, fmt="binary", extra_face=False, double=False):
    pts = np.asarray(pts)
    n = len(pts)
    rgb = np.zeros((n, 3), np.uint8) if rgb is None else np.asarray(rgb, np.uint8)
    ft = "double" if double else "float"
    hdr = ["ply", f"format {'ascii' if fmt == 'ascii' else 'binary_little_endian'} 1.0", "comment made by tests/ply_util.py",
           f"element vertex {n}", f"property {ft} x", f"property {ft} y", f"property {ft} z",
           "property uchar red", "property uchar green", "property uchar blue"]
    if extra_face:
        hdr += ["element face 1", "property list uchar int vertex_indices"]
    hdr.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        if fmt == "ascii":
            for p, c in zip(pts, rgb):
                f.write((" ".join(repr(float(v)) if np.isfinite(v) else ("nan" if np.isnan(v) else ("inf" if v > 0 else "-inf"))
                                  for v in p) + f" {c[0]} {c[1]} {c[2]}\n").encode())
            if extra_face:
                f.write(b"3 0 1 2\n")
        else:
            dt = np.dtype([("p", "<f8" if double else "<f4", 3), ("c", "u1", 3)])
            rec = np.zeros(n, dt)
            rec["p"] = pts
            rec["c"] = rgb
            f.write(rec.tobytes())
            if extra_face:
                f.write(bytes([3]) + np.array([0, 1, 2], "<i4").tobytes())
