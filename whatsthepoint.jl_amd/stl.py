"""Binary STL reader for the plumbing config (C1): face centroids as boundary points, the way
the reference's import makes one boundary point per face (src/io.jl:27-56).  84-byte header +
50 B per triangle, little-endian float32."""
from __future__ import annotations

import numpy as np


def read_binary_stl(path: str):
    with open(path, "rb") as f:
        f.seek(80)
        n = int(np.frombuffer(f.read(4), dtype="<u4")[0])
        rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
        data = np.frombuffer(f.read(n * 50), dtype=rec, count=n)
    return data["v"].astype(np.float32)  # (n, 3 vertices, 3 coords)


def face_centroids(path: str, dtype=np.float32):
    v = read_binary_stl(path).astype(np.float64)
    return v.mean(axis=1).astype(dtype)
