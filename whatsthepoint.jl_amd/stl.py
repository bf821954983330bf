"""Binary STL reader: one boundary element per face — centroid, unit normal, area — the way the
reference's import_surface builds a PointBoundary (src/io.jl:36-56: `centroid`, the file's facet
normals normalised, `Meshes.area`).  84-byte header + 50 B per triangle, little-endian float32."""
from __future__ import annotations

import numpy as np


def read_binary_stl(path: str):
    with open(path, "rb") as f:
        f.seek(80)
        n = int(np.frombuffer(f.read(4), dtype="<u4")[0])
        rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
        data = np.frombuffer(f.read(n * 50), dtype=rec, count=n)
    return data["v"].astype(np.float32)  # (n, 3 vertices, 3 coords)


def read_binary_stl_normals(path: str):
    with open(path, "rb") as f:
        f.seek(80)
        n = int(np.frombuffer(f.read(4), dtype="<u4")[0])
        rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
        data = np.frombuffer(f.read(n * 50), dtype=rec, count=n)
    return data["n"].astype(np.float32)


def surface_elements(path: str, dtype=np.float32):
    """(centroids, unit normals, areas) of the faces, in file order (src/io.jl:44-55).  Facets whose
    stored normal is zero get the geometric normal of their vertices."""
    v = read_binary_stl(path).astype(np.float64)
    nf = read_binary_stl_normals(path).astype(np.float64)
    cr = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    area = 0.5 * np.linalg.norm(cr, axis=1)
    ln = np.linalg.norm(nf, axis=1)
    bad = ~(ln > 0)
    nf[bad] = cr[bad]
    ln = np.linalg.norm(nf, axis=1)
    nf = nf / np.where(ln > 0, ln, 1.0)[:, None]
    return v.mean(axis=1).astype(dtype), nf.astype(dtype), area.astype(dtype)


def face_centroids(path: str, dtype=np.float32):
    v = read_binary_stl(path).astype(np.float64)
    return v.mean(axis=1).astype(dtype)
