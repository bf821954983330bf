"""Surface normals (src/normals.jl) and `split_surface!` (src/surface_operations.jl:58-94) — consumers of
the k-NN rows.

`compute_normals` / `update_normals` run on the GPU end to end (k-NN rows + per-point PCA,
csrc/wtp_consumers.hip).  `orient_normals` and `split_surface` take their k-NN rows from the device and
do the graph part — a minimum spanning tree walk, connected components — on the calling thread, where
the reference does it too (Graphs.jl `kruskal_mst`, `connected_components`; "TODO below is slow",
src/normals.jl:95)."""
from __future__ import annotations

import numpy as np

from .cloud import PointBoundary, PointCloud, PointSurface
from .engine import default_context


def _pts(x):
    return x.points() if hasattr(x, "points") else np.asarray(x)


def compute_normals(points, k: int = 5, ctx=None):
    """compute_normals(points|surf; k) (src/normals.jl:8-46): unit normals, sign not yet oriented."""
    p = np.ascontiguousarray(_pts(points))
    k = min(int(k), len(p))
    return (ctx or default_context()).pca_normals(p, k)


def update_normals(surf: PointSurface, k: int = 5, ctx=None):
    """update_normals!(surf; k) (src/normals.jl:57-63)."""
    surf.normals = compute_normals(surf, k, ctx)
    return surf.normals


class _Dsu:
    def __init__(self, n):
        self.p = np.arange(n)

    def find(self, a):
        p = self.p
        r = a
        while p[r] != r:
            r = p[r]
        while p[a] != r:
            p[a], a = r, p[a]
        return r


def _rows(p, k, ctx):
    return (ctx or default_context()).knn(np.ascontiguousarray(p), k, include_self=True)


def orient_normals(normals, points=None, k: int = 5, ctx=None):
    """orient_normals!(normals, points; k) (src/normals.jl:79-147), in place: minimum spanning tree of the
    k-NN graph weighted by 1 - |n_i . n_j|, the highest point (last coordinate) made to face up, signs
    propagated along the tree.  Accepts a PointSurface / PointCloud as the only argument, too."""
    if points is None and isinstance(normals, PointCloud):
        for s in normals.boundary.surfaces.values():
            orient_normals(s, k=k, ctx=ctx)
        return None
    if points is None and isinstance(normals, PointSurface):
        surf = normals
        orient_normals(surf.normals, surf.points(), k=k, ctx=ctx)
        return None
    p = np.ascontiguousarray(_pts(points))
    nrm = normals
    n = len(p)
    k = min(int(k), n)
    rows = _rows(p, k, ctx)
    src = np.repeat(rows[:, 0], k - 1).astype(np.int64)        # n[1] of the reference (the query itself)
    dst = rows[:, 1:].reshape(-1).astype(np.int64)
    eps = np.finfo(nrm.dtype).eps * 1.0e2                        # build_normal_weighted_graph (:150-161)
    w = 1.0 - np.abs(np.einsum("ij,ij->i", nrm[src], nrm[dst])) + eps
    # undirected simple graph: the later add_edge! of a pair overwrites the earlier one (same weight up to rounding)
    a, b = np.minimum(src, dst), np.maximum(src, dst)
    key = a * n + b
    _, last = np.unique(key[::-1], return_index=True)
    sel = len(key) - 1 - last
    a, b, w = a[sel], b[sel], w[sel]
    order = np.argsort(w, kind="stable")
    dsu = _Dsu(n)
    adj_a, adj_b = [], []
    for e in order:                                              # kruskal_mst
        ra, rb = dsu.find(int(a[e])), dsu.find(int(b[e]))
        if ra != rb:
            dsu.p[ra] = rb
            adj_a.append(int(a[e]))
            adj_b.append(int(b[e]))
            if len(adj_a) == n - 1:
                break
    ea = np.array(adj_a + adj_b, dtype=np.int64)
    eb = np.array(adj_b + adj_a, dtype=np.int64)
    o = np.argsort(ea, kind="stable")
    ea, eb = ea[o], eb[o]
    start_of = np.searchsorted(ea, np.arange(n + 1))
    start = int(np.argmax(p[:, -1]))
    if nrm[start, -1] < 0:
        nrm[start] = -nrm[start]
    visited = np.zeros(n, dtype=bool)
    visited[start] = True
    stack = [(start, start)]
    while stack:                                                 # tree walk: every vertex after its parent
        v, parent = stack.pop()
        if float(np.dot(nrm[v], nrm[parent])) < 0:
            nrm[v] = -nrm[v]
        for u in eb[start_of[v]:start_of[v + 1]]:
            if not visited[u]:
                visited[u] = True
                stack.append((int(u), v))
    return None


def _angle(u, v):
    """_angle (src/utils.jl:18-23): 2-D signed, 3-D unsigned; radians, row-wise."""
    if u.shape[1] == 2:
        th = np.arctan2(u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0], (u * v).sum(axis=1))
        return np.where(th == -np.pi, -th, th)
    return np.arctan2(np.linalg.norm(np.cross(u, v), axis=1), (u * v).sum(axis=1))


def combine_surfaces(cloud, *surfs):
    """combine_surfaces!(boundary, surfs...) (src/surface_operations.jl:7-31): the named surfaces are merged,
    in the boundary's own order, into one surface that takes the first given name."""
    bnd = cloud.boundary if isinstance(cloud, PointCloud) else cloud
    for name in surfs:
        assert name in bnd.surfaces, "Surface does not exist. Check spelling."
    parts = [bnd.surfaces[name] for name in bnd.surfaces if name in surfs]
    pts = np.concatenate([s.points() for s in parts])
    nrm = None if any(s.normals is None for s in parts) else np.concatenate([s.normals for s in parts])
    areas = None if any(s.areas is None for s in parts) else np.concatenate([s.areas for s in parts])
    for name in surfs:
        del bnd.surfaces[name]
    bnd.surfaces[surfs[0]] = PointSurface(pts, nrm, areas)
    return None


def split_surface(cloud, angle: float, target=None, k: int = 10, ctx=None):
    """split_surface!(cloud|boundary, [target], angle; k) (src/surface_operations.jl:33-94): splits a surface
    into the connected components of its k-NN graph restricted to edges whose normals differ by less than
    `angle` (radians).  target: a surface name, a PointSurface already taken out of the boundary, or None
    (the boundary must then hold exactly one surface)."""
    bnd = cloud.boundary if isinstance(cloud, PointCloud) else cloud
    assert isinstance(bnd, PointBoundary)
    if target is None:
        assert len(bnd.surfaces) == 1, "More than 1 surface in this cloud. Please specify a target surface."
        target = next(iter(bnd.surfaces))
    if isinstance(target, PointSurface):
        surf = target
    else:
        assert target in bnd.surfaces, "Target surface not found in cloud."
        surf = bnd.surfaces.pop(target)
    p, nrm, areas = surf.points(), surf.normals, surf.areas
    assert nrm is not None, "split_surface needs normals"
    n = len(p)
    kk = min(int(k), n)
    rows = _rows(p, kk, ctx)
    src = np.repeat(rows[:, 0], kk - 1).astype(np.int64)
    dst = rows[:, 1:].reshape(-1).astype(np.int64)
    keep = np.abs(_angle(nrm[src].astype(np.float64), nrm[dst].astype(np.float64))) < angle
    src, dst = src[keep], dst[keep]
    dsu = _Dsu(n)
    for a, b in zip(src.tolist(), dst.tolist()):
        ra, rb = dsu.find(a), dsu.find(b)
        if ra != rb:                                             # the smaller root wins: label = smallest member
            if ra < rb:
                dsu.p[rb] = ra
            else:
                dsu.p[ra] = rb
    label = np.array([dsu.find(i) for i in range(n)])
    for comp in np.unique(label):                                # connected_components: by first vertex
        ids = np.nonzero(label == comp)[0]
        i = 1
        while f"surface{i}" in bnd.surfaces:                     # _generate_surface_name (:95-102)
            i += 1
        bnd.surfaces[f"surface{i}"] = PointSurface(p[ids], nrm[ids], None if areas is None else areas[ids])
    return cloud
