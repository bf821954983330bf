"""Point containers — only what the hot path touches (src/surface.jl:167-205,
src/boundary.jl:164, src/volume.jl:97-125, src/cloud.jl:178-237): coordinates as numpy
arrays, `points()` = boundary first then volume (the global index space), and the functional
`set_topology` / in-place `rebuild_topology` verbs.  Surfaces optionally carry the per-point
normals and areas the `isinside` filter of `repel` integrates over (src/isinside.jl:86-106);
units and I/O stay in Julia."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from . import topology as T
from .engine import default_context


def _as_points(p, dtype=None):
    p = np.asarray(p)
    if p.ndim != 2 or p.shape[1] not in (2, 3):
        raise ValueError("points must have shape (n, 2) or (n, 3)")
    if dtype is not None:
        p = p.astype(dtype, copy=False)
    elif p.dtype not in (np.float32, np.float64):
        p = p.astype(np.float64)
    return np.ascontiguousarray(p)


class _HasTopology:
    topology: T.AbstractTopology

    def hastopology(self) -> bool:
        return not isinstance(self.topology, T.NoTopology)

    def neighbors(self, i=None):
        return T.neighbors(self.topology, i)

    def _with(self, topo):
        raise NotImplementedError

    def set_topology(self, kind, param, ctx=None):
        """Functional: returns a NEW container carrying the built topology
        (src/cloud.jl:200-217, src/surface.jl:167-194, src/volume.jl:97-114)."""
        ctx = ctx or default_context()
        pts = self.points()
        if kind is T.KNNTopology:
            return self._with(T.KNNTopology(T.build_knn_neighbors(ctx, pts, int(param)), int(param)))
        if kind is T.RadiusTopology:
            return self._with(T.RadiusTopology(T.build_radius_neighbors(ctx, pts, param), param))
        raise TypeError("set_topology expects KNNTopology or RadiusTopology")

    def _topo_line(self) -> str:
        t = self.topology
        return t.show().splitlines()[0] + "".join(" " + l.strip("├─└ ") for l in t.show().splitlines()[1:-1]) if hasattr(t, "show") else repr(t)

    def __repr__(self):
        """show(io, MIME"text/plain", x): a short tree naming the container, its size and its topology
        (src/cloud.jl:239-272, src/surface.jl, src/volume.jl)."""
        return f"{type(self).__name__}\n├─{len(self)} points\n└─topology: {self._topo_line()}\n"

    def rebuild_topology(self, ctx=None):
        """In place, same parameters; no-op for NoTopology (src/cloud.jl:224-228)."""
        if isinstance(self.topology, T.NoTopology):
            return None
        T.rebuild_topology(ctx or default_context(), self.topology, self.points())
        return None


class PointSurface(_HasTopology):
    def __init__(self, points, normals=None, areas=None, topology=None):
        """PointSurface(points[, normals, areas]) — src/surface.jl: one (point, normal, area)
        element per boundary point."""
        if isinstance(normals, T.AbstractTopology) and areas is None and topology is None:
            normals, topology = None, normals  # PointSurface(points, topology)
        self._points = _as_points(points)
        self.normals = None if normals is None else _as_points(normals, self._points.dtype)
        self.areas = None if areas is None else np.ascontiguousarray(areas, dtype=self._points.dtype).reshape(-1)
        if self.normals is not None and self.normals.shape != self._points.shape:
            raise ValueError("normals must match points")
        if self.areas is not None and self.areas.shape != (len(self._points),):
            raise ValueError("areas must be one value per point")
        self.topology = topology or T.NoTopology()

    def points(self):
        return self._points

    def __len__(self):
        return len(self._points)

    def _with(self, topo):
        return PointSurface(self._points, self.normals, self.areas, topo)


class PointVolume(_HasTopology):
    def __init__(self, points=None, dim: int = 3, dtype=np.float64, topology=None):
        self._points = _as_points(points) if points is not None else np.zeros((0, dim), dtype=dtype)
        self.topology = topology or T.NoTopology()

    def points(self):
        return self._points

    def __len__(self):
        return len(self._points)

    def _with(self, topo):
        return PointVolume(self._points, topology=topo)


class PointBoundary:
    """Named surfaces; points(boundary) concatenates them in insertion order (src/boundary.jl:164)."""

    def __init__(self, points=None, normals=None, areas=None, name: str = "surface1", surfaces=None):
        self.surfaces = OrderedDict()
        if isinstance(normals, str) and areas is None:
            normals, name = None, normals  # PointBoundary(points, name)
        if surfaces is not None:
            for k, v in surfaces.items():
                self.surfaces[k] = v if isinstance(v, PointSurface) else PointSurface(v)
        elif points is not None:
            self.surfaces[name] = points if isinstance(points, PointSurface) else PointSurface(points, normals, areas)

    @classmethod
    def from_stl(cls, path: str, dtype=np.float32, name: str = "surface1"):
        """PointBoundary(filepath, unit) (src/io.jl:36-56): one element per face."""
        from . import stl

        return cls(PointSurface(*stl.surface_elements(path, dtype)), name=name)

    def elements(self):
        """(points, normals, areas) over all surfaces, or None if a surface lacks normals/areas."""
        ss = list(self.surfaces.values())
        if not ss or any(s.normals is None or s.areas is None for s in ss):
            return None
        return (np.concatenate([s.points() for s in ss]), np.concatenate([s.normals for s in ss]),
                np.concatenate([s.areas for s in ss]))

    def points(self):
        parts = [s.points() for s in self.surfaces.values()]
        return np.concatenate(parts, axis=0) if parts else np.zeros((0, 3))

    def __len__(self):
        return sum(len(s) for s in self.surfaces.values())

    def __getitem__(self, name):
        return self.surfaces[name]


class PointCloud(_HasTopology):
    def __init__(self, boundary, volume=None, topology=None):
        self.boundary = boundary if isinstance(boundary, PointBoundary) else PointBoundary(boundary)
        bp = self.boundary.points()
        self.volume = volume if isinstance(volume, PointVolume) else PointVolume(
            volume, dim=bp.shape[1], dtype=bp.dtype)
        self.topology = topology or T.NoTopology()

    def points(self):
        """vcat(points(boundary), points(volume)): fresh array, boundary first (src/cloud.jl:235-237)."""
        bp, vp = self.boundary.points(), self.volume.points()
        if len(vp) == 0:
            return np.array(bp, copy=True)
        dt = np.promote_types(bp.dtype, vp.dtype)
        return np.concatenate([bp.astype(dt, copy=False), vp.astype(dt, copy=False)], axis=0)

    def __len__(self):
        return len(self.boundary) + len(self.volume)

    def _with(self, topo):
        return PointCloud(self.boundary, self.volume, topo)

    def __setitem__(self, name, surf):
        """cloud[:name] = surf rebuilds the topology implicitly (src/cloud.jl:86-90)."""
        self.boundary.surfaces[name] = surf if isinstance(surf, PointSurface) else PointSurface(surf)
        self.rebuild_topology()


# free-function spellings of the reference verbs
def points(x):
    return x.points()


def set_topology(x, kind, param, ctx=None):
    return x.set_topology(kind, param, ctx=ctx)


def rebuild_topology(x, ctx=None):
    return x.rebuild_topology(ctx=ctx)


def hastopology(x) -> bool:
    return x.hastopology()


def neighbors(x, i=None):
    return x.neighbors(i) if isinstance(x, _HasTopology) else T.neighbors(x, i)


# KNearestSearch / search / searchdists wrappers (src/neighbors.jl:1-21)
class KNearestSearch:
    def __init__(self, cloud, k: int):
        self.points = cloud.points() if hasattr(cloud, "points") else _as_points(cloud)
        self.k = int(k)


def search(cloud, method: KNearestSearch, ctx=None):
    """Per point the k nearest INCLUDING itself, ascending (self first: test/neighbors.jl:54-56)."""
    return (ctx or default_context()).knn(method.points, method.k, include_self=True)


def searchdists(cloud, method: KNearestSearch, ctx=None):
    return (ctx or default_context()).knn(method.points, method.k, include_self=True, return_dist=True)
