"""Topology types of the reference (src/topology.jl:7-70,131-148).  Neighbour storage is an
(n, k) int32 matrix (KNN) or a CSR pair (radius) as libwtp returns them — 0-based indices;
`neighbors(t, i)` gives row i like `t.neighbors[i]` does in Julia (1-based there)."""
from __future__ import annotations

import numpy as np

from ._lib import WtpArgumentError


class AbstractTopology:
    pass


class NoTopology(AbstractTopology):
    def __repr__(self):
        return "NoTopology()"

    def show(self) -> str:
        return "NoTopology\n"


class KNNTopology(AbstractTopology):
    def __init__(self, neighbors, k: int):
        self.neighbors = neighbors
        self.k = int(k)

    def __repr__(self):
        return f"KNNTopology(k={self.k})"

    def show(self) -> str:
        return f"KNNTopology\n├─k: {self.k}\n└─points: {len(self.neighbors)}\n"


class CSR:
    """Ragged adjacency: row i = idx[offsets[i]:offsets[i+1]]."""

    def __init__(self, offsets, idx):
        self.offsets = offsets
        self.idx = idx

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, i):
        return self.idx[self.offsets[i]:self.offsets[i + 1]]


class RadiusTopology(AbstractTopology):
    def __init__(self, neighbors, radius):
        self.neighbors = neighbors
        self.radius = radius

    def __repr__(self):
        return f"RadiusTopology(r={self.radius})"

    def show(self) -> str:
        return f"RadiusTopology\n├─radius: {self.radius}\n└─points: {len(self.neighbors)}\n"


def neighbors(t, i=None):
    if isinstance(t, NoTopology):
        raise WtpArgumentError("NoTopology has no neighbors")  # src/topology.jl:61-62
    return t.neighbors if i is None else t.neighbors[i]


def isvalid(t) -> bool:
    return True  # src/topology.jl:69-70


def _get_radius(radius, points):
    return radius(points) if callable(radius) else radius  # src/topology.jl:99-100


def build_knn_neighbors(ctx, points, k: int):
    """_build_knn_neighbors (src/topology.jl:79-84): k+1 query, self dropped."""
    return ctx.knn(points, k, include_self=False)


def build_radius_neighbors(ctx, points, radius):
    """_build_radius_neighbors (src/topology.jl:91-97)."""
    r = float(_get_radius(radius, points))
    off, idx = ctx.radius(points, r)
    return CSR(off, idx)


def rebuild_topology(ctx, topo, points):
    """rebuild_topology! (src/topology.jl:109-129): in place, stored k / radius kept."""
    if isinstance(topo, NoTopology):
        return None
    if isinstance(topo, KNNTopology):
        topo.neighbors = build_knn_neighbors(ctx, points, topo.k)
    elif isinstance(topo, RadiusTopology):
        topo.neighbors = build_radius_neighbors(ctx, points, topo.radius)
    return None
