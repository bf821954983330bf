"""Spacing callables (src/discretization/spacings.jl).  Inside the sweep the device uses a
constant or a per-point array (include/wtp.h wtp_spacing_desc); the variable laws are evaluated
on the host against the boundary points and handed over as that array (their device-side
evaluation is SURVEY.md §8f item 3)."""
from __future__ import annotations

import numpy as np


class AbstractSpacing:
    def __call__(self, pts):
        raise NotImplementedError


class ConstantSpacing(AbstractSpacing):  # spacings.jl:35-39
    def __init__(self, dx: float):
        self.dx = float(dx)

    def __call__(self, pts=None):
        if pts is None:
            return self.dx
        pts = np.asarray(pts)
        return self.dx if pts.ndim == 1 else np.full(len(pts), self.dx, dtype=pts.dtype)


def _min_distance(pts, boundary):  # spacings.jl:19-23 (1-NN to the boundary points)
    from scipy.spatial import cKDTree

    d, _ = cKDTree(np.asarray(boundary, dtype=np.float64)).query(np.asarray(pts, dtype=np.float64), k=1)
    return d


class LogLike(AbstractSpacing):  # spacings.jl:49-72
    def __init__(self, boundary_points, base_size: float, growth_rate: float):
        if len(boundary_points) == 0:
            raise ValueError("boundary_points must be non-empty")
        self.boundary = np.asarray(boundary_points)
        self.base_size, self.growth_rate = float(base_size), float(growth_rate)

    def __call__(self, pts):
        pts = np.atleast_2d(pts)
        x = _min_distance(pts, self.boundary)
        a = self.base_size * (1 - (self.growth_rate - 1))
        return (self.base_size * x / (a + x)).astype(pts.dtype)


class BoundaryLayerSpacing(AbstractSpacing):  # spacings.jl:93-133
    def __init__(self, boundary_points, at_wall: float, bulk: float, layer_thickness: float):
        if len(boundary_points) == 0:
            raise ValueError("boundary_points must be non-empty")
        if not layer_thickness > 0:
            raise ValueError(f"layer_thickness must be positive, got {layer_thickness}")
        self.boundary = np.asarray(boundary_points)
        self.at_wall, self.bulk, self.layer_thickness = float(at_wall), float(bulk), float(layer_thickness)

    def __call__(self, pts):
        pts = np.atleast_2d(pts)
        d = _min_distance(pts, self.boundary)
        center, width = self.layer_thickness / 2, self.layer_thickness / 6
        sig = 1.0 / (1.0 + np.exp(-(d - center) / width))
        return (self.at_wall + (self.bulk - self.at_wall) * sig).astype(pts.dtype)
