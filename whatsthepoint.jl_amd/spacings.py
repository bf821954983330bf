"""Spacing callables (src/discretization/spacings.jl).  `ConstantSpacing` is a scalar inside the
sweep; `LogLike` and `BoundaryLayerSpacing` are evaluated by the library on the GPU — inside the
sweep at every point's current position (include/wtp.h wtp_spacing_desc kinds 2 and 3) and, when
called on points here, through wtp_spacing_eval: the distance to the nearest boundary point is a
kd-tree 1-NN query on the device (csrc/wtp_spacing.hip).  Any other callable is evaluated by the
caller on the host and handed over as a per-point array."""
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


def _eval(law, pts, ctx=None):
    from .engine import default_context

    pts = np.asarray(pts)
    single = pts.ndim == 1
    out = (ctx or default_context()).spacing_eval(law.desc(), np.atleast_2d(pts))
    return out[0] if single else out


class LogLike(AbstractSpacing):  # spacings.jl:49-72
    def __init__(self, boundary_points, base_size: float, growth_rate: float):
        if len(boundary_points) == 0:
            raise ValueError("boundary_points must be non-empty")
        self.boundary = np.asarray(boundary_points)
        self.base_size, self.growth_rate = float(base_size), float(growth_rate)

    def desc(self):
        return dict(kind=2, p0=self.base_size, p1=self.growth_rate, p2=0.0, boundary=self.boundary)

    def __call__(self, pts, ctx=None):
        return _eval(self, pts, ctx)


class BoundaryLayerSpacing(AbstractSpacing):  # spacings.jl:93-133
    def __init__(self, boundary_points, at_wall: float, bulk: float, layer_thickness: float):
        if len(boundary_points) == 0:
            raise ValueError("boundary_points must be non-empty")
        if not layer_thickness > 0:
            raise ValueError(f"layer_thickness must be positive, got {layer_thickness}")
        self.boundary = np.asarray(boundary_points)
        self.at_wall, self.bulk, self.layer_thickness = float(at_wall), float(bulk), float(layer_thickness)

    def desc(self):
        return dict(kind=3, p0=self.at_wall, p1=self.bulk, p2=self.layer_thickness, boundary=self.boundary)

    def __call__(self, pts, ctx=None):
        return _eval(self, pts, ctx)
