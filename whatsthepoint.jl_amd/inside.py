"""`isinside` — host mirror of src/isinside.jl for whole arrays of test points.

The reference tests one point per call and `repel` filters its result with
`filter(x -> isinside(x, cloud), p)` (src/repel.jl:90): O(N·M) work that dominates once the sweep
is fast.  Here the pair sums run on the GPU (csrc/wtp_inside.hip); dispatch,
polygon validation and error behaviour stay on the host and follow the reference:

  2-D  points / PointSurface / PointCloud      winding number over the ordered boundary points
  3-D  PointCloud / PointBoundary              Green's-function sum over (centroid, normal, area)
  3-D  PointSurface                            TypeError (the reference has no such method)
"""
from __future__ import annotations

import numpy as np

from ._lib import WtpArgumentError
from .cloud import PointBoundary, PointCloud, PointSurface
from .engine import default_context


def validate_polygon_ordering(pts) -> None:
    """_validate_polygon_ordering (src/isinside.jl:36-69): at least 3 points; a signed area below
    1e-10 of the bounding box means unordered (self-intersecting) or collinear points."""
    pts = np.asarray(pts)
    n = len(pts)
    if n < 3:
        raise WtpArgumentError(f"need at least 3 points to define a polygon, got {n}")
    x, y = pts[:, 0].astype(pts.dtype), pts[:, 1].astype(pts.dtype)
    xn, yn = np.roll(x, -1), np.roll(y, -1)
    sa = pts.dtype.type(0)
    for i in range(n):  # the reference accumulates sequentially in T
        sa = sa + (x[i] * yn[i] - xn[i] * y[i])
    sa = sa / 2
    bbox_area = (x.max() - x.min()) * (y.max() - y.min())
    if bbox_area > 0 and abs(sa) < pts.dtype.type(1.0e-10) * bbox_area:
        raise WtpArgumentError(
            "polygon points do not appear to be ordered sequentially around the boundary; "
            "the 2D isinside winding number algorithm requires points ordered in a loop "
            "(clockwise or counter-clockwise)")


def _test_array(testpoints, dim):
    t = np.asarray(testpoints)
    single = t.ndim == 1
    t = np.atleast_2d(t)
    if t.shape[1] != dim:
        raise WtpArgumentError(f"test points are {t.shape[1]}-D, the boundary is {dim}-D")
    if t.dtype not in (np.float32, np.float64):
        t = t.astype(np.float64)
    return np.ascontiguousarray(t), single


def isinside(testpoints, obj, ctx=None):
    """bool (one point) or bool[n] (an (n, dim) array): is the point inside the closed domain?"""
    ctx = ctx or default_context()
    if isinstance(obj, PointCloud):
        obj = obj.boundary
    if isinstance(obj, PointBoundary):
        dim = obj.points().shape[1]
        if dim == 2:
            poly = obj.points()
        else:
            el = obj.elements()
            if el is None:
                raise WtpArgumentError("3-D isinside needs boundary normals and areas "
                                       "(PointSurface(points, normals, areas) / PointBoundary.from_stl)")
            t, single = _test_array(testpoints, 3)
            out = ctx.isinside_greens(t, *(e.astype(t.dtype, copy=False) for e in el))
            return bool(out[0]) if single else out
    elif isinstance(obj, PointSurface):
        if obj.points().shape[1] != 2:
            raise TypeError("isinside(point, ::PointSurface) exists in 2-D only (test/isinside.jl:75-80): "
                            "pass the PointBoundary or PointCloud")
        poly = obj.points()
    else:
        poly = np.asarray(obj)
        if poly.ndim != 2 or poly.shape[1] != 2:
            raise TypeError("isinside expects 2-D polygon points, a PointSurface, PointBoundary or PointCloud")
    validate_polygon_ordering(poly)
    t, single = _test_array(testpoints, 2)
    out = ctx.isinside_winding(t, poly.astype(t.dtype, copy=False))
    return bool(out[0]) if single else out
