"""`TriangleOctree` — host mirror of the reference's triangle-mesh geometry index
(src/octree/triangle_octree.jl:122-126,419-486) as far as `repel(cloud, spacing, octree)` and its
tests use it: `isinside`, signed distance, nearest triangle and `project_to_boundary`.

The reference subdivides an octree over the triangles and caches a per-leaf inside/outside class.
Here the mesh goes to the GPU once (csrc/wtp_mesh.hip: face normals, angle-weighted pseudonormals,
a bounding-volume tree in heap order) and every query is a device kernel; there is no host search
structure and no CPU path.  The constructor keeps the reference's two orientation guards."""
from __future__ import annotations

import numpy as np

from ._lib import WtpArgumentError
from .engine import default_context

_DEGENERATE_EPS = 1.0e-10  # src/octree/triangle_octree.jl:141


def _weld(tri_soup):
    """(nt, 3, 3) triangle soup -> (vertices, triangles) with exactly equal corners shared."""
    flat = np.ascontiguousarray(tri_soup).reshape(-1, 3)
    verts, inv = np.unique(flat, axis=0, return_inverse=True)
    return verts, inv.reshape(-1, 3).astype(np.int32)


def has_consistent_normals(vertices, triangles) -> bool:
    """No edge is traversed twice in the same direction (src/octree/triangle_octree.jl:338-367)."""
    v = np.asarray(vertices)
    t = np.asarray(triangles)
    if len(t) <= 1:
        return True
    # compare by coordinates, as the reference does: identical corners of a soup count as one vertex
    _, canon = np.unique(v, axis=0, return_inverse=True)
    c = canon.reshape(-1)[t]
    a = np.concatenate([c[:, 0], c[:, 1], c[:, 2]])
    b = np.concatenate([c[:, 1], c[:, 2], c[:, 0]])
    directed = a.astype(np.int64) * (int(c.max()) + 1) + b
    return len(np.unique(directed)) == len(directed)


def signed_volume(vertices, triangles) -> float:
    """Σ v1 · (v2 × v3) / 6 (src/octree/triangle_octree.jl:378-385)."""
    v = np.asarray(vertices, dtype=np.float64)
    t = np.asarray(triangles)
    return float(np.einsum("ij,ij->i", v[t[:, 0]], np.cross(v[t[:, 1]], v[t[:, 2]])).sum() / 6.0)


class TriangleOctree:
    def __init__(self, vertices, triangles, *, classify_leaves: bool = True, verify_orientation: bool = True,
                 ctx=None):
        v = np.ascontiguousarray(vertices)
        if v.dtype not in (np.float32, np.float64):
            v = v.astype(np.float64)
        t = np.ascontiguousarray(triangles, dtype=np.int32)
        if v.ndim != 2 or v.shape[1] != 3 or t.ndim != 2 or t.shape[1] != 3:
            raise WtpArgumentError("TriangleOctree requires a pure-triangle mesh: (nv, 3) vertices, (nt, 3) triangles")
        if verify_orientation and not has_consistent_normals(v, t):
            raise WtpArgumentError("Triangle mesh has orientation errors (flipped faces). To skip this check, "
                                   "pass verify_orientation=False.")
        if verify_orientation and classify_leaves:
            lo, hi = v.min(axis=0), v.max(axis=0)
            floor = -_DEGENERATE_EPS * float(np.linalg.norm((hi - lo).astype(np.float64))) ** 3
            if signed_volume(v, t) < floor:
                raise WtpArgumentError("Triangle mesh is inside-out (negative signed volume): flip the triangle "
                                       "winding. To skip this check, pass verify_orientation=False.")
        self.vertices, self.triangles = v, t
        self.dtype = v.dtype
        self._ctx = ctx
        self._token = object()
        lo, hi = v.min(axis=0), v.max(axis=0)
        self.bbox_min, self.bbox_max = lo, hi

    @classmethod
    def from_stl(cls, path: str, dtype=np.float64, **kw):
        """import_mesh(path) |> TriangleOctree: binary STL, corners welded by exact coordinates."""
        from .stl import read_binary_stl

        verts, tris = _weld(read_binary_stl(path).astype(dtype))
        return cls(verts, tris, **kw)

    # the mesh lives in a context; another octree (or context) re-uploads on first use
    def _resident(self, ctx=None):
        ctx = ctx or self._ctx or default_context()
        if getattr(ctx, "_mesh_owner", None) is not self._token:
            ctx.mesh_set(self.vertices, self.triangles)
            ctx._mesh_owner = self._token
        return ctx

    def __len__(self):
        return len(self.triangles)

    num_triangles = property(lambda self: len(self.triangles))

    def face_normals(self, ctx=None):
        return self._resident(ctx).mesh_face_normals()

    def query(self, pts, offset: float = 0.0, want=("sd", "tri", "closest", "inside", "projected"), ctx=None):
        pts = np.asarray(pts)
        single = pts.ndim == 1
        out = self._resident(ctx).mesh_query(np.atleast_2d(pts), offset, want)
        return {k: v[0] for k, v in out.items()} if single else out

    def isinside(self, pts, ctx=None):
        """isinside(point(s), octree) (src/octree/triangle_octree.jl:97-116)."""
        return self.query(pts, want=("inside",), ctx=ctx)["inside"]

    def signed_distance(self, pts, ctx=None):
        """_compute_signed_distance_octree (:583-607): negative inside."""
        return self.query(pts, want=("sd",), ctx=ctx)["sd"]

    def nearest_triangle(self, pts, ctx=None):
        q = self.query(pts, want=("tri", "closest"), ctx=ctx)
        return q["tri"], q["closest"]

    def project_to_boundary(self, pts, offset: float = 0.0, ctx=None):
        """_project_to_boundary (src/repel.jl:522-537): (points on the mesh nudged inward, triangle ids)."""
        q = self.query(pts, offset, want=("projected", "tri"), ctx=ctx)
        return q["projected"], q["tri"]


def isinside_octree(pts, octree: TriangleOctree, ctx=None):
    return octree.isinside(pts, ctx=ctx)
