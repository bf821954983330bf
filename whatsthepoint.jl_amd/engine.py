"""Thin host wrappers over the C ABI (include/wtp.h): one `Context` per GPU.

Everything here only marshals numpy arrays to libwtp; the algorithms live in csrc/.
There is deliberately no CPU implementation behind these calls.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float32:
        return L.WTP_F32
    if dt == np.float64:
        return L.WTP_F64
    raise L.WtpArgumentError(f"coordinates must be float32 or float64, got {dt}")


def _cloud(xyz):
    xyz = np.asarray(xyz)
    if xyz.ndim != 2 or xyz.shape[1] not in (2, 3):
        raise L.WtpArgumentError("coordinates must have shape (n, 2) or (n, 3)")
    if xyz.dtype not in (np.float32, np.float64):
        xyz = xyz.astype(np.float64)
    return np.ascontiguousarray(xyz)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """Owns the device buffers, stream and timers of one GPU (wtp_create / wtp_destroy)."""

    def __init__(self, device: int = 0):
        self._lib = L.load()
        h = C.c_void_p()
        dev = (C.c_int * 1)(device)
        rc = self._lib.wtp_create(dev, 1, C.byref(h))
        if rc != L.WTP_OK:
            L.check(None, rc)
        self._h = h
        self.device = device
        self._mesh_key = None   # identity of the mesh resident in this context (octree.py)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.wtp_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- KNNTopology / KNearestSearch (src/topology.jl:79-84, src/neighbors.jl) ---------
    def knn(self, xyz, k: int, include_self: bool = False, return_dist: bool = False):
        xyz = _cloud(xyz)
        n, dim = xyz.shape
        if k < 1 or k > n - (0 if include_self else 1):
            raise L.WtpArgumentError(f"k={k} needs k+1 <= n={n} points")
        idx = np.empty((n, k), dtype=np.int32)
        dist = np.empty((n, k), dtype=xyz.dtype) if return_dist else None
        rc = self._lib.wtp_knn(self._h, _vp(xyz), n, dim, _dtype_code(xyz.dtype), int(k), int(bool(include_self)),
                               _vp(idx), _vp(dist))
        L.check(self._h, rc)
        return (idx, dist) if return_dist else idx

    def knn_dev(self, d_xyz_ptr: int, n: int, dim: int, dtype, k: int, include_self: bool, d_idx_ptr: int,
                d_dist_ptr: int = 0):
        rc = self._lib.wtp_knn_dev(self._h, C.c_void_p(d_xyz_ptr), n, dim, _dtype_code(dtype), int(k),
                                   int(bool(include_self)), C.c_void_p(d_idx_ptr),
                                   C.c_void_p(d_dist_ptr) if d_dist_ptr else None)
        L.check(self._h, rc)

    # ---- RadiusTopology / BallSearch (src/topology.jl:91-97) -----------------------------
    def radius(self, xyz, r: float):
        """CSR stencils: (offsets int64[n+1], idx int32[nnz]); rows ascending (d2, index)."""
        xyz = _cloud(xyz)
        n, dim = xyz.shape
        # row lengths are scanned on the device; the offsets come back once and stay resident for the fill
        offsets = np.empty(n + 1, dtype=np.int64)
        rc = self._lib.wtp_radius_offsets(self._h, _vp(xyz), n, dim, _dtype_code(xyz.dtype), float(r), _vp(offsets))
        L.check(self._h, rc)
        idx = np.empty(max(int(offsets[-1]), 1), dtype=np.int32)
        rc = self._lib.wtp_radius_fill(self._h, None, _vp(idx))
        L.check(self._h, rc)
        return offsets, idx[: int(offsets[-1])]

    def radius_two_phase(self, xyz, r: float):
        """The caller-side scan of the C ABI's first form (wtp_radius_count -> offsets -> wtp_radius_fill)."""
        xyz = _cloud(xyz)
        n, dim = xyz.shape
        counts = np.empty(n, dtype=np.int32)
        rc = self._lib.wtp_radius_count(self._h, _vp(xyz), n, dim, _dtype_code(xyz.dtype), float(r), _vp(counts))
        L.check(self._h, rc)
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        idx = np.empty(max(int(offsets[-1]), 1), dtype=np.int32)
        rc = self._lib.wtp_radius_fill(self._h, _vp(offsets), _vp(idx))
        L.check(self._h, rc)
        return offsets, idx[: int(offsets[-1])]

    # ---- repel -------------------------------------------------------------------------------
    def relax(self, snap, n_fixed: int, spacing, force, k: int, alpha_lo: float, alpha_max: float, device_ptr=None):
        return RelaxSession(self, snap, n_fixed, spacing, force, k, alpha_lo, alpha_max, device_ptr=device_ptr)

    # ---- variable spacing laws at arbitrary points (spacing.(points)) ---------------------------
    def spacing_eval(self, law: dict, xyz):
        xyz = _cloud(xyz)
        sd, keep = _law_desc(law, xyz.dtype, xyz.shape[1])
        out = np.empty(len(xyz), dtype=xyz.dtype)
        rc = self._lib.wtp_spacing_eval(self._h, C.byref(sd), _vp(xyz), len(xyz), xyz.shape[1], _dtype_code(xyz.dtype),
                                        _vp(out))
        L.check(self._h, rc)
        del keep
        return out

    # ---- isinside post-filter (src/repel.jl:90; src/isinside.jl) -------------------------------
    def isinside_greens(self, test, elem_xyz, elem_normal, elem_area, return_g: bool = False):
        """3-D: bool[n] (and g[n]) for test points against boundary elements (centroid, normal, area)."""
        test = _cloud(test)
        if test.shape[1] != 3:
            raise L.WtpArgumentError("the Green's-function test is 3-D")
        dt = test.dtype
        p = np.ascontiguousarray(elem_xyz, dtype=dt)
        nrm = np.ascontiguousarray(elem_normal, dtype=dt)
        a = np.ascontiguousarray(elem_area, dtype=dt).reshape(-1)
        if p.ndim != 2 or p.shape[1] != 3 or nrm.shape != p.shape or a.shape != (len(p),):
            raise L.WtpArgumentError("boundary elements need (m, 3) points, (m, 3) normals and m areas")
        inside = np.zeros(len(test), dtype=np.uint8)
        g = np.zeros(len(test), dtype=dt) if return_g else None
        rc = self._lib.wtp_isinside_greens(self._h, _vp(test), len(test), _vp(p), _vp(nrm), _vp(a), len(p),
                                           _dtype_code(dt), _vp(inside), _vp(g))
        L.check(self._h, rc)
        return (inside.astype(bool), g) if return_g else inside.astype(bool)

    def isinside_winding(self, test, poly, return_sum: bool = False):
        """2-D: bool[n] (and angle sums) for test points against an ordered, closed polygon."""
        test = _cloud(test)
        if test.shape[1] != 2:
            raise L.WtpArgumentError("the winding-number test is 2-D")
        dt = test.dtype
        poly = np.ascontiguousarray(poly, dtype=dt)
        if poly.ndim != 2 or poly.shape[1] != 2:
            raise L.WtpArgumentError("polygon must be (m, 2)")
        inside = np.zeros(len(test), dtype=np.uint8)
        s = np.zeros(len(test), dtype=dt) if return_sum else None
        rc = self._lib.wtp_isinside_winding(self._h, _vp(test), len(test), _vp(poly), len(poly), _dtype_code(dt),
                                            _vp(inside), _vp(s))
        L.check(self._h, rc)
        return (inside.astype(bool), s) if return_sum else inside.astype(bool)

    # ---- consumers of the rows -------------------------------------------------
    def pca_normals(self, xyz, k: int = 5):
        """compute_normals (src/normals.jl:15-46): (n, dim) unit normals, largest component positive."""
        xyz = _cloud(xyz)
        out = np.empty_like(xyz)
        rc = self._lib.wtp_pca_normals(self._h, _vp(xyz), len(xyz), xyz.shape[1], _dtype_code(xyz.dtype), int(k), _vp(out))
        L.check(self._h, rc)
        return out

    def gradient_limit(self, centers, h0, g: float, k: int = 12, tol: float = 1.0e-3, max_sweeps: int = 2000):
        """_gradient_limit_field on leaf centres (octree.jl:677-717): (limited field, sweeps applied)."""
        c = _cloud(centers)
        h0 = np.ascontiguousarray(h0, dtype=c.dtype).reshape(-1)
        if h0.shape != (len(c),):
            raise L.WtpArgumentError("h0 needs one value per centre")
        out = np.empty_like(h0)
        sw = C.c_int(0)
        rc = self._lib.wtp_gradient_limit(self._h, _vp(c), len(c), c.shape[1], _dtype_code(c.dtype), int(min(k, len(c))),
                                          _vp(h0), float(g), float(tol), int(max_sweeps), _vp(out), C.byref(sw))
        L.check(self._h, rc)
        return out, sw.value

    # ---- triangle-mesh geometry index (octree method of repel; src/octree/triangle_octree.jl) -----
    def mesh_set(self, vertices, triangles):
        """vertices (nv, 3) float32/float64 = the index's machine type; triangles (nt, 3) 0-based."""
        v = np.ascontiguousarray(vertices)
        if v.dtype not in (np.float32, np.float64):
            v = v.astype(np.float64)
        t = np.ascontiguousarray(triangles, dtype=np.int32)
        if v.ndim != 2 or v.shape[1] != 3 or t.ndim != 2 or t.shape[1] != 3:
            raise L.WtpArgumentError("mesh needs (nv, 3) vertices and (nt, 3) triangles")
        L.check(self._h, self._lib.wtp_mesh_set(self._h, _vp(v), len(v), _vp(t), len(t), _dtype_code(v.dtype)))
        self._mesh_key = (len(v), len(t), v.dtype)
        self._mesh_owner = None   # octree.py marks the TriangleOctree it uploaded

    def mesh_clear(self):
        L.check(self._h, self._lib.wtp_mesh_clear(self._h))
        self._mesh_key = None
        self._mesh_owner = None

    def mesh_face_normals(self):
        nt = self._mesh_key[1]
        out = np.empty((nt, 3), dtype=np.float64)
        L.check(self._h, self._lib.wtp_mesh_face_normals(self._h, _vp(out)))
        return out

    def mesh_bounds(self):
        out = np.empty(6, dtype=np.float64)
        L.check(self._h, self._lib.wtp_mesh_bounds(self._h, _vp(out)))
        return out[:3].copy(), out[3:].copy()

    def mesh_query(self, pts, offset: float = 0.0, want=("sd", "tri", "closest", "inside", "projected")):
        """Nearest-triangle queries for (n, 3) points; returns a dict with the requested outputs."""
        pts = _cloud(pts)
        if pts.shape[1] != 3:
            raise L.WtpArgumentError("mesh queries are 3-D")
        n, dt = len(pts), pts.dtype
        out = {}
        if "sd" in want:
            out["sd"] = np.empty(n, dtype=dt)
        if "tri" in want:
            out["tri"] = np.empty(n, dtype=np.int32)
        if "closest" in want:
            out["closest"] = np.empty((n, 3), dtype=dt)
        if "inside" in want:
            out["inside"] = np.zeros(n, dtype=np.uint8)
        if "projected" in want:
            out["projected"] = np.empty((n, 3), dtype=dt)
        rc = self._lib.wtp_mesh_query(self._h, _vp(pts), n, _dtype_code(dt), float(offset), _vp(out.get("sd")),
                                      _vp(out.get("tri")), _vp(out.get("closest")), _vp(out.get("inside")),
                                      _vp(out.get("projected")))
        L.check(self._h, rc)
        if "inside" in out:
            out["inside"] = out["inside"].astype(bool)
        return out

    def set_stream(self, stream_handle=None):
        """Run the library on a caller-owned HIP stream (handle as an int; 0 = the device's default
        stream, which is torch's current stream unless the caller switched); None = own stream."""
        ext = stream_handle is not None
        L.check(self._h, self._lib.wtp_set_stream(self._h, C.c_void_p(int(stream_handle or 0)), int(ext)))

    # ---- measurement ---------------------------------------------------------------------------
    # ---- the context's RCCL communicator (include/wtp.h: wtp_comm_*; one process per GPU) ----------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        L.check(self._h, self._lib.wtp_comm_unique_id(self._h, buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        L.check(self._h, self._lib.wtp_comm_init(self._h, buf, int(rank), int(nranks)))

    def comm_finalize(self):
        L.check(self._h, self._lib.wtp_comm_finalize(self._h))

    def comm_exchange_rows(self, peer_lo: int, peer_hi: int, send_lo_ptr: int, n_send_lo: int, send_hi_ptr: int,
                           n_send_hi: int, recv_lo_ptr: int, recv_hi_ptr: int, cap: int):
        """One round with the two neighbours along an axis (peer -1: none); device pointers to 16-byte rows.
        Returns the received row counts (lo, hi); the rows are ordered on the context's stream."""
        n_lo, n_hi = C.c_int64(0), C.c_int64(0)
        L.check(self._h, self._lib.wtp_comm_exchange_rows(
            self._h, int(peer_lo), int(peer_hi), C.c_void_p(send_lo_ptr or None), int(n_send_lo),
            C.c_void_p(send_hi_ptr or None), int(n_send_hi), C.c_void_p(recv_lo_ptr or None), C.c_void_p(recv_hi_ptr or None),
            int(cap), C.byref(n_lo), C.byref(n_hi)))
        return int(n_lo.value), int(n_hi.value)

    def comm_allreduce_stats(self, st: dict) -> dict:
        s = L.StepStats()
        for k, v in st.items():
            if hasattr(s, k):
                setattr(s, k, v)
        L.check(self._h, self._lib.wtp_comm_allreduce_stats(self._h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in L.StepStats._fields_}

    def timers(self):
        out = (C.c_double * 4)()
        L.check(self._h, self._lib.wtp_timers_get(self._h, out))
        return dict(hash_ms=out[0], sweep_ms=out[1], other_ms=out[2], sweep_launches=int(out[3]))

    def timers_reset(self):
        L.check(self._h, self._lib.wtp_timers_reset(self._h))

    def gen_uniform_dev(self, seed: int, first: int, n: int, dim: int, dtype, d_out_ptr: int):
        rc = self._lib.wtp_gen_uniform_dev(self._h, C.c_uint64(seed), first, n, dim, _dtype_code(dtype),
                                           C.c_void_p(d_out_ptr))
        L.check(self._h, rc)


def _law_desc(law: dict, dtype, dim: int):
    """wtp_spacing_desc for a device-evaluated law: dict(kind=2|3, p0, p1, p2, boundary=(m, dim) array).
    Returns (desc, keepalive)."""
    b = np.ascontiguousarray(law["boundary"], dtype=dtype)
    if b.ndim != 2 or b.shape[1] != dim:
        raise L.WtpArgumentError(f"the spacing law's boundary points must be (m, {dim})")
    sd = L.SpacingDesc()
    sd.kind, sd.constant, sd.per_point = int(law["kind"]), 0.0, None
    sd.p0, sd.p1, sd.p2 = float(law["p0"]), float(law["p1"]), float(law.get("p2", 0.0))
    sd.boundary_xyz, sd.n_boundary = b.ctypes.data, len(b)
    return sd, b


def _stats_dict(s: L.StepStats):
    return dict(max_force=s.max_force, sum_u=s.sum_u, sum_u2=s.sum_u2, n_move=s.n_move, argmin_i=s.argmin_i,
                argmin_j=s.argmin_j, argmin_r=s.argmin_r, n_fallback=s.n_fallback, n_uncovered=s.n_uncovered,
                n_escaped=s.n_escaped)


class RelaxSession:
    """Device-resident state of one `_relax!` call (src/repel.jl:202-339): coordinates stay in
    HBM between sweeps; the host sees three scalars per iteration."""

    def __init__(self, ctx: Context, snap, n_fixed: int, spacing, force, k: int, alpha_lo: float, alpha_max: float,
                 device_ptr=None):
        """snap: host (n, dim) array — or, with device_ptr=(ptr, n, dim, dtype), a snapshot that is
        already resident on the context's GPU (bench / sharded driver)."""
        self.ctx = ctx
        self._lib = ctx._lib
        if device_ptr is None:
            snap = _cloud(snap)
            self.n, self.dim = snap.shape
            self.dtype = snap.dtype
        else:
            _, self.n, self.dim, dt = device_ptr
            self.dtype = np.dtype(dt)
        self.n_fixed = int(n_fixed)
        sd = L.SpacingDesc()
        self._sp_keep = None
        if np.isscalar(spacing):
            sd.kind, sd.constant, sd.per_point = 0, float(spacing), None
        elif isinstance(spacing, dict):  # a law the library evaluates itself (LogLike / BoundaryLayerSpacing)
            sd, self._sp_keep = _law_desc(spacing, self.dtype, self.dim)
        else:
            sp = np.ascontiguousarray(spacing, dtype=self.dtype)
            if sp.shape != (self.n,):
                raise L.WtpArgumentError("per-point spacing needs one value per snapshot point")
            self._sp_keep = sp
            sd.kind, sd.constant, sd.per_point = 1, 0.0, sp.ctypes.data
        fd = L.ForceDesc(int(force["kind"]), float(force["beta"]), float(force.get("u0", 1.0)),
                         float(force.get("gamma", 3.0)))
        if device_ptr is None:
            rc = self._lib.wtp_relax_init(ctx._h, _vp(snap), self.n, self.n_fixed, self.dim, _dtype_code(self.dtype),
                                          C.byref(sd), C.byref(fd), int(k), float(alpha_lo), float(alpha_max))
        else:
            rc = self._lib.wtp_relax_init_dev(ctx._h, C.c_void_p(device_ptr[0]), self.n, self.n_fixed, self.dim,
                                              _dtype_code(self.dtype), C.byref(sd), C.byref(fd), int(k),
                                              float(alpha_lo), float(alpha_max))
        L.check(ctx._h, rc)
        self._open = True

    def step(self, rebuild: bool = True):
        st = L.StepStats()
        L.check(self.ctx._h, self._lib.wtp_relax_step(self.ctx._h, int(bool(rebuild)), C.byref(st)))
        return _stats_dict(st)

    def run(self, n_iters: int, rebuild_every: int = 1):
        conv = np.zeros(max(n_iters, 1), dtype=np.float64)
        st = L.StepStats()
        L.check(self.ctx._h, self._lib.wtp_relax_run(self.ctx._h, int(n_iters), int(rebuild_every), _vp(conv),
                                                     C.byref(st)))
        return conv[:n_iters], _stats_dict(st)

    def run_async_free(self, n_iters: int, rebuild_every: int = 1):
        """n_iters sweeps with no read-back at all (bench inner loop)."""
        L.check(self.ctx._h, self._lib.wtp_relax_run(self.ctx._h, int(n_iters), int(rebuild_every), None, None))

    def run_until(self, max_iters: int, rebuild_every: int = 1, tol: float = 1e-6, stall_after: int = 50,
                  cv_target: float = 0.0):
        """The loop with the reference's stop rules evaluated on the device (src/repel.jl:305-334): returns
        (conv, reason, last stats); reason 0 max_iters, 1 tol, 2 cv_target (positions reverted), 3 stall."""
        conv = np.zeros(max(max_iters, 1), dtype=np.float64)
        st = L.StepStats()
        n_done, reason = C.c_int(0), C.c_int(0)
        L.check(self.ctx._h, self._lib.wtp_relax_run_until(self.ctx._h, int(max_iters), int(rebuild_every), float(tol),
                                                           int(stall_after), float(cv_target), _vp(conv), C.byref(n_done),
                                                           C.byref(reason), C.byref(st)))
        return conv[: n_done.value], int(reason.value), _stats_dict(st)

    def positions(self):
        out = np.empty((self.n - self.n_fixed, self.dim), dtype=self.dtype)
        L.check(self.ctx._h, self._lib.wtp_relax_get(self.ctx._h, _vp(out)))
        return out

    def positions_dev(self, d_out_ptr: int):
        """Movable points into device memory ((n - n_fixed) x dim of dtype)."""
        L.check(self.ctx._h, self._lib.wtp_relax_get_dev(self.ctx._h, C.c_void_p(d_out_ptr)))

    def point_data(self):
        m = self.n - self.n_fixed
        forces = np.empty(m, dtype=self.dtype)
        nn_dist = np.empty(m, dtype=self.dtype)
        nn_id = np.empty(m, dtype=np.int32)
        L.check(self.ctx._h, self._lib.wtp_relax_get_point_data(self.ctx._h, _vp(forces), _vp(nn_dist), _vp(nn_id)))
        return dict(forces=forces, nn_dist=nn_dist, nn_id=nn_id)

    def set_point(self, i: int, xyz):
        v = np.ascontiguousarray(xyz, dtype=self.dtype).reshape(self.dim)
        L.check(self.ctx._h, self._lib.wtp_relax_set(self.ctx._h, int(i), _vp(v)))

    def set_points(self, idx, xyz):
        """Place many movable points at once (indices strictly increasing)."""
        idx = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1)
        v = np.ascontiguousarray(xyz, dtype=self.dtype).reshape(len(idx), self.dim)
        L.check(self.ctx._h, self._lib.wtp_relax_set_batch(self.ctx._h, _vp(idx), _vp(v), len(idx)))

    def revert(self):
        L.check(self.ctx._h, self._lib.wtp_relax_revert(self.ctx._h))

    def spacings(self):
        """The per-point spacings the session holds, snapshot order (src/repel.jl:209,251)."""
        out = np.empty(self.n, dtype=self.dtype)
        L.check(self.ctx._h, self._lib.wtp_relax_get_spacing(self.ctx._h, _vp(out)))
        return out

    def set_spacing(self, spacing):
        sp = np.ascontiguousarray(spacing, dtype=self.dtype)
        if sp.shape != (self.n,):
            raise L.WtpArgumentError("per-point spacing needs one value per snapshot point")
        L.check(self.ctx._h, self._lib.wtp_relax_set_spacing(self.ctx._h, _vp(sp)))

    # ---- wall rule of the octree method (src/repel.jl:448-469) ----------------------------------
    def set_wall(self, n_boundary: int, offset_dist: float):
        """After every sweep: boundary points are re-projected onto the context's mesh, volume points
        that left the domain return to their previous position (stats['n_escaped'])."""
        L.check(self.ctx._h, self._lib.wtp_relax_set_wall(self.ctx._h, int(n_boundary), float(offset_dist)))

    def get_wall(self, clear_escaped: bool = False):
        m = self.n - self.n_fixed
        tri = np.empty(m, dtype=np.int32)
        is_bnd = np.empty(m, dtype=np.uint8)
        esc = np.empty(m, dtype=np.uint8)
        L.check(self.ctx._h, self._lib.wtp_relax_get_wall(self.ctx._h, _vp(tri), _vp(is_bnd), _vp(esc),
                                                          int(bool(clear_escaped))))
        return dict(tri=tri, is_bnd=is_bnd.astype(bool), escaped=esc.astype(bool))

    def query_knn(self, xyz, k: int, return_dist: bool = False):
        """k nearest snapshot points (of the last rebuild) of arbitrary positions: (nq, k) indices."""
        q = np.ascontiguousarray(np.atleast_2d(xyz), dtype=self.dtype)
        if q.shape[1] != self.dim:
            raise L.WtpArgumentError("queries must have the session's dimension")
        idx = np.empty((len(q), int(k)), dtype=np.int32)
        dist = np.empty((len(q), int(k)), dtype=self.dtype) if return_dist else None
        L.check(self.ctx._h, self._lib.wtp_relax_query_knn(self.ctx._h, _vp(q), len(q), int(k), _vp(idx), _vp(dist)))
        return (idx, dist) if return_dist else idx

    def set_wall_flags(self, is_bnd, tri):
        m = self.n - self.n_fixed
        b = np.ascontiguousarray(is_bnd, dtype=np.uint8)
        t = np.ascontiguousarray(tri, dtype=np.int32)
        if b.shape != (m,) or t.shape != (m,):
            raise L.WtpArgumentError("wall flags need one entry per movable point")
        L.check(self.ctx._h, self._lib.wtp_relax_set_wall_flags(self.ctx._h, _vp(b), _vp(t)))

    # ---- sharded sessions ------------------------------------------------------
    def layers_dev(self, axis: int, lo_in: float, hi_in: float, lo_out: float, hi_out: float, d_lo_ptr: int,
                   d_hi_ptr: int, cap: int):
        """Boundary layers of the movable points into device buffers of `cap` packed 4-vectors each;
        returns (n_lo, n_hi, n_stray_lo, n_stray_hi) — the true counts, also when they exceed cap."""
        cnt = (C.c_int64 * 4)()
        rc = self._lib.wtp_relax_layers_dev(self.ctx._h, int(axis), float(lo_in), float(hi_in), float(lo_out),
                                            float(hi_out), C.c_void_p(d_lo_ptr), C.c_void_p(d_hi_ptr), int(cap), cnt)
        L.check(self.ctx._h, rc)
        return tuple(int(c) for c in cnt)

    def step_layers(self, rebuild, axis, lo_in, hi_in, lo_out, hi_out, d_lo_ptr, d_hi_ptr, cap):
        """One sweep + the boundary layers of the positions it produced, one synchronisation for both:
        returns (stats, (n_lo, n_hi, n_stray_lo, n_stray_hi))."""
        st = L.StepStats()
        cnt = (C.c_int64 * 4)()
        rc = self._lib.wtp_relax_step_layers(self.ctx._h, int(bool(rebuild)), C.byref(st), int(axis), float(lo_in),
                                             float(hi_in), float(lo_out), float(hi_out), C.c_void_p(d_lo_ptr),
                                             C.c_void_p(d_hi_ptr), int(cap), cnt)
        L.check(self.ctx._h, rc)
        return _stats_dict(st), tuple(int(c) for c in cnt)

    def step_layers3(self, rebuild, axes_mask, lo_in3, hi_in3, lo_out3, hi_out3, d_lo_ptrs, d_hi_ptrs, cap):
        """step_layers along up to three axes (bit a of axes_mask), one synchronisation for everything:
        returns (stats, [(n_lo, n_hi, n_stray_lo, n_stray_hi)] * 3)."""
        st = L.StepStats()
        cnt = (C.c_int64 * 12)()
        d3 = lambda v: (C.c_double * 3)(*[float(x) for x in v])
        p3 = lambda v: (C.c_void_p * 3)(*[C.c_void_p(int(x)) for x in v])
        rc = self._lib.wtp_relax_step_layers3(self.ctx._h, int(bool(rebuild)), C.byref(st), int(axes_mask), d3(lo_in3),
                                              d3(hi_in3), d3(lo_out3), d3(hi_out3), p3(d_lo_ptrs), p3(d_hi_ptrs), int(cap), cnt)
        L.check(self.ctx._h, rc)
        return _stats_dict(st), [tuple(int(cnt[4 * a + j]) for j in range(4)) for a in range(3)]

    def set_fixed_dev(self, d_fixed4_ptr: int, n_fixed_new: int):
        """Replace the fixed head of the snapshot by n_fixed_new packed 4-vectors in device memory."""
        rc = self._lib.wtp_relax_set_fixed_dev(self.ctx._h, C.c_void_p(d_fixed4_ptr) if n_fixed_new else None,
                                               int(n_fixed_new))
        L.check(self.ctx._h, rc)
        self.n += int(n_fixed_new) - self.n_fixed
        self.n_fixed = int(n_fixed_new)

    def set_coverage_box(self, lo3, hi3):
        """Block decompositions: the snapshot is complete inside the box lo3 .. hi3 (ends may be +-inf)."""
        lo = (C.c_double * 3)(*[float(v) for v in lo3])
        hi = (C.c_double * 3)(*[float(v) for v in hi3])
        L.check(self.ctx._h, self._lib.wtp_relax_set_coverage_box(self.ctx._h, lo, hi))

    def set_coverage(self, axis: int, lo: float = 0.0, hi: float = 0.0):
        """The snapshot is complete for lo <= coord[axis] <= hi (slab + ghost layers); sweeps report
        in n_uncovered the points whose neighbourhood reaches past it.  axis < 0: unlimited."""
        L.check(self.ctx._h, self._lib.wtp_relax_set_coverage(self.ctx._h, int(axis), float(lo), float(hi)))

    def close(self):
        if self._open and self.ctx._h:
            self._lib.wtp_relax_end(self.ctx._h)
        self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None or _default_ctx._h is None:
        _default_ctx = Context(0)
    return _default_ctx
