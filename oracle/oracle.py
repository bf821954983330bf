"""ctypes loader for the CPU oracle (oracle/wtp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg — never by the product package (whatsthepoint.jl_amd/).  The oracle is a
CPU restatement of the reference's neighbour/stencil path; see the header of
wtp_oracle.c for what it follows (reference file:line) and what is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libwtp_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Returns the .so path."""
    src = [os.path.join(_HERE, f) for f in ("wtp_oracle.c", "wtp_oracle_impl.h", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.wtpo_force_f32.restype = C.c_float
        _lib.wtpo_force_f64.restype = C.c_double
        _lib.wtpo_dnn_cv_f32.restype = C.c_float
        _lib.wtpo_dnn_cv_f64.restype = C.c_double
        _lib.wtpo_force_f32.argtypes = [C.c_int] + [C.c_float] * 4
        _lib.wtpo_force_f64.argtypes = [C.c_int] + [C.c_double] * 4
    return _lib


def _suf(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _ct(dtype):
    return C.c_float if np.dtype(dtype) == np.float32 else C.c_double


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _xyz(xyz):
    xyz = np.ascontiguousarray(xyz)
    if xyz.ndim != 2 or xyz.shape[1] not in (2, 3):
        raise ValueError("xyz must be (n, 2|3)")
    return xyz


def set_num_threads(n: int) -> None:
    lib().wtpo_set_num_threads(C.c_int(n))


def num_threads() -> int:
    return int(lib().wtpo_num_threads())


def gen_uniform(seed: int, n: int, dim: int = 3, dtype=np.float32, first: int = 0):
    out = np.empty((n, dim), dtype=dtype)
    getattr(lib(), f"wtpo_gen_uniform_{_suf(dtype)}")(
        C.c_uint64(seed), C.c_int64(first), C.c_int64(n), C.c_int(dim), _p(out)
    )
    return out


def knn(xyz, k: int, include_self: bool = False, method: str = "kdtree", want_dist: bool = True):
    """Canonical k-NN lists (ascending (d2, index)).  method: 'brute' | 'kdtree'."""
    xyz = _xyz(xyz)
    n, dim = xyz.shape
    idx = np.empty((n, k), dtype=np.int32)
    dist = np.empty((n, k), dtype=xyz.dtype) if want_dist else None
    fn = getattr(lib(), f"wtpo_knn_{method}_{_suf(xyz.dtype)}")
    rc = fn(_p(xyz), C.c_int64(n), C.c_int(dim), C.c_int(k), C.c_int(int(include_self)), _p(idx), _p(dist))
    if rc != 0:
        raise ValueError("oracle knn: bad argument (k too large for n?)")
    return (idx, dist) if want_dist else idx


def radius(xyz, r: float, method: str = "kdtree"):
    """CSR radius stencils: (offsets int64[n+1], idx int32[nnz]), rows sorted by (d2, index)."""
    xyz = _xyz(xyz)
    n, dim = xyz.shape
    suf = _suf(xyz.dtype)
    counts = np.empty(n, dtype=np.int32)
    rr = _ct(xyz.dtype)(r)
    if method == "brute":
        rc = getattr(lib(), f"wtpo_radius_count_brute_{suf}")(_p(xyz), C.c_int64(n), C.c_int(dim), rr, _p(counts))
    else:
        rc = getattr(lib(), f"wtpo_radius_kdtree_{suf}")(_p(xyz), C.c_int64(n), C.c_int(dim), rr, _p(counts), None, None)
    if rc != 0:
        raise ValueError("oracle radius: bad argument")
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    idx = np.empty(int(offsets[-1]), dtype=np.int32)
    if method == "brute":
        getattr(lib(), f"wtpo_radius_fill_brute_{suf}")(_p(xyz), C.c_int64(n), C.c_int(dim), rr, _p(offsets), _p(idx))
    else:
        getattr(lib(), f"wtpo_radius_kdtree_{suf}")(_p(xyz), C.c_int64(n), C.c_int(dim), rr, _p(counts), _p(offsets), _p(idx))
    return offsets, idx


def force(kind: int, beta, u0, gamma, u, dtype=np.float64):
    fn = getattr(lib(), f"wtpo_force_{_suf(dtype)}")
    return float(fn(int(kind), float(beta), float(u0), float(gamma), float(u)))


def _spacings(spacing, n, dtype):
    if np.isscalar(spacing):
        return np.full(n, spacing, dtype=dtype)
    s = np.ascontiguousarray(spacing, dtype=dtype)
    if s.shape != (n,):
        raise ValueError("spacing array must have n entries (snapshot order)")
    return s


def relax_sweep(snap, n_fixed: int, spacing, force_kind=2, beta=0.2, u0=1.0, gamma=3.0, k=21,
                alpha_lo=None, alpha_max=None, p_old=None):
    """One sweep of _relax! (src/repel.jl:256-292).  Returns dict(p, forces, nn_dist, nn_id)."""
    snap = _xyz(snap)
    n, dim = snap.shape
    dt = snap.dtype
    n_move = n - n_fixed
    sp = _spacings(spacing, n, dt)
    p_old = np.ascontiguousarray(snap[n_fixed:] if p_old is None else p_old, dtype=dt)
    p = np.empty((n_move, dim), dtype=dt)
    forces = np.empty(n_move, dtype=dt)
    nn_dist = np.empty(n_move, dtype=dt)
    nn_id = np.empty(n_move, dtype=np.int32)
    ct = _ct(dt)
    rc = getattr(lib(), f"wtpo_relax_sweep_{_suf(dt)}")(
        _p(snap), C.c_int64(n), C.c_int64(n_fixed), C.c_int(dim), _p(p_old), _p(sp),
        C.c_int(force_kind), ct(beta), ct(u0), ct(gamma), C.c_int(k), ct(alpha_lo), ct(alpha_max),
        _p(p), _p(forces), _p(nn_dist), _p(nn_id),
    )
    if rc != 0:
        raise ValueError("oracle relax_sweep: bad argument")
    return dict(p=p, forces=forces, nn_dist=nn_dist, nn_id=nn_id)


def kd_build_seconds(xyz) -> float:
    """Wall time of one (serial) kd-tree build over xyz."""
    xyz = _xyz(xyz)
    f = getattr(lib(), f"wtpo_kd_build_seconds_{_suf(xyz.dtype)}")
    f.restype = C.c_double
    return float(f(_p(xyz), C.c_int64(xyz.shape[0]), C.c_int(xyz.shape[1])))


def relax_loop(snap, n_fixed: int, spacing, force_kind=2, beta=0.2, u0=1.0, gamma=3.0, k=21,
               alpha_lo=None, alpha_max=None, max_iters=1000, tol=1e-6, rebuild_every=1,
               stall_after=50, cv_target=0.0):
    """The whole _relax! loop (src/repel.jl:243-334).  Returns dict(p, conv, stop_reason).
    A callable `spacing` (positions -> values) is re-evaluated like the reference does: on every rebuild
    for the CV monitor's array (:251) and at p_old in every sweep for the force and the step (:260)."""
    snap = np.array(_xyz(snap), copy=True)
    n, dim = snap.shape
    dt = snap.dtype
    if callable(spacing):
        return _relax_loop_callable(snap, n_fixed, spacing, force_kind, beta, u0, gamma, k, alpha_lo, alpha_max,
                                    max_iters, tol, rebuild_every, stall_after, cv_target)
    sp = _spacings(spacing, n, dt)
    p = np.array(snap[n_fixed:], copy=True)
    conv = np.zeros(max(max_iters, 1), dtype=dt)
    reason = C.c_int(0)
    ct = _ct(dt)
    nconv = getattr(lib(), f"wtpo_relax_loop_{_suf(dt)}")(
        _p(snap), C.c_int64(n), C.c_int64(n_fixed), C.c_int(dim), _p(p), _p(sp), C.c_int(force_kind),
        ct(beta), ct(u0), ct(gamma), C.c_int(k), ct(alpha_lo), ct(alpha_max), C.c_int(max_iters),
        C.c_double(tol), C.c_int(rebuild_every), C.c_int(stall_after), C.c_double(cv_target),
        _p(conv), C.byref(reason),
    )
    if nconv < 0:
        raise ValueError("rebuild_every must be >= 1")
    return dict(p=p, conv=conv[:nconv].copy(), stop_reason=int(reason.value))


def _relax_loop_callable(snap, n_fixed, spacing, force_kind, beta, u0, gamma, k, alpha_lo, alpha_max, max_iters, tol,
                         rebuild_every, stall_after, cv_target):
    n, dim = snap.shape
    dt = snap.dtype
    ct = _ct(dt)
    sp = np.ascontiguousarray(np.asarray(spacing(snap), dtype=dt).reshape(n))
    p = np.array(snap[n_fixed:], copy=True)
    conv = np.zeros(max(max_iters, 1), dtype=dt)
    reason = C.c_int(0)
    rp = C.POINTER(ct)
    CB = C.CFUNCTYPE(None, rp, C.c_int64, rp, C.c_void_p)

    def refresh(pos, n_move, out, _user):
        x = np.ctypeslib.as_array(pos, shape=(n_move, dim))
        np.ctypeslib.as_array(out, shape=(n_move,))[:] = np.asarray(spacing(x), dtype=dt).reshape(n_move)

    cb = CB(refresh)
    nconv = getattr(lib(), f"wtpo_relax_loop_cb_{_suf(dt)}")(
        _p(snap), C.c_int64(n), C.c_int64(n_fixed), C.c_int(dim), _p(p), _p(sp), C.c_int(force_kind),
        ct(beta), ct(u0), ct(gamma), C.c_int(k), ct(alpha_lo), ct(alpha_max), C.c_int(max_iters),
        C.c_double(tol), C.c_int(rebuild_every), C.c_int(stall_after), C.c_double(cv_target),
        _p(conv), C.byref(reason), cb, None,
    )
    if nconv < 0:
        raise ValueError("rebuild_every must be >= 1")
    return dict(p=p, conv=conv[:nconv].copy(), stop_reason=int(reason.value))


def set_cv_double(on: bool):
    """Test switch of the loop's CV monitor: rules on double sums (what the device evaluates) instead of sums in the
    cloud's float type (what the reference evaluates, src/repel.jl:374-386)."""
    lib().wtpo_set_cv_double(C.c_int(1 if on else 0))


def dnn_cv(nn_dist, spacings, n_fixed: int):
    nn_dist = np.ascontiguousarray(nn_dist)
    dt = nn_dist.dtype
    sp = np.ascontiguousarray(spacings, dtype=dt)
    s1, s2 = C.c_double(0), C.c_double(0)
    cv = getattr(lib(), f"wtpo_dnn_cv_{_suf(dt)}")(
        _p(nn_dist), _p(sp), C.c_int64(len(nn_dist)), C.c_int64(n_fixed), C.byref(s1), C.byref(s2)
    )
    return float(cv), float(s1.value), float(s2.value)


def closest_pair(nn_dist, nn_id, spacings, n_fixed: int):
    nn_dist = np.ascontiguousarray(nn_dist)
    dt = nn_dist.dtype
    sp = np.ascontiguousarray(spacings, dtype=dt)
    nn_id = np.ascontiguousarray(nn_id, dtype=np.int32)
    ct = _ct(dt)
    r, s = ct(0), ct(0)
    a, b = C.c_int64(0), C.c_int64(0)
    getattr(lib(), f"wtpo_closest_pair_{_suf(dt)}")(
        _p(nn_dist), _p(nn_id), _p(sp), C.c_int64(len(nn_dist)), C.c_int64(n_fixed),
        C.byref(r), C.byref(s), C.byref(a), C.byref(b),
    )
    return dict(r=float(r.value), s=float(s.value), idx_a=int(a.value), idx_b=int(b.value))


def cull_mask(xyz, spacings, ratio: float):
    xyz = _xyz(xyz)
    n, dim = xyz.shape
    sp = np.ascontiguousarray(spacings, dtype=xyz.dtype)
    keep = np.empty(n, dtype=np.uint8)
    getattr(lib(), f"wtpo_cull_mask_{_suf(xyz.dtype)}")(
        _p(xyz), C.c_int64(n), C.c_int(dim), _p(sp), _ct(xyz.dtype)(ratio), _p(keep)
    )
    return keep.astype(bool)


def spacing_loglike(xyz, bnd, base_size, growth_rate):
    xyz, bnd = _xyz(xyz), _xyz(bnd).astype(np.asarray(xyz).dtype)
    out = np.empty(len(xyz), dtype=xyz.dtype)
    ct = _ct(xyz.dtype)
    getattr(lib(), f"wtpo_spacing_loglike_{_suf(xyz.dtype)}")(
        _p(xyz), C.c_int64(len(xyz)), C.c_int(xyz.shape[1]), _p(bnd), C.c_int64(len(bnd)),
        ct(base_size), ct(growth_rate), _p(out),
    )
    return out


def spacing_boundary_layer(xyz, bnd, at_wall, bulk, layer_thickness):
    xyz, bnd = _xyz(xyz), _xyz(bnd).astype(np.asarray(xyz).dtype)
    out = np.empty(len(xyz), dtype=xyz.dtype)
    ct = _ct(xyz.dtype)
    getattr(lib(), f"wtpo_spacing_boundary_layer_{_suf(xyz.dtype)}")(
        _p(xyz), C.c_int64(len(xyz)), C.c_int(xyz.shape[1]), _p(bnd), C.c_int64(len(bnd)),
        ct(at_wall), ct(bulk), ct(layer_thickness), _p(out),
    )
    return out


def isinside_greens(test, pts, normals, areas):
    """3-D isinside (src/isinside.jl:86-106): returns (inside bool[n], g[n])."""
    test = _xyz(test)
    dt = test.dtype
    pts, normals = _xyz(pts).astype(dt), _xyz(normals).astype(dt)
    areas = np.ascontiguousarray(areas, dtype=dt)
    g = np.empty(len(test), dtype=dt)
    inside = np.empty(len(test), dtype=np.uint8)
    getattr(lib(), f"wtpo_isinside_greens_{_suf(dt)}")(
        _p(test), C.c_int64(len(test)), _p(pts), _p(normals), _p(areas), C.c_int64(len(pts)), _p(g), _p(inside))
    return inside.astype(bool), g


def isinside_winding(test, poly):
    """2-D isinside (src/isinside.jl:17-33): returns (inside bool[n], angle sums[n])."""
    test = _xyz(test)
    dt = test.dtype
    poly = _xyz(poly).astype(dt)
    sums = np.empty(len(test), dtype=dt)
    inside = np.empty(len(test), dtype=np.uint8)
    getattr(lib(), f"wtpo_isinside_winding_{_suf(dt)}")(
        _p(test), C.c_int64(len(test)), _p(poly), C.c_int64(len(poly)), _p(sums), _p(inside))
    return inside.astype(bool), sums


# ---- triangle-mesh geometry index (octree repel method) ------------------------------------------
def tri_closest(p, a, b, c):
    """closest_point_on_triangle_feature (src/octree/geometric_utils.jl:68-136): (point, feature)."""
    p = np.ascontiguousarray(p)
    dt = p.dtype
    a, b, c = (np.ascontiguousarray(v, dtype=dt) for v in (a, b, c))
    out = np.empty(3, dtype=dt)
    feat = C.c_int32(0)
    getattr(lib(), f"wtpo_tri_closest_{_suf(dt)}")(_p(p), _p(a), _p(b), _p(c), _p(out), C.byref(feat))
    return out, feat.value


def mesh_bbox(verts):
    """_compute_bbox_raw (src/octree/triangle_octree.jl:279-291): flat axes are widened."""
    verts = np.asarray(verts)
    lo, hi = verts.min(axis=0).copy(), verts.max(axis=0).copy()
    e = max(np.finfo(verts.dtype).eps * 100, verts.dtype.type(1.0e-10))
    flat = lo == hi
    lo[flat] -= e
    hi[flat] += e
    return np.concatenate([lo, hi]).astype(verts.dtype)


def mesh_pseudonormals(verts, tris):
    """nt x 7 x 3: face normal, vertex 1..3 and edge 12/13/23 pseudonormals (triangle_octree.jl:221-277)."""
    verts = _xyz(verts)
    tris = np.ascontiguousarray(tris, dtype=np.int32)
    pn = np.zeros((len(tris), 7, 3), dtype=verts.dtype)
    getattr(lib(), f"wtpo_mesh_pseudonormals_{_suf(verts.dtype)}")(_p(verts), _p(tris), C.c_int64(len(tris)), _p(pn))
    return pn


def mesh_query(verts, tris, pts, offset=0.0):
    """Brute-force nearest triangle per point with the canonical (d2, triangle) order, and what the
    reference derives from it.  Points are converted to the mesh's type first (the reference's seam
    policy, triangle_octree.jl:80-83).  Returns a dict: d2, tri (0-based), closest, feature, sd,
    inside, projected."""
    verts = _xyz(verts)
    dt = verts.dtype
    tris = np.ascontiguousarray(tris, dtype=np.int32)
    pts = np.ascontiguousarray(np.atleast_2d(pts), dtype=dt)
    n = len(pts)
    d2 = np.empty(n, dtype=dt)
    tri = np.empty(n, dtype=np.int32)
    cp = np.empty((n, 3), dtype=dt)
    feat = np.empty(n, dtype=np.int32)
    L = lib()
    getattr(L, f"wtpo_mesh_nearest_{_suf(dt)}")(
        _p(verts), _p(tris), C.c_int64(len(tris)), _p(pts), C.c_int64(n), _p(d2), _p(tri), _p(cp), _p(feat))
    pn = mesh_pseudonormals(verts, tris)
    bbox = mesh_bbox(verts)
    sd = np.empty(n, dtype=dt)
    inside = np.empty(n, dtype=np.uint8)
    proj = np.empty((n, 3), dtype=dt)
    ct = C.c_float if dt == np.float32 else C.c_double
    getattr(L, f"wtpo_mesh_classify_{_suf(dt)}")(
        _p(pts), C.c_int64(n), _p(d2), _p(tri), _p(cp), _p(feat), _p(pn), _p(bbox), ct(offset), _p(sd), _p(inside),
        _p(proj))
    return dict(d2=d2, tri=tri, closest=cp, feature=feat, sd=sd, inside=inside.astype(bool), projected=proj, pn=pn)


# ---- consumers of the rows --------------------------------------------------------
def pca_normals(xyz, k: int = 5):
    """compute_normals (src/normals.jl:15-46,65-69) with the oracle's own k-NN rows (self included)."""
    xyz = _xyz(xyz)
    rows, _ = knn(xyz, k, include_self=True)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    out = np.empty_like(xyz)
    getattr(lib(), f"wtpo_pca_normals_{_suf(xyz.dtype)}")(
        _p(xyz), C.c_int64(len(xyz)), C.c_int(xyz.shape[1]), _p(rows), C.c_int(k), _p(out))
    return out


def gradient_limit(centers, h0, g, k: int = 12, tol: float = 1.0e-3, max_sweeps: int = 2000):
    """_gradient_limit_field (octree.jl:677-717) on the centres: (limited field, sweeps applied)."""
    c = _xyz(centers)
    dt = c.dtype
    kk = min(k, len(c))
    rows, dist = knn(c, kk, include_self=True)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=dt)
    h0 = np.ascontiguousarray(h0, dtype=dt)
    out = np.empty_like(h0)
    ct = C.c_float if dt == np.float32 else C.c_double
    f = getattr(lib(), f"wtpo_gradient_limit_{_suf(dt)}")
    f.restype = C.c_int
    sweeps = f(_p(rows), _p(dist), C.c_int64(len(c)), C.c_int(kk), _p(h0), ct(g), C.c_double(tol), C.c_int(max_sweeps),
               _p(out))
    return out, int(sweeps)
