"""GPU parity of the triangle-mesh geometry index and of the octree method of repel (SURVEY.md §8 a8;
src/repel.jl:122-181,448-469,522-537; src/octree/triangle_octree.jl:71-99,532-607) against the
brute-force CPU oracle: nearest triangle (canonical (d2, index) order), closest point, signed
distance, isinside and projection are compared BIT FOR BIT — the device walks a bounding-volume tree,
the oracle scans every triangle.  Meshes: the reference's own test surfaces box.stl / cavity.stl as
welded-vertex fixtures (tests/golden/*_mesh.npz, made by tools/make_golden.py) and its unit cube.

Nothing the reference holds pins which of two equidistant triangles its octree traversal keeps, nor
repelled coordinates: those are "parity unpinned" (oracle header); the known answers its tests do
hold (test/octree_isinside.jl, test/repel.jl:1-100) are checked here on the device."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _mesh(stem, dtype):
    z = np.load(os.path.join(GOLD, f"{stem}_mesh.npz"))
    return z["vertices"].astype(dtype), z["triangles"].astype(np.int32)


def _unit_cube(dtype=np.float64):
    v = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], dtype=dtype)
    t = np.array([(1, 3, 2), (1, 4, 3), (5, 6, 7), (5, 7, 8), (1, 2, 6), (1, 6, 5), (3, 4, 8), (3, 8, 7), (1, 5, 8),
                  (1, 8, 4), (2, 3, 7), (2, 7, 6)], dtype=np.int32) - 1
    return v, t


def _probe_points(v, t, n, dtype, seed):
    """A mix that hits every feature: uniform in the enlarged bbox, points hugging the surface on
    both sides, points exactly on vertices / edge midpoints / face centroids."""
    rng = np.random.default_rng(seed)
    lo, hi = v.min(axis=0).astype(np.float64), v.max(axis=0).astype(np.float64)
    ext = hi - lo
    a = lo - 0.2 * ext + rng.random((n // 2, 3)) * 1.4 * ext
    tri = v[t[rng.integers(0, len(t), n // 4)]].astype(np.float64)
    w = rng.dirichlet((1, 1, 1), len(tri))
    on = (tri * w[:, :, None]).sum(axis=1)
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1), 1e-300)[:, None]
    near = on + nrm * (rng.standard_normal((len(on), 1)) * 0.01 * ext.max())
    pick = t[rng.integers(0, len(t), n // 8)]
    exact = np.concatenate([v[pick[:, 0]], 0.5 * (v[pick[:, 0]] + v[pick[:, 1]]), v[pick].mean(axis=1)]).astype(np.float64)
    return np.ascontiguousarray(np.concatenate([a, near, on, exact]).astype(dtype))


def _compare(ctx, O, v, t, pts, offset):
    ctx.mesh_set(v, t)
    got = ctx.mesh_query(pts, offset)
    ref = O.mesh_query(v, t, pts, offset)
    dt = pts.dtype
    assert (got["tri"] == ref["tri"]).all(), f"{(got['tri'] != ref['tri']).sum()} nearest triangles differ"
    assert np.array_equal(got["closest"], ref["closest"].astype(dt))
    assert np.array_equal(got["sd"], ref["sd"].astype(dt))
    assert np.array_equal(got["inside"], ref["inside"])
    assert np.array_equal(got["projected"], ref["projected"].astype(dt))
    return got


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("stem", ["cube", "cavity"])
def test_mesh_queries_bit_exact(ctx, O, stem, dtype):
    v, t = _unit_cube(dtype) if stem == "cube" else _mesh(stem, dtype)
    pts = _probe_points(v, t, 20000, dtype, 11)
    got = _compare(ctx, O, v, t, pts, 1.0e-4)
    assert 0.05 < got["inside"].mean() < 0.95


def test_mesh_queries_box_stl(ctx, O):
    """46 786 triangles (the surface test/repel.jl runs on); 4000 probes keep the brute force short."""
    v, t = _mesh("box", np.float32)
    pts = _probe_points(v, t, 4000, np.float32, 12)
    _compare(ctx, O, v, t, pts, 2.5e-5)


def test_mixed_precision_seam(ctx, O):
    """Float32 points against a Float64 index: converted once at the entry, results converted back
    (src/octree/triangle_octree.jl:80-83; src/repel.jl:455-463)."""
    v, t = _mesh("cavity", np.float64)
    pts = _probe_points(v, t, 8000, np.float32, 13)
    _compare(ctx, O, v, t, pts, 1.0e-6)
    v32 = v.astype(np.float32)
    _compare(ctx, O, v32, t, pts.astype(np.float64), 1.0e-6)


def test_reference_known_answers_on_device(ctx, wtp):
    """test/octree_isinside.jl:8-12,61-63,66-104,113-135 through TriangleOctree."""
    v, t = _unit_cube()
    oc = wtp.TriangleOctree(v, t, ctx=ctx)
    d = 1.0e-3
    assert oc.isinside(np.array([0.5, 0.5, 0.5]))
    assert not oc.isinside(np.array([-0.5, 0.5, 0.5])) and not oc.isinside(np.array([1.5, 0.5, 0.5]))
    assert oc.isinside(np.array([(0.5, 0.5, 0.5), (-0.5, 0.5, 0.5), (0.3, 0.3, 0.3)])).tolist() == [True, False, True]
    assert oc.isinside(np.array([0.5, 0.5, d])) and not oc.isinside(np.array([0.5, 0.5, -d]))
    assert oc.isinside(np.array([d, 0.5, d])) and not oc.isinside(np.array([-d, 0.5, -d]))
    c = np.array([0.5, 0.5, 0.5])
    for corner in [(0, 0, 0), (1, 1, 1), (1, 0, 1), (0, 1, 0)]:
        corner = np.array(corner, dtype=float)
        out = (corner - c) / np.linalg.norm(corner - c)
        assert oc.isinside(corner - d * out) and not oc.isinside(corner + d * out)
    big = wtp.TriangleOctree(v * np.array([20.0, 7.0, 3.0]), t, ctx=ctx)
    pts = np.array([(5, 3.5, 1.5), (5, 3.5, 10), (5, 3.5, 20), (25, 3.5, 1.5), (5, 10, 1.5)], dtype=np.float64)
    assert big.isinside(pts).tolist() == [True, False, False, False, False]
    assert oc.isinside(np.array([0.25, 0.5, 0.5]))  # the first octree is uploaded again on use
    # orientation guards (test/octree_isinside.jl:138-165)
    assert abs(wtp.signed_volume(v, t) - 1.0) < 1e-12
    with pytest.raises(wtp.WtpArgumentError):
        wtp.TriangleOctree(v, t[:, ::-1].copy(), ctx=ctx)
    assert isinstance(wtp.TriangleOctree(v, t[:, ::-1].copy(), classify_leaves=False, ctx=ctx), wtp.TriangleOctree)
    flipped = t.copy()
    flipped[0] = flipped[0, ::-1]
    with pytest.raises(wtp.WtpArgumentError):
        wtp.TriangleOctree(v, flipped, ctx=ctx)


def _cloud_in_box(wtp, n_vol, dtype, seed, stem="cavity", n_bnd=None):
    v, t = _mesh(stem, dtype)
    tri = v[t].astype(np.float64)
    cen = tri.mean(axis=1)
    cr = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    area = 0.5 * np.linalg.norm(cr, axis=1)
    nrm = cr / np.maximum(2 * area, 1e-300)[:, None]
    if n_bnd is not None:
        sel = np.linspace(0, len(cen) - 1, n_bnd).astype(int)
        cen, nrm, area = cen[sel], nrm[sel], area[sel]
    rng = np.random.default_rng(seed)
    lo, hi = v.min(axis=0).astype(np.float64), v.max(axis=0).astype(np.float64)
    return v, t, cen.astype(dtype), nrm.astype(dtype), area.astype(dtype), (lo, hi, rng)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_wall_rule_sweep_matches_oracle(ctx, O, wtp, dtype):
    """One sweep + _constrain_octree (src/repel.jl:256-292,448-469) against the oracle's sweep followed
    by the oracle's projection / inside test.  Float64: bit for bit."""
    v, t, cen, nrm, area, (lo, hi, rng) = _cloud_in_box(wtp, 0, dtype, 21)
    oc = wtp.TriangleOctree(v, t, ctx=ctx)
    cand = (lo + rng.random((30000, 3)) * (hi - lo)).astype(dtype)
    vol = cand[oc.isinside(cand)][:6000]
    s = 0.09
    # volume points hugging the wall, so that some of them leave
    n_b = len(cen)
    snap = np.ascontiguousarray(np.concatenate([cen, vol]).astype(dtype))
    offset = float(dtype(1.0e-6) * np.sqrt(((hi - lo).astype(dtype) ** 2).sum(dtype=dtype)))
    sess = ctx.relax(snap, 0, s, dict(kind=2, beta=0.2, u0=1.0), 21, s / 2000, s / 20 * 8)
    try:
        sess.set_wall(n_b, offset)
        st = sess.step(True)
        got = sess.positions()
        wall = sess.get_wall()
    finally:
        sess.close()
    ref = O.relax_sweep(snap, 0, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s / 20 * 8)
    q = O.mesh_query(v, t, ref["p"], offset)
    want = ref["p"].copy()
    want[:n_b] = q["projected"][:n_b].astype(dtype)
    esc = ~q["inside"][n_b:]
    want[n_b:][esc] = snap[n_b:][esc]
    if dtype == np.float64:
        assert np.array_equal(got, want)
        assert (wall["tri"][:n_b] == q["tri"][:n_b]).all()
        assert (wall["escaped"][n_b:] == esc).all() and st["n_escaped"] == int(esc.sum())
    else:
        # fp32 sweeps agree to 1e-5 spacings; a point within that of the wall may land on either side
        clear = np.ones(len(snap), dtype=bool)
        clear[n_b:] = np.abs(q["sd"][n_b:]) > 1e-4 * s
        # (a neighbouring landing triangle changes the inward nudge by at most 2 offsets)
        assert np.abs(got - want)[clear].max() < 2e-5 * s + 4 * np.finfo(np.float32).eps * np.abs(snap).max() + 2 * offset
        assert (wall["escaped"][n_b:] == esc)[clear[n_b:]].all()
    assert wall["is_bnd"][:n_b].all() and not wall["is_bnd"][n_b:].any()
    assert (wall["tri"][n_b:] == -1).all() and not wall["escaped"][:n_b].any()
    assert esc.sum() > 0, "the test cloud should push some points through the wall"


def test_repel_octree_mirrors_reference_tests(ctx, wtp):
    """test/repel.jl:1-100: counts conserved, convergence recorded, every volume point inside, boundary
    points on the surface with unit normals."""
    dtype = np.float32
    v, t, cen, nrm, area, (lo, hi, rng) = _cloud_in_box(wtp, 0, dtype, 31, stem="box")
    oc = wtp.TriangleOctree(v, t, ctx=ctx)
    cand = (lo + rng.random((40000, 3)) * (hi - lo)).astype(dtype)
    vol = cand[oc.isinside(cand)][:30000]
    s = 0.6
    cloud = wtp.PointCloud(wtp.PointBoundary(cen, nrm, area), wtp.PointVolume(vol))
    conv = []
    out = wtp.repel(cloud, wtp.ConstantSpacing(s), oc, max_iters=10, convergence=conv, ctx=ctx)
    assert 0 < len(conv) <= 10
    assert len(out) == len(cloud) and len(out.volume) > 0
    assert len(out.boundary) == len(cloud.boundary) and list(out.boundary.surfaces) == ["boundary"]
    assert oc.isinside(out.volume.points()).all()
    q = oc.query(out.boundary.points(), want=("sd", "inside"))
    diag = float(np.linalg.norm(hi - lo))
    assert np.abs(q["sd"]).max() < 4e-6 * diag          # on the mesh, nudged ~1e-6 diag inward
    bn = out.boundary["boundary"].normals
    assert np.all(np.abs(np.linalg.norm(bn, axis=1) - 1) < 1e-3)
    assert np.array_equal(out.boundary["boundary"].areas, area)
    moved = np.linalg.norm(out.boundary.points() - cen, axis=1)
    assert moved.max() > 0 and not np.allclose(out.volume.points(), vol)
    # strong repulsion stress test (test/repel.jl:76-101): nothing is lost
    out2 = wtp.repel(cloud, wtp.ConstantSpacing(s), oc, beta=0.1, alpha=0.5, max_iters=20, ctx=ctx)
    assert len(out2) == len(cloud) and oc.isinside(out2.volume.points()).all()
    # a cv_target stop on the first iteration returns the pre-sweep configuration (test/repel.jl:355-364)
    stopped = wtp.repel(cloud, wtp.ConstantSpacing(s), oc, max_iters=30, cv_target=10.0, ctx=ctx)
    assert np.array_equal(stopped.points(), cloud.points())
    with pytest.raises(wtp.WtpArgumentError):
        wtp.repel(cloud, wtp.ConstantSpacing(s), oc, deposit_ratio=-1.0, ctx=ctx)
    with pytest.raises(wtp.WtpArgumentError):
        wtp.repel(cloud, wtp.ConstantSpacing(s), oc, rebuild_every=0, ctx=ctx)


def test_mesh_argument_errors(ctx, wtp):
    v, t = _unit_cube()
    with pytest.raises(wtp.WtpArgumentError):
        ctx.mesh_set(v, t + 8)            # indices out of range (1-based input)
    with pytest.raises(wtp.WtpArgumentError):
        ctx.mesh_set(v[:, :2], t)
    ctx.mesh_set(v, t)
    with pytest.raises(wtp.WtpArgumentError):
        ctx.mesh_query(np.zeros((4, 2)))
    ctx.mesh_clear()
    with pytest.raises(wtp.WtpError):
        ctx.mesh_query(np.zeros((4, 3)))
    snap = np.random.default_rng(1).random((500, 3)).astype(np.float32)
    sess = ctx.relax(snap, 0, 0.1, dict(kind=2, beta=0.2, u0=1.0), 21, 1e-5, 1e-3)
    try:
        with pytest.raises(wtp.WtpError):
            sess.set_wall(10, 1e-6)      # no mesh resident
        ctx.mesh_set(v, t)
        with pytest.raises(wtp.WtpArgumentError):
            sess.set_wall(501, 1e-6)
        sess.set_wall(10, 1e-6)
        with pytest.raises(wtp.WtpError):
            ctx.mesh_clear()             # in use by the session
    finally:
        sess.close()
    ctx.mesh_clear()


def test_query_knn_against_the_session_snapshot(ctx, O):
    """wtp_relax_query_knn: arbitrary positions against the tree of the last rebuild == oracle k-NN of the
    snapshot with the query appended (the query is then its own nearest hit, dropped here)."""
    rng = np.random.default_rng(3)
    snap = rng.random((5000, 3)).astype(np.float32)
    q = (rng.random((300, 3)) * 1.2 - 0.1).astype(np.float32)
    sess = ctx.relax(snap, 0, 0.05, dict(kind=2, beta=0.2, u0=1.0), 21, 1e-9, 1e-8)
    try:
        with pytest.raises(Exception):
            sess.query_knn(q, 21)        # no tree yet
        sess.step(True)
        idx, dist = sess.query_knn(q, 21, return_dist=True)
    finally:
        sess.close()
    for a in range(0, 300, 7):
        d2 = ((snap - q[a]) ** 2)
        d2 = (d2[:, 0] + d2[:, 1]) + d2[:, 2]
        order = np.lexsort((np.arange(len(snap)), d2))[:21]
        assert np.array_equal(idx[a], order)
        assert np.allclose(dist[a], np.sqrt(d2[order]), rtol=1e-6)


def test_repel_deposit_ratio_grows_the_boundary(ctx, wtp):
    """test/repel.jl:326-372: a deliberately sparse boundary; deposition converts escaped volume points
    (total conserved, boundary grows), a first-iteration cv_target stop leaves everything untouched, and
    deposition is off by default."""
    dtype = np.float32
    v, t, cen, nrm, area, (lo, hi, rng) = _cloud_in_box(wtp, 0, dtype, 41, stem="box")
    oc = wtp.TriangleOctree(v, t, ctx=ctx)
    sel = np.arange(0, len(cen), 200)
    n_sparse = len(sel)
    s = 3.0
    cand = (lo + rng.random((4000, 3)) * (hi - lo)).astype(dtype)
    vol = cand[oc.isinside(cand)][:600]
    cloud = wtp.PointCloud(wtp.PointBoundary(cen[sel], nrm[sel], area[sel]), wtp.PointVolume(vol))
    sp = wtp.ConstantSpacing(s)
    out = wtp.repel(cloud, sp, oc, max_iters=30, deposit_ratio=0.5, ctx=ctx)
    assert len(out) == len(cloud)
    assert len(out.boundary) > n_sparse
    bn = out.boundary["boundary"].normals
    assert np.all(np.abs(np.linalg.norm(bn, axis=1) - 1) < 1e-3)
    # deposited points: on the mesh, area = spacing^2, pairwise no closer than ~deposit_ratio*spacing at deposit time
    q = oc.query(out.boundary.points(), want=("sd",))
    assert np.abs(q["sd"]).max() < 1e-3
    assert np.allclose(out.boundary["boundary"].areas[n_sparse:], s * s)
    assert oc.isinside(out.volume.points()).all()
    stopped = wtp.repel(cloud, sp, oc, max_iters=30, deposit_ratio=0.5, cv_target=10.0, ctx=ctx)
    assert len(stopped.boundary) == n_sparse and np.array_equal(stopped.points(), cloud.points())
    no_dep = wtp.repel(cloud, sp, oc, max_iters=5, ctx=ctx)
    assert len(no_dep.boundary) == n_sparse


def test_degenerate_triangles_and_exact_ties(ctx, O):
    """Triangle soups on a coarse lattice: zero-area triangles (their edge parameter is 0/0, the distance NaN,
    never the minimum — `d2 < best` in the reference, src/octree/triangle_octree.jl:541), repeated corners,
    and exact distance ties between triangles (canonical: the smaller index).  tools/fuzz_mesh.py found the
    NaN-guess case this pins."""
    rng = np.random.default_rng(17)
    for nt, lattice, mdt, pdt in ((3, 4, np.float32, np.float64), (7, 2, np.float64, np.float32), (400, 2, np.float64, np.float64),
                                  (3000, 4, np.float32, np.float32)):
        v = (rng.integers(0, lattice + 1, (3 * nt, 3)).astype(np.float64) / lattice).astype(mdt)
        t = rng.integers(0, 3 * nt, (nt, 3)).astype(np.int32)
        q = np.concatenate([rng.integers(-1, lattice + 2, (500, 3)).astype(np.float64) / lattice, rng.random((500, 3)) * 1.4 - 0.2]).astype(pdt)
        ctx.mesh_set(v, t)
        got = ctx.mesh_query(q, 1e-6)
        ref = O.mesh_query(v, t, q, 1e-6)
        ok = ref["tri"] >= 0
        assert ok.any() and np.array_equal(got["tri"], ref["tri"])
        assert np.array_equal(got["closest"][ok], ref["closest"].astype(pdt)[ok])
        assert np.array_equal(got["sd"][ok], ref["sd"].astype(pdt)[ok]) and np.array_equal(got["inside"], ref["inside"])
    # nothing but degenerate triangles: no nearest element at all
    v = np.array([(0, 0, 0), (0, 0, 0), (1, 0, 0)], dtype=np.float64)
    t1 = np.array([(0, 1, 2)], dtype=np.int32)
    assert O.mesh_query(v, t1, np.array([(0.5, 0.3, 0.0)]))["tri"][0] == -1
    ctx.mesh_set(v, t1)
    got = ctx.mesh_query(np.array([(0.5, 0.3, 0.0)]))
    assert got["tri"][0] == -1 and not got["inside"][0] and np.isinf(got["sd"][0])
    ctx.mesh_clear()


def test_set_points_batch(ctx, wtp):
    """wtp_relax_set_batch == a loop of wtp_relax_set (used by the deposition pass)."""
    rng = np.random.default_rng(9)
    snap = rng.random((3000, 3)).astype(np.float32)
    idx = np.sort(rng.choice(2900, 400, replace=False))
    newp = rng.random((400, 3)).astype(np.float32)
    outs = []
    for batch in (True, False):
        sess = ctx.relax(snap, 100, 0.07, dict(kind=2, beta=0.2, u0=1.0), 21, 1e-5, 1e-3)
        try:
            sess.step(True)
            if batch:
                sess.set_points(idx, newp)
            else:
                for i, p in zip(idx, newp):
                    sess.set_point(int(i), p)
            moved = sess.positions()
            sess.step(True)
            outs.append((moved, sess.positions()))
            with pytest.raises(wtp.WtpArgumentError):
                sess.set_points([5, 5], newp[:2])
            with pytest.raises(wtp.WtpArgumentError):
                sess.set_points([2900], newp[:1])
        finally:
            sess.close()
    assert np.array_equal(outs[0][0][idx], newp)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
