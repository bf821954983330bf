"""The reference's own test items for the topology / neighbour API, one function per @testitem group,
run through the device (test/topology.jl:1-340, test/neighbors.jl:1-200).  Indices here are 0-based
(the Julia shim adds 1); unit handling (Unitful) is the shim's and not mirrored."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _grid(n):
    return np.array([(i * 0.1, j * 0.1) for i in range(n) for j in range(n)], dtype=np.float64)


def test_notopology_default_rebuild_and_neighbors(ctx, wtp):
    """topology.jl:1-10,86-104"""
    cloud = wtp.PointCloud(wtp.PointBoundary(np.random.default_rng(0).random((10, 3))))
    assert isinstance(cloud.topology, wtp.NoTopology) and not wtp.hastopology(cloud) and wtp.isvalid(cloud.topology)
    wtp.rebuild_topology(cloud, ctx=ctx)            # a no-op, no error
    assert isinstance(cloud.topology, wtp.NoTopology)
    with pytest.raises(wtp.WtpArgumentError):
        wtp.neighbors(cloud)
    with pytest.raises(wtp.WtpArgumentError):
        wtp.neighbors(cloud, 0)


def test_knn_topology_construction_and_rebuild(ctx, wtp):
    """topology.jl:12-41,68-84"""
    N, k = 20, 5
    pts = np.random.default_rng(1).random((N, 3))
    cloud = wtp.set_topology(wtp.PointCloud(wtp.PointBoundary(pts)), wtp.KNNTopology, k, ctx=ctx)
    assert wtp.hastopology(cloud) and isinstance(cloud.topology, wtp.KNNTopology) and cloud.topology.k == k
    assert wtp.isvalid(cloud.topology)
    nbrs = wtp.neighbors(cloud)
    assert len(nbrs) == N and all(len(r) == k for r in nbrs)
    assert len(wtp.neighbors(cloud, 0)) == k
    assert all(i not in wtp.neighbors(cloud, i) for i in range(N))      # no self
    before = np.array(nbrs, copy=True)
    wtp.rebuild_topology(cloud, ctx=ctx)                                 # in place, same parameters
    assert wtp.isvalid(cloud.topology) and cloud.topology.k == k and len(wtp.neighbors(cloud, 0)) == k
    assert np.array_equal(before, np.asarray(wtp.neighbors(cloud)))


def test_radius_topology_construction_and_rebuild(ctx, wtp):
    """topology.jl:43-66,265-288: 5 x 5 grid h = 0.1, r = 0.15 -> the 8-neighbourhoods (corner 3, edge 5, inside 8)"""
    pts = _grid(5)
    cloud = wtp.set_topology(wtp.PointCloud(wtp.PointBoundary(pts)), wtp.RadiusTopology, 0.15, ctx=ctx)
    assert wtp.hastopology(cloud) and isinstance(cloud.topology, wtp.RadiusTopology) and cloud.topology.radius == 0.15
    nbrs = wtp.neighbors(cloud)
    assert len(nbrs) == len(pts) and all(i not in wtp.neighbors(cloud, i) for i in range(len(pts)))
    counts = sorted(len(wtp.neighbors(cloud, i)) for i in range(len(pts)))
    assert counts == [3] * 4 + [5] * 12 + [8] * 9
    wtp.rebuild_topology(cloud, ctx=ctx)
    assert isinstance(cloud.topology, wtp.RadiusTopology) and cloud.topology.radius == 0.15
    assert 0 not in wtp.neighbors(cloud, 0)


def test_topology_pretty_printing(ctx, wtp):
    """topology.jl:106-145"""
    cloud = wtp.PointCloud(wtp.PointBoundary(np.random.default_rng(2).random((10, 3))))
    assert "NoTopology" in repr(cloud) and "NoTopology" in repr(wtp.NoTopology())
    cloud = wtp.set_topology(cloud, wtp.KNNTopology, 3, ctx=ctx)
    out = cloud.topology.show()
    assert "KNNTopology" in out and "k: 3" in out
    assert "KNNTopology" in repr(cloud)


@pytest.mark.parametrize("kind", ["surface", "volume"])
def test_surface_and_volume_level_topology(ctx, wtp, kind):
    """topology.jl:165-263,290-340: local ids, KNN and radius, rebuild keeps the parameters"""
    N, k = 20, 5
    pts = np.random.default_rng(3).random((N, 3))
    make = (lambda p: wtp.PointSurface(p, np.tile([0.0, 0.0, 1.0], (len(p), 1)), np.zeros(len(p)))) if kind == "surface" \
        else (lambda p: wtp.PointVolume(p))
    x = make(pts)
    assert isinstance(x.topology, wtp.NoTopology) and not wtp.hastopology(x)
    x = wtp.set_topology(x, wtp.KNNTopology, k, ctx=ctx)
    assert wtp.hastopology(x) and x.topology.k == k and len(wtp.neighbors(x)) == N and len(wtp.neighbors(x, 0)) == k
    g = make(np.concatenate([_grid(5), np.zeros((25, 1))], axis=1))
    g = wtp.set_topology(g, wtp.RadiusTopology, 0.15, ctx=ctx)
    assert isinstance(g.topology, wtp.RadiusTopology) and g.topology.radius == 0.15 and wtp.isvalid(g.topology)
    assert len(wtp.neighbors(g)) == 25 and all(i not in wtp.neighbors(g, i) for i in range(25))
    wtp.rebuild_topology(g, ctx=ctx)
    assert wtp.hastopology(g) and g.topology.radius == 0.15 and len(wtp.neighbors(g)) == 25


def test_knearestsearch_constructor_and_search(ctx, wtp):
    """neighbors.jl:1-136: constructor on cloud / boundary / surface; search returns self first, searchdists
    ascending distances starting at 0"""
    N, k = 20, 5
    pts = np.random.default_rng(4).random((N, 3))
    for holder in (wtp.PointCloud(wtp.PointBoundary(pts)), wtp.PointBoundary(pts), wtp.PointSurface(pts)):
        m = wtp.KNearestSearch(holder, k)
        assert m.k == k
        rows = wtp.search(holder, m, ctx=ctx)
        assert len(rows) == N and all(len(r) == k for r in rows) and all(rows[i][0] == i for i in range(N))
        idx, dist = wtp.searchdists(holder, m, ctx=ctx)
        assert np.array_equal(idx, rows) and np.allclose(dist[:, 0], 0.0, atol=1e-10)
        assert (np.diff(dist, axis=1) >= 0).all() and (dist >= 0).all()
    # the 8-point circle: self, then the two ring neighbours (neighbors.jl:34-57)
    th = np.arange(8) * np.pi / 4
    circle = np.stack([np.cos(th), np.sin(th)], axis=1)
    rows = wtp.search(wtp.PointBoundary(circle), wtp.KNearestSearch(wtp.PointBoundary(circle), 3), ctx=ctx)
    for i in range(8):
        assert rows[i][0] == i and set(rows[i][1:]) == {(i - 1) % 8, (i + 1) % 8}


def test_knearestsearch_real_geometry_and_edge_cases(ctx, wtp):
    """neighbors.jl:138-183: the box surface (its face centres are the committed fixture), k = n, k = 1"""
    z = np.load(os.path.join(GOLD, "box_surface.npz"))
    cloud = wtp.PointCloud(wtp.PointBoundary(z["centroid"], z["normal"], z["area"]))
    m = wtp.KNearestSearch(cloud, 10)
    rows = wtp.search(cloud, m, ctx=ctx)
    assert len(rows) == len(cloud) and all(len(r) == 10 for r in rows[:100])
    idx, dist = wtp.searchdists(cloud, m, ctx=ctx)
    assert dist.shape == (len(cloud), 10) and (dist >= 0).all()
    pts = np.random.default_rng(5).random((5, 3))
    five = wtp.PointCloud(wtp.PointBoundary(pts))
    assert all(len(r) == 5 for r in wtp.search(five, wtp.KNearestSearch(five, 5), ctx=ctx))     # k == n
    pts = np.random.default_rng(6).random((20, 3))
    c20 = wtp.PointCloud(wtp.PointBoundary(pts))
    rows = wtp.search(c20, wtp.KNearestSearch(c20, 1), ctx=ctx)
    assert all(len(r) == 1 and r[0] == i for i, r in enumerate(rows))
    _, d = wtp.searchdists(c20, wtp.KNearestSearch(c20, 1), ctx=ctx)
    assert np.allclose(d[:, 0], 0.0, atol=1e-10)


# ---- test/repel.jl items on the octree method (the input clouds come from `discretize` there — out of scope
# here: a thinned boundary of the same surface + random interior points stand in) ----------------------------
def _box_cloud(wtp, ctx, n_vol, seed, thin=40, dtype=np.float32):
    z = np.load(os.path.join(GOLD, "box_mesh.npz"))
    v, t = z["vertices"].astype(dtype), z["triangles"]
    oc = wtp.TriangleOctree(v, t, ctx=ctx)
    s = np.load(os.path.join(GOLD, "box_surface.npz"))
    sel = np.arange(0, len(s["centroid"]), thin)
    rng = np.random.default_rng(seed)
    cand = (rng.random((4 * n_vol + 50, 3)) * 25).astype(dtype)
    vol = cand[oc.isinside(cand)][:n_vol]
    cloud = wtp.PointCloud(wtp.PointBoundary(s["centroid"][sel].astype(dtype), s["normal"][sel].astype(dtype),
                                             s["area"][sel].astype(dtype)), wtp.PointVolume(vol))
    return cloud, oc


def test_repel_beta_kwarg_feeds_default_force_model(ctx, wtp):
    """repel.jl:185-208"""
    cloud, oc = _box_cloud(wtp, ctx, 300, 1)
    sp = wtp.ConstantSpacing(3.1)
    a = wtp.repel(cloud, sp, oc, beta=0.5, max_iters=5, ctx=ctx)
    b = wtp.repel(cloud, sp, oc, force_model=wtp.ClippedSpacingForce(0.5), max_iters=5, ctx=ctx)
    assert len(a) == len(b) and np.array_equal(a.points(), b.points())


def test_repel_other_force_models_with_octree(ctx, wtp):
    """repel.jl:210-260: SpacingEquilibriumForce / InverseDistanceForce keep every point and stay inside; the
    equilibrium law leaves a cloud whose boundary and interior densities match the target at least as well
    spaced as it started (the reference's weak invariant; it calls it tuning-dependent for the other law)."""
    cloud, oc = _box_cloud(wtp, ctx, 880, 2, thin=84)          # ~2.6 apart on the surface and inside
    sp = wtp.ConstantSpacing(2.6)
    before = wtp.spacing_metrics(cloud, sp, k=10, ctx=ctx)
    conv = []
    c_eq = wtp.repel(cloud, sp, oc, force_model=wtp.SpacingEquilibriumForce(0.2), max_iters=40, convergence=conv, ctx=ctx)
    assert len(c_eq) == len(cloud) and len(c_eq.volume) > 0 and np.isfinite(conv).all()
    assert oc.isinside(c_eq.volume.points()).all()
    assert wtp.spacing_metrics(c_eq, sp, k=10, ctx=ctx)["mean_error"] <= before["mean_error"] * 1.1
    conv = []
    c_inv = wtp.repel(cloud, sp, oc, force_model=wtp.InverseDistanceForce(0.2), max_iters=40, convergence=conv, ctx=ctx)
    assert len(c_inv) == len(cloud) and np.isfinite(conv).all() and oc.isinside(c_inv.volume.points()).all()
    assert np.isfinite(wtp.spacing_metrics(c_inv, sp, k=10, ctx=ctx)["mean_error"])


def test_repel_stall_after_and_cv_target_with_octree(ctx, wtp):
    """repel.jl:262-299"""
    cloud, oc = _box_cloud(wtp, ctx, 400, 3)
    sp = wtp.ConstantSpacing(2.8)
    conv = []
    wtp.repel(cloud, sp, oc, max_iters=200, tol=1e-12, stall_after=5, convergence=conv, ctx=ctx)
    assert 5 < len(conv) < 200
    conv_t = []
    c_t = wtp.repel(cloud, sp, oc, max_iters=200, tol=1e-12, cv_target=10.0, convergence=conv_t, ctx=ctx)
    assert len(conv_t) == 1 and np.array_equal(c_t.points(), cloud.points())
    conv_off = []
    wtp.repel(cloud, sp, oc, max_iters=30, tol=1e-12, stall_after=0, convergence=conv_off, ctx=ctx)
    assert len(conv_off) == 30


def test_repel_cull_kick_trace_rebuild_every(ctx, wtp):
    """repel.jl:375-470"""
    z = np.load(os.path.join(GOLD, "box_mesh.npz"))
    oc = wtp.TriangleOctree(z["vertices"], z["triangles"], ctx=ctx)
    s = np.load(os.path.join(GOLD, "box_surface.npz"))
    rng = np.random.default_rng(4)
    vol = (rng.random((4000, 3)) * 24.6 + 0.2).astype(np.float32)
    full = wtp.PointCloud(wtp.PointBoundary(s["centroid"], s["normal"], s["area"]), wtp.PointVolume(vol))
    sp = wtp.ConstantSpacing(0.25)                     # consistent with the tessellation (face centres ~0.22 apart)
    nocull = wtp.repel(full, sp, oc, max_iters=20, ctx=ctx)
    culled = wtp.repel(full, sp, oc, max_iters=20, cull_ratio=0.5, ctx=ctx)
    assert len(culled) <= len(nocull)
    m = wtp.metrics(culled, k=2, ctx=ctx)
    assert m["separation"] >= 0.5 * 0.25 * (1 - 1e-6)
    cloud, oc2 = _box_cloud(wtp, ctx, 300, 5)
    sp = wtp.ConstantSpacing(3.1)
    kicked = wtp.repel(cloud, sp, oc2, max_iters=30, kick_after=5, ctx=ctx)
    assert len(kicked.volume) > 0 and len(kicked) == len(cloud)
    # the volume-only method needs the whole surface for its isinside filter: the full boundary again
    sp = wtp.ConstantSpacing(0.25)
    traces = []
    wtp.repel(full, sp, max_iters=5, trace=traces, ctx=ctx)
    assert len(traces) == 5 and all(t["r_over_s"] > 0 and "idx_a" in t and "idx_b" in t for t in traces)
    c3 = wtp.repel(full, sp, max_iters=10, rebuild_every=3, ctx=ctx)
    assert len(c3.volume) > 0
    with pytest.raises(wtp.WtpArgumentError):
        wtp.repel(full, sp, rebuild_every=0, ctx=ctx)
    # 2-D clouds, a kick on the first iteration (repel.jl:417-432)
    corners = np.array([(0.0, 0.0), (1.0, 0.0), (1.0, 1.0), (0.0, 1.0)])
    inner = rng.random((40, 2)) * 0.8 + 0.1
    c2 = wtp.repel(wtp.PointCloud(wtp.PointBoundary(corners), wtp.PointVolume(inner)), wtp.ConstantSpacing(0.2), max_iters=3,
                   kick_after=1, ctx=ctx)
    assert c2.points().shape[1] == 2
