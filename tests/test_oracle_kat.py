"""Pin the CPU oracle against every known answer the reference's own tests hold for this path
 — closed-form fixtures, no Julia needed — and against independent exact k-NN
(scipy cKDTree, numpy brute force).  What stays unpinned (tie order vs NearestNeighbors.jl,
repelled coordinates) is stated in oracle/wtp_oracle.c."""
import math

import numpy as np
import pytest


def test_circle_k3_self_first(O):
    # test/neighbors.jl:34-57: 20 points on the unit circle, k=3 -> {i, i±1}, self first
    N = 20
    th = np.linspace(0, 2 * np.pi, N + 1)[:-1]
    pts = np.stack([np.cos(th), np.sin(th)], 1)
    idx, dist = O.knn(pts, 3, True, "brute")
    assert idx.shape == (N, 3)
    assert (idx[:, 0] == np.arange(N)).all()
    for i in range(N):
        assert set(idx[i, 1:]) == {(i - 1) % N, (i + 1) % N}
    assert (dist >= 0).all() and np.allclose(dist[:, 0], 0, atol=1e-10)   # test/neighbors.jl:103-107
    assert (np.diff(dist, axis=1) >= 0).all()


def test_grid_radius_8_neighbourhoods(O):
    # test/topology.jl:43-66: 5x5 grid h=0.1, r=0.15 -> corner 3, edge 5, interior 8; self not in list
    pts = np.array([(i * 0.1, j * 0.1) for i in range(5) for j in range(5)])
    for method in ("brute", "kdtree"):
        off, idx = O.radius(pts, 0.15, method)
        cnt = np.diff(off).reshape(5, 5)
        assert cnt[0, 0] == cnt[4, 4] == 3 and cnt[0, 2] == 5 and (cnt[1:4, 1:4] == 8).all()
        for i in range(25):
            assert i not in idx[off[i]:off[i + 1]]


def test_collinear_25_points(O):
    # test/metrics.jl:115-143: 25 collinear points spacing 1, k=10 / k=20
    pts = np.array([(i * 1.0, 0.0, 0.0) for i in range(1, 26)])
    for k in (10, 20):
        idx, dist = O.knn(pts, k, False, "kdtree")
        i = 12  # centre point: distances 1,1,2,2,... with ties broken by index
        assert np.allclose(dist[i], np.repeat(np.arange(1, k // 2 + 1), 2)[:k])
        assert list(idx[i][:4]) == [11, 13, 10, 14]


def test_k_edge_cases(O):
    # test/neighbors.jl:156-183: k == N and k == 1 (only self)
    pts = np.random.default_rng(0).random((5, 3))
    idx = O.knn(pts, 5, True, "brute", want_dist=False)
    assert all(sorted(r) == [0, 1, 2, 3, 4] for r in idx)
    idx = O.knn(np.random.default_rng(1).random((20, 3)), 1, True, "brute", want_dist=False)
    assert (idx[:, 0] == np.arange(20)).all()
    with pytest.raises(ValueError):
        O.knn(pts, 5, False, "brute")  # k+1 > n


def test_topology_structure(O):
    # test/topology.jl:12-41: N lists of exactly k, self never a member
    pts = np.random.default_rng(2).random((20, 3))
    idx = O.knn(pts, 5, False, "kdtree", want_dist=False)
    assert idx.shape == (20, 5) and not (idx == np.arange(20)[:, None]).any()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_force_law_known_answers(O, dtype):
    # test/repel.jl:117-183
    for u in (0.0, 0.5, 1.0, 2.0):
        assert O.force(0, 0.2, 1, 3, u, dtype) == pytest.approx(1 / (u * u + 0.2) ** 2, rel=1e-6)
        assert O.force(3, 0.2, 1, 3.0, u, dtype) == pytest.approx((1 - u * u) / (u * u + 0.2) ** 3, rel=1e-6)
        assert O.force(3, 0.2, 1, 2.0, u, dtype) == pytest.approx(O.force(1, 0.2, 1, 3, u, dtype), rel=1e-6)
    assert O.force(1, 0.2, 1, 3, 1.0, dtype) == 0.0
    assert O.force(1, 0.2, 1, 3, 0.5, dtype) > 0 > O.force(1, 0.2, 1, 3, 2.0, dtype)
    for u in (0.0, 0.3, 0.7, 0.99):
        assert O.force(2, 0.2, 1.0, 3, u, dtype) == pytest.approx(O.force(1, 0.2, 1, 3, u, dtype), rel=1e-6)
        assert O.force(2, 0.2, 1.0, 3, u, dtype) > 0
    for u in (1.0, 1.5, 10.0):
        assert O.force(2, 0.2, 1.0, 3, u, dtype) == 0.0
    assert O.force(2, 0.2, 0.8, 3, 0.79, dtype) > 0 and O.force(2, 0.2, 0.8, 3, 0.8, dtype) == 0.0
    assert O.force(3, 0.2, 1, 3.0, 1.0, dtype) == 0.0


def test_cull_mask_known_answers(O):
    # test/repel.jl:301-325
    pts = np.array([(0.0, 0, 0), (1.0, 0, 0), (2.0, 0, 0), (2.01, 0, 0), (3.0, 0, 0)])
    keep = O.cull_mask(pts, np.ones(5), 0.5)
    assert keep.sum() == 4 and keep[2] and not keep[3]
    assert O.cull_mask(pts, np.ones(5), 0.0).all()
    cluster = np.array([(10.0 + 1e-3 * i, 0, 0) for i in range(1, 13)])
    cp = np.concatenate([pts, cluster])
    ck = O.cull_mask(cp, np.ones(len(cp)), 0.5)
    assert ck[5:].sum() == 1 and ck[5]


def test_relax_loop_stop_rules(O, wtp):
    # test/repel.jl:262-299, test/float32_pipeline.jl:49-52
    n = 400
    x = wtp.synth.uniform(n, 3, np.float32, 5)
    s = float(n) ** (-1 / 3)
    kw = dict(k=21, alpha_lo=s / 2000, alpha_max=s / 20)
    r = O.relax_loop(x, 100, s, max_iters=30, tol=1e-12, stall_after=0, **kw)
    assert len(r["conv"]) == 30 and r["stop_reason"] == 0           # stall_after=0 burns the budget
    r = O.relax_loop(x, 100, s, max_iters=200, tol=1e-12, cv_target=10.0, **kw)
    assert len(r["conv"]) == 1 and r["stop_reason"] == 2             # generous cv_target: 1 iteration...
    assert np.array_equal(r["p"], x[100:])                           # ...and the cloud comes back untouched
    r = O.relax_loop(x, 100, s, max_iters=3, tol=1e-6, stall_after=50, **kw)
    assert len(r["conv"]) == 3 and np.isfinite(r["conv"]).all() and (r["conv"] >= 0).all()
    # stall stop: a jittered lattice whose pair distances all exceed s is an exact equilibrium of
    # the clipped law, so the d_NN/s CV cannot improve -> stop after stall_after more iterations
    g = np.stack(np.meshgrid(*[np.arange(7, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    g += 0.01 * wtp.synth.uniform(len(g), 3, np.float32, 9)
    r = O.relax_loop(g, 0, 0.9, max_iters=200, tol=0.0, stall_after=5, k=21, alpha_lo=1e-4, alpha_max=0.05)
    assert len(r["conv"]) == 6 and r["stop_reason"] == 3 and (r["conv"] == 0).all()
    with pytest.raises(ValueError):
        O.relax_loop(x, 100, s, max_iters=3, rebuild_every=0, **kw)  # ArgumentError (test/repel.jl:466)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", [2, 3])
def test_kdtree_equals_brute_and_scipy(O, wtp, dtype, dim):
    from scipy.spatial import cKDTree

    x = wtp.synth.uniform(4000, dim, dtype, 11)
    for k, inc in ((21, False), (22, True), (1, True), (7, False)):
        ib, db = O.knn(x, k, inc, "brute")
        ik, dk = O.knn(x, k, inc, "kdtree")
        assert np.array_equal(ib, ik) and np.array_equal(db, dk)
    d, i = cKDTree(x.astype(np.float64)).query(x.astype(np.float64), 22)
    ib, db = O.knn(x, 22, True, "brute")
    agree = (i == ib).all(axis=1)
    assert agree.mean() > 0.999                                     # tie-free rows agree index for index
    assert np.allclose(np.sort(d, 1), db, rtol=1e-5 if dtype == np.float32 else 1e-12)
    off, idx = O.radius(x, 0.07, "brute")
    off2, idx2 = O.radius(x, 0.07, "kdtree")
    assert np.array_equal(off, off2) and np.array_equal(idx, idx2)
    nb = cKDTree(x.astype(np.float64)).query_ball_point(x.astype(np.float64)[:200], 0.07)
    for q in range(200):
        mine = set(idx[off[q]:off[q + 1]])
        # pairs within 4 ulp of the radius may differ between float types
        assert len(mine ^ (set(nb[q]) - {q})) <= (1 if dtype == np.float32 else 0)


def test_numpy_brute_force_ties_by_index(O):
    # lattice: every shell is a tie; canonical order = (d2, index)
    g = np.stack(np.meshgrid(*[np.arange(6.0)] * 3, indexing="ij"), -1).reshape(-1, 3)
    idx, _ = O.knn(g, 10, False, "kdtree")
    d2 = ((g[:, None, :] - g[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    order = np.lexsort((np.broadcast_to(np.arange(len(g)), d2.shape), d2), axis=1)[:, :10]
    assert np.array_equal(idx, order)


def test_spacing_laws(O):
    # src/discretization/spacings.jl:67-72,121-133 against their closed forms
    bnd = np.array([[0.0, 0, 0], [1.0, 0, 0]])
    q = np.array([[0.5, 0.3, 0.0], [0.1, 0.0, 0.0]])
    d = np.array([math.hypot(0.5, 0.3), 0.1])
    out = O.spacing_loglike(q, bnd, 0.2, 1.3)
    assert np.allclose(out, 0.2 * d / (0.2 * (1 - 0.3) + d))
    out = O.spacing_boundary_layer(q, bnd, 0.05, 0.4, 0.6)
    assert np.allclose(out, 0.05 + 0.35 / (1 + np.exp(-(d - 0.3) / 0.1)))


def test_generator_three_implementations_agree(O, wtp):
    a = wtp.synth.uniform(5000, 3, np.float32, 20260821, first=17)
    b = O.gen_uniform(20260821, 5000, 3, np.float32, first=17)
    assert np.array_equal(a, b) and a.min() >= 0 and a.max() < 1


@pytest.mark.parametrize("m,dim,k,dtype", [(7, 3, 21, np.float32), (45, 2, 21, np.float32), (173, 2, 12, np.float32),
                                           (12, 3, 21, np.float64), (30, 2, 5, np.float64)])
def test_kdtree_equals_brute_force_on_exact_lattices(O, m, dim, k, dtype):
    """All-ties inputs: a candidate's d2, evaluated in T, may round below the exact bound of the
    subtree that holds it, so the kd-tree's pruning bound carries a few ulp of slack (found by
    tools/fuzz_parity.py: the exact bound dropped tying candidates on lattices)."""
    g = np.stack(np.meshgrid(*[np.arange(m)] * dim, indexing="ij"), -1).reshape(-1, dim) / m
    x = g.astype(dtype)
    ik, dk = O.knn(x, k)
    ib, db = O.knn(x, k, False, "brute")
    assert np.array_equal(ik, ib) and np.array_equal(dk, db)
    r = float(np.median(db[:, min(k - 1, 5)]))
    ok, rk = O.radius(x, r)
    ob, rb = O.radius(x, r, "brute")
    assert np.array_equal(ok, ob) and np.array_equal(rk, rb)


# ---- triangle-mesh geometry index of the octree repel method -------------------------------------------
def _unit_cube():
    """OctreeTestData.unit_cube_mesh (test/testsetup.jl:34-50), 0-based."""
    v = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], dtype=np.float64)
    t = np.array([(1, 3, 2), (1, 4, 3), (5, 6, 7), (5, 7, 8), (1, 2, 6), (1, 6, 5), (3, 4, 8), (3, 8, 7), (1, 5, 8),
                  (1, 8, 4), (2, 3, 7), (2, 7, 6)], dtype=np.int32) - 1
    return v, t


def test_closest_point_on_triangle_known_answers(O):
    """test/octree_geometric.jl:4-50."""
    v1, v2, v3 = np.array([0.0, 0, 0]), np.array([1.0, 0, 0]), np.array([0.0, 1, 0])
    cases = [((0.25, 0.25, 1.0), (0.25, 0.25, 0.0), 0), ((-1, -1, 0), (0, 0, 0), 1), ((2, -1, 0), (1, 0, 0), 2),
             ((-1, 2, 0), (0, 1, 0), 3), ((0.5, -0.5, 0), (0.5, 0, 0), 4), ((-0.5, 0.5, 0), (0, 0.5, 0), 5),
             ((0.3, 0.3, 0), (0.3, 0.3, 0), 0), ((1.0, 1.0, 0.0), (0.5, 0.5, 0.0), 6)]
    for dt in (np.float64, np.float32):
        for p, q, feat in cases:
            got, f = O.tri_closest(np.array(p, dtype=dt), v1, v2, v3)
            assert np.allclose(got, q, atol=1e-6 if dt == np.float32 else 1e-10)
            assert f == feat


def test_unit_cube_isinside_known_answers(O):
    """test/octree_isinside.jl:8-12,61-63 and the pseudonormal cases :113-135."""
    v, t = _unit_cube()
    d = 1.0e-3
    pts = [(0.5, 0.5, 0.5), (-0.5, 0.5, 0.5), (1.5, 0.5, 0.5), (0.3, 0.3, 0.3), (0.5, 0.5, d), (0.5, 0.5, -d),
           (d, 0.5, d), (-d, 0.5, -d)]
    want = [True, False, False, True, True, False, True, False]
    c = np.array([0.5, 0.5, 0.5])
    for corner in [(0, 0, 0), (1, 1, 1), (1, 0, 1), (0, 1, 0)]:
        corner = np.array(corner, dtype=float)
        out = (corner - c) / np.linalg.norm(corner - c)
        pts += [tuple(corner - d * out), tuple(corner + d * out)]
        want += [True, False]
    for dt in (np.float64, np.float32):
        r = O.mesh_query(v.astype(dt), t, np.array(pts, dtype=dt))
        assert r["inside"].tolist() == want
        assert np.allclose(r["sd"][:4], [-0.5, 0.5, 0.5, -0.3], atol=1e-6)
        assert abs(r["sd"][7] - np.sqrt(2) * d) < 1e-6  # edge feature: distance to the edge x = 0, z = 0


def test_cuboid_bbox_known_answers(O):
    """test/octree_isinside.jl:66-104: points far outside the 20 x 7 x 3 cuboid are exterior."""
    v, t = _unit_cube()
    v = v * np.array([20.0, 7.0, 3.0])
    pts = np.array([(5, 3.5, 1.5), (5, 3.5, 10), (5, 3.5, 20), (25, 3.5, 1.5), (5, 10, 1.5)], dtype=np.float64)
    assert O.mesh_query(v, t, pts)["inside"].tolist() == [True, False, False, False, False]


def test_mesh_projection_lands_on_the_surface(O):
    v, t = _unit_cube()
    rng = np.random.default_rng(5)
    pts = rng.random((500, 3)) * 1.6 - 0.3
    r = O.mesh_query(v, t, pts, offset=1.0e-3)
    # closest points lie on the cube's surface; the projection sits 1e-3 inside along the face normal
    on = np.isclose(r["closest"], 0.0, atol=1e-12) | np.isclose(r["closest"], 1.0, atol=1e-12)
    assert on.any(axis=1).all()
    # (a landing on an edge or a corner moves along ONE face's normal and stays on the other face's plane)
    face = r["feature"] == 0
    assert face.sum() > 100
    back = O.mesh_query(v, t, r["projected"][face])
    assert back["inside"].all() and (back["sd"] >= -1.0e-3 - 1e-9).all()
    assert np.isclose(back["sd"], -1.0e-3, atol=1e-9).mean() > 0.9  # the rest sit closer to a neighbouring face
    # brute-force distance against a dense sampling-free check: |p - closest| equals the box distance
    dbox = np.linalg.norm(np.maximum(np.maximum(-pts, pts - 1.0), 0.0), axis=1)
    inside = (pts > 0).all(axis=1) & (pts < 1).all(axis=1)
    din = np.minimum(pts, 1 - pts).min(axis=1)
    assert np.allclose(np.sqrt(r["d2"]), np.where(inside, din, dbox), atol=1e-12)
    assert (r["inside"] == inside).all()


def test_pseudonormals_of_a_closed_mesh(O):
    """Angle-weighted vertex pseudonormals of the cube point along the corner diagonals, edge
    pseudonormals along the face bisectors (Baerentzen & Aanaes; triangle_octree.jl:255-268)."""
    v, t = _unit_cube()
    pn = O.mesh_pseudonormals(v, t)
    c = np.array([0.5, 0.5, 0.5])
    for ti in range(len(t)):
        for s in range(3):
            vert = v[t[ti, s]]
            n = pn[ti, 1 + s]
            assert np.allclose(n / np.linalg.norm(n), (vert - c) / np.linalg.norm(vert - c), atol=1e-12)
        assert abs(np.linalg.norm(pn[ti, 0]) - 1) < 1e-12


# ---- consumers of the rows: PCA normals, gradient limiter ----------------------------------------------
def _fib_sphere(N=100):
    ids = np.arange(0.0, N + 0.5 + 1e-9, 1.0)          # 0.0:(N + 0.5) of test/normals.jl:60
    phi = np.arccos(1 - 2 * ids / N)
    th = np.pi * (1 + np.sqrt(5)) * ids
    return np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], axis=1)


def test_pca_normals_known_answers(O):
    """test/normals.jl:1-24,54-82: the 8-point circle with k = 3 gives the radial directions (either sign),
    three points of a plane give its normal, the Fibonacci sphere is radial within 10 degrees."""
    th = np.arange(8) * np.pi / 4
    circle = np.stack([np.cos(th), np.sin(th)], axis=1)
    nrm = O.pca_normals(circle, 3)
    assert np.allclose(np.abs((nrm * circle).sum(axis=1)), 1.0, atol=1e-12)
    tri = np.array([(1.0, 0, 0), (0, 1.0, 0), (0, 0, 1.0)])
    n3 = O.pca_normals(tri, 3)
    assert np.allclose(np.abs(n3 @ (np.ones(3) / np.sqrt(3))), 1.0, atol=1e-12)
    sph = _fib_sphere()
    ns = O.pca_normals(sph, 5)
    ang = np.degrees(np.arccos(np.clip(np.abs((ns * sph).sum(axis=1)), 0, 1)))
    assert ang.max() < 10.0
    assert np.allclose(np.linalg.norm(ns, axis=1), 1.0, atol=1e-12)


@pytest.mark.parametrize("dtype,dim", [(np.float64, 3), (np.float32, 3), (np.float64, 2)])
def test_pca_normals_match_lapack(O, dtype, dim):
    """The Jacobi restatement against numpy.linalg.eigh (LAPACK, what eigen(Symmetric(...)) calls) on the
    same covariance: the same eigenvector up to sign wherever the two smallest eigenvalues are separated."""
    rng = np.random.default_rng(8)
    base = rng.random((3000, dim)).astype(dtype)
    if dim == 3:
        base[:, 2] = (0.3 * np.sin(3 * base[:, 0]) + 0.05 * rng.standard_normal(len(base))).astype(dtype)
    k = 9
    got = O.pca_normals(base, k)
    rows, _ = O.knn(base, k, include_self=True)
    v = base[rows].astype(np.float64)
    c = v - v.mean(axis=1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", c, c) / (k - 1)
    lam, Q = np.linalg.eigh(cov)
    ref = Q[:, :, 0]
    gap = (lam[:, 1] - lam[:, 0]) / lam[:, -1]
    ok = gap > 1e-3
    dots = np.abs((got.astype(np.float64) * ref).sum(axis=1))
    assert ok.mean() > 0.9 and dots[ok].min() > (1 - 1e-10 if dtype == np.float64 else 1 - 1e-4)
    big = np.argmax(np.abs(got), axis=1)
    assert (got[np.arange(len(got)), big] > 0).all()        # the canonical sign


def test_gradient_limit_known_answer(O):
    """A chain of leaf centres 1 apart, one small source: the envelope grows by g per step until it meets
    the field (closed form), and the sweep count is the graph distance the front travels + the final check."""
    n, g = 40, 0.25
    centers = np.stack([np.arange(n, dtype=np.float64), np.zeros(n), np.zeros(n)], axis=1)
    h0 = np.full(n, 5.0)
    h0[7] = 1.0
    h, sweeps = O.gradient_limit(centers, h0, g, k=3, tol=1e-12, max_sweeps=100)
    want = np.minimum(5.0, 1.0 + g * np.abs(np.arange(n) - 7))
    assert np.allclose(h, want, atol=1e-12)
    assert sweeps == 15 + 1                                  # the front lowers 15 nodes a side (1 + 0.25*16 = 5 changes nothing), + 1 idle sweep
    h1, s1 = O.gradient_limit(centers, h0, g, k=3, tol=1e-12, max_sweeps=5)
    assert s1 == 5 and (h1 >= want - 1e-12).all() and not np.allclose(h1, want)
    # a field that is already g-Lipschitz is a fixpoint after one sweep
    h2, s2 = O.gradient_limit(centers, want, g, k=3, tol=1e-3, max_sweeps=100)
    assert s2 == 1 and np.array_equal(h2, want)
