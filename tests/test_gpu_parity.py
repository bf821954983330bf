"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bar: neighbour indices and distances bit-exact; repelled coordinates within
1e-5 spacings after one sweep (fp32; summation order differs inside the brick kernel)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORCE = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)


def _cloud(wtp, n, dim, dtype, seed=20260821):
    return wtp.synth.uniform(n, dim, dtype, seed)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("n,k,inc", [(5, 4, False), (5, 5, True), (20, 1, True), (20, 3, True), (200, 5, False),
                                     (3000, 21, False), (3000, 22, True), (50000, 21, False), (50000, 40, False),
                                     (20000, 60, False)])
def test_knn_matches_oracle(ctx, O, wtp, dtype, dim, n, k, inc):
    x = _cloud(wtp, n, dim, dtype)
    idx, dist = ctx.knn(x, k, include_self=inc, return_dist=True)
    oi, od = O.knn(x, k, inc, "kdtree" if n > 2000 else "brute")
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)


def test_knn_ties_lattice(ctx, O):
    # a perfect lattice: every distance shell is a tie; canonical (d2, index) order must hold
    g = np.stack(np.meshgrid(*[np.arange(12, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    idx, dist = ctx.knn(g, 21, include_self=False, return_dist=True)
    oi, od = O.knn(g, 21, False, "brute")
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def test_knn_coincident_and_clustered(ctx, O, wtp):
    x = _cloud(wtp, 4000, 3, np.float32)
    x[100:140] = x[100]            # 40 coincident points
    x[2000:3000] = x[2000] + 1e-4 * (x[2000:3000] - 0.5)  # a dense cluster (LDS halo / ring stress)
    idx = ctx.knn(x, 21, include_self=True)
    oi = O.knn(x, 21, True, "brute", want_dist=False)
    assert np.array_equal(idx, oi)


def test_knn_errors(ctx, wtp):
    x = _cloud(wtp, 10, 3, np.float32)
    with pytest.raises(wtp.WtpArgumentError):
        ctx.knn(x, 10, include_self=False)   # k+1 > n: the reference's kd-tree throws
    with pytest.raises(wtp.WtpArgumentError):
        ctx.knn(x, 0)
    with pytest.raises(wtp.WtpArgumentError):
        ctx.knn(np.zeros((10, 4), np.float32), 2)
    assert ctx.knn(x, 9, include_self=False).shape == (10, 9)   # k == n-1 is the edge that works


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim,n,r", [(2, 500, 0.1), (3, 4000, 0.08), (3, 30000, 0.03), (3, 300, 5.0), (3, 300, 0.0),
                                     (3, 1500, 5.0)])  # last: rows of 1499 > the wave kernel's LDS list
def test_radius_matches_oracle(ctx, O, wtp, dtype, dim, n, r):
    x = _cloud(wtp, n, dim, dtype)
    off, idx = ctx.radius(x, r)
    ooff, oidx = O.radius(x, r, "kdtree" if n > 2000 else "brute")
    assert np.array_equal(off, ooff)
    assert np.array_equal(idx, oidx)


def test_radius_grid_known_answer(ctx):
    # test/topology.jl:43-66: 5x5 grid h=0.1, r=0.15 -> 8-neighbourhoods (corner 3, edge 5, interior 8)
    pts = np.array([(i * 0.1, j * 0.1) for i in range(5) for j in range(5)], dtype=np.float64)
    off, idx = ctx.radius(pts, 0.15)
    cnt = np.diff(off).reshape(5, 5)
    assert cnt[0, 0] == 3 and cnt[0, 2] == 5 and cnt[2, 2] == 8
    for i in range(25):
        assert i not in idx[off[i]:off[i + 1]]


def _sweep_args(n):
    s = float(n) ** (-1.0 / 3.0)
    return s, s / 2000, s / 20


@pytest.mark.parametrize("n,n_fixed,k", [(3000, 0, 21), (3000, 1000, 21), (40000, 0, 21), (40000, 5000, 12)])
def test_relax_one_sweep_matches_oracle(ctx, O, wtp, n, n_fixed, k):
    x = _cloud(wtp, n, 3, np.float32)
    s, alo, amax = _sweep_args(n)
    with ctx.relax(x, n_fixed, s, FORCE, k, alo, amax) as sess:
        st = sess.step(True)
        p = sess.positions()
        pd = sess.point_data()
    ref = O.relax_sweep(x, n_fixed, s, 2, 0.2, 1.0, 3.0, k, alo, amax)
    assert np.array_equal(pd["nn_id"], ref["nn_id"])
    assert np.array_equal(pd["nn_dist"], ref["nn_dist"])
    assert np.abs(p - ref["p"]).max() <= 1e-5 * s
    assert np.allclose(pd["forces"], ref["forces"], rtol=1e-4, atol=1e-6)
    assert st["n_move"] == n - n_fixed
    assert st["max_force"] == pytest.approx(float(ref["forces"].max()), rel=1e-4)
    cv, s1, s2 = O.dnn_cv(ref["nn_dist"], np.full(n, s, np.float32), n_fixed)
    assert st["sum_u"] == pytest.approx(s1, rel=1e-6) and st["sum_u2"] == pytest.approx(s2, rel=1e-6)
    cp = O.closest_pair(ref["nn_dist"], ref["nn_id"], np.full(n, s, np.float32), n_fixed)
    assert {st["argmin_i"], st["argmin_j"]} == {cp["idx_a"], cp["idx_b"]}
    assert st["argmin_r"] == pytest.approx(cp["r"], rel=0, abs=0)


@pytest.mark.parametrize("dtype,dim", [(np.float64, 3), (np.float32, 2), (np.float64, 2)])
def test_relax_generic_path_bit_exact(ctx, O, wtp, dtype, dim):
    # fp64 and 2-D run the generic kernel, which sums forces in ascending (d2, id) order like the
    # reference: coordinates must match the oracle to the last bit
    n = 5000
    x = _cloud(wtp, n, dim, dtype)
    s = float(n) ** (-1.0 / dim)
    with ctx.relax(x, 500, s, FORCE, 21, s / 2000, s / 20) as sess:
        sess.step(True)
        p = sess.positions()
    ref = O.relax_sweep(x, 500, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s / 20)
    if dim == 3 or dtype == np.float64:
        assert np.array_equal(p, ref["p"])
    else:
        assert np.abs(p - ref["p"]).max() <= 1e-5 * s


@pytest.mark.parametrize("kind,beta,u0,gamma", [(0, 0.2, 1.0, 3.0), (1, 0.2, 1.0, 3.0), (2, 0.5, 0.8, 3.0), (3, 0.2, 1.0, 3.0),
                                                 (3, 0.3, 1.0, 2.0)])
def test_relax_force_models(ctx, O, wtp, kind, beta, u0, gamma):
    n = 6000
    x = _cloud(wtp, n, 3, np.float32)
    s, alo, amax = _sweep_args(n)
    with ctx.relax(x, 0, s, dict(kind=kind, beta=beta, u0=u0, gamma=gamma), 21, alo, amax) as sess:
        sess.step(True)
        p = sess.positions()
    ref = O.relax_sweep(x, 0, s, kind, beta, u0, gamma, 21, alo, amax)
    assert np.abs(p - ref["p"]).max() <= 2e-5 * s


def test_relax_multi_iteration_tracks_oracle(ctx, O, wtp):
    n = 20000
    x = _cloud(wtp, n, 3, np.float32)
    s, alo, amax = _sweep_args(n)
    with ctx.relax(x, 0, s, FORCE, 21, alo, amax) as sess:
        conv, last = sess.run(10, 1)
        p = sess.positions()
    ref = O.relax_loop(x, 0, s, 2, 0.2, 1.0, 3.0, 21, alo, amax, max_iters=10, tol=0.0, rebuild_every=1, stall_after=0)
    assert len(conv) == 10 == len(ref["conv"])
    assert np.allclose(conv, ref["conv"], rtol=1e-3)
    # 10 chained sweeps: rounding differences may flip a neighbour set here and there
    err = np.abs(p - ref["p"]).max(axis=1) / s
    assert np.quantile(err, 0.999) < 1e-3


def test_relax_stale_snapshot_with_fixed_points_and_larger_cloud(ctx, O, wtp):
    # rebuild_every = 3 (src/repel.jl:245): two of three sweeps run against a stale snapshot — the default law takes the
    # ball kernel there (every query, the point where it is now), fixed points and whatever it cannot certify the exact path
    n, n_fixed = 40000, 4000
    x = _cloud(wtp, n, 3, np.float32)
    s, alo, amax = _sweep_args(n)
    with ctx.relax(x, n_fixed, s, FORCE, 21, alo, amax) as sess:
        conv, last = sess.run(6, 3)
        p = sess.positions()
    ref = O.relax_loop(x, n_fixed, s, 2, 0.2, 1.0, 3.0, 21, alo, amax, max_iters=6, tol=0.0, rebuild_every=3, stall_after=0)
    assert np.allclose(conv, ref["conv"], rtol=1e-3)
    err = np.abs(p - ref["p"]).max(axis=1) / s
    assert np.quantile(err, 0.999) < 1e-3
    assert last["n_move"] == n - n_fixed


def test_relax_stale_snapshot_float64_bit_exact(wtp, O, monkeypatch):
    # Float64, rebuild_every = 3: the stale sweeps take the Float64 ball kernel (sums in ascending (d2, index) order) —
    # the same bits as the exact wave path (WTP_BALL64=0) and as the oracle's loop
    n, n_fixed = 30000, 2000
    x = _cloud(wtp, n, 3, np.float64)
    s, alo, amax = _sweep_args(n)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("WTP_BALL64", flag)
        with wtp.Context(0) as c:
            with c.relax(x, n_fixed, s, FORCE, 21, alo, amax) as sess:
                conv, last = sess.run(6, 3)
                res[flag] = (sess.positions(), np.asarray(conv))
    assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1])
    ref = O.relax_loop(x, n_fixed, s, 2, 0.2, 1.0, 3.0, 21, alo, amax, max_iters=6, tol=0.0, rebuild_every=3, stall_after=0)
    assert np.array_equal(res["1"][0], ref["p"])


def test_relax_stale_snapshot_rebuild_every(ctx, O, wtp):
    n = 8000
    x = _cloud(wtp, n, 3, np.float32)
    s, alo, amax = _sweep_args(n)
    with ctx.relax(x, 0, s, FORCE, 21, alo, amax) as sess:
        conv, _ = sess.run(6, 3)
        p = sess.positions()
    ref = O.relax_loop(x, 0, s, 2, 0.2, 1.0, 3.0, 21, alo, amax, max_iters=6, tol=0.0, rebuild_every=3, stall_after=0)
    assert np.allclose(conv, ref["conv"], rtol=1e-3)
    err = np.abs(p - ref["p"]).max(axis=1) / s
    assert np.quantile(err, 0.999) < 1e-3


def test_relax_revert_set_and_per_point_spacing(ctx, O, wtp):
    n = 5000
    x = _cloud(wtp, n, 3, np.float32)
    s, alo, amax = _sweep_args(n)
    sp = (s * (1.0 + 0.5 * x[:, 0])).astype(np.float32)
    with ctx.relax(x, 0, sp, FORCE, 21, alo, amax) as sess:
        sess.step(True)
        p1 = sess.positions()
        sess.revert()                       # p .= p_old (cv_target stop, src/repel.jl:314)
        assert np.array_equal(sess.positions(), x)
        with pytest.raises(wtp.WtpError):
            sess.revert()
        sess.set_point(7, [0.5, 0.5, 0.5])  # the kick's write (src/repel.jl:431)
        q = sess.positions()
        assert np.array_equal(q[7], np.float32([0.5, 0.5, 0.5])) and np.array_equal(np.delete(q, 7, 0), np.delete(x, 7, 0))
    ref = O.relax_sweep(x, 0, sp, 2, 0.2, 1.0, 3.0, 21, alo, amax)
    assert np.abs(p1 - ref["p"]).max() <= 1e-5 * s


def test_relax_errors(ctx, wtp):
    x = _cloud(wtp, 100, 3, np.float32)
    with pytest.raises(wtp.WtpArgumentError):
        ctx.relax(x, 0, -1.0, FORCE, 21, 0.0, 1.0)
    with pytest.raises(wtp.WtpArgumentError):
        ctx.relax(x, 101, 0.1, FORCE, 21, 0.0, 1.0)
    with ctx.relax(x, 0, 0.1, FORCE, 21, 0.0, 1.0) as sess:
        with pytest.raises(wtp.WtpArgumentError):
            sess.run(3, 0)  # rebuild_every=0 throws ArgumentError (test/repel.jl:466)
        with pytest.raises(wtp.WtpError):
            ctx.knn(x, 3)   # context busy with the relax session


def test_large_cloud_properties(ctx, wtp):
    # size-independent properties at 2 M points (the oracle would take minutes)
    n, k = 2_000_000, 21
    x = _cloud(wtp, n, 3, np.float32)
    idx, dist = ctx.knn(x, k, include_self=False, return_dist=True)
    assert idx.min() >= 0 and idx.max() < n
    assert not (idx == np.arange(n)[:, None]).any()                  # self excluded (test/topology.jl:40)
    assert (np.diff(dist, axis=1) >= 0).all()                         # ascending (test/neighbors.jl:105)
    rows = np.random.default_rng(0).integers(0, n, 2000)
    d = np.sqrt(((x[rows, None, :] - x[idx[rows]]) ** 2).sum(-1, dtype=np.float32))
    assert np.allclose(d, dist[rows], rtol=1e-6, atol=0)
    # each sampled row is the true k-NN set: nothing closer than the k-th distance is missing
    for r in rows[:200]:
        d2 = ((x - x[r]) ** 2).sum(1)
        assert (d2 < np.float32(dist[r, -1]) ** 2 * (1 - 1e-6)).sum() - 1 <= k


def test_compact_support_sweep_equals_full_selection(wtp, O):
    """ClippedSpacingForce: the default compact-support sweep (count-certified, no k-selection) and
    the explicit k-selection path (WTP_FULL_SELECT=1) must give the same step — both are the
    reference's sum over the k nearest, the former just proves it does not need the list."""
    import os

    n = 60000
    x = _cloud(wtp, n, 3, np.float32)
    s, alo, amax = _sweep_args(n)
    outs = {}
    for mode in ("0", "1"):
        os.environ["WTP_FULL_SELECT"] = mode
        try:
            with wtp.Context(0) as c, c.relax(x, 5000, s, FORCE, 21, alo, amax) as sess:
                st = sess.step(True)
                outs[mode] = (sess.positions(), sess.point_data(), st)
        finally:
            os.environ.pop("WTP_FULL_SELECT", None)
    (p0, d0, s0), (p1, d1, s1) = outs["0"], outs["1"]
    assert np.array_equal(d0["nn_id"], d1["nn_id"]) and np.array_equal(d0["nn_dist"], d1["nn_dist"])
    assert np.abs(p0 - p1).max() <= 1e-5 * s          # same terms; the two grids scan (sum) in different orders
    assert (p0 == p1).mean() > 0.9
    assert s0["n_fallback"] < s1["n_fallback"]         # and it certifies far more queries locally
    ref = O.relax_sweep(x, 5000, s, 2, 0.2, 1.0, 3.0, 21, alo, amax)
    assert np.array_equal(d0["nn_id"], ref["nn_id"]) and np.abs(p0 - ref["p"]).max() <= 1e-5 * s


def test_compact_support_dense_cluster_and_large_spacing(ctx, O, wtp):
    # more than k points inside the law's support (dense cluster) and a spacing far above the
    # mean distance: the count certificate must hand those queries to the exact path
    n = 20000
    x = _cloud(wtp, n, 3, np.float32)
    x[3000:3400] = x[3000] + 2e-3 * (x[3000:3400] - 0.5)
    for s in (float(n) ** (-1 / 3), 3.0 * float(n) ** (-1 / 3)):
        with ctx.relax(x, 0, s, FORCE, 21, s / 2000, s / 20) as sess:
            sess.step(True)
            p = sess.positions()
            pd = sess.point_data()
        ref = O.relax_sweep(x, 0, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s / 20)
        assert np.array_equal(pd["nn_id"], ref["nn_id"])
        assert np.abs(p - ref["p"]).max() <= 2e-5 * s


def _surface_cloud(n=46786, L=25.0, seed=5, dtype=np.float32):
    """Stand-in for BASELINE config C1 (box.stl face centroids: 46 786 points on the faces of a
    25^3 box; the STL itself does not travel to the GPU box): points on a 2-D manifold in 3-D,
    i.e. cells along the faces are crowded and the interior is empty."""
    rng = np.random.default_rng(seed)
    p = rng.random((n, 3)) * L
    f = rng.integers(0, 6, n)
    p[np.arange(n), f % 3] = (f // 3) * L
    return p.astype(dtype)


def test_c1_surface_cloud_knn_and_sweep(ctx, O):
    x = _surface_cloud()
    idx, dist = ctx.knn(x, 21, include_self=False, return_dist=True)
    oi, od = O.knn(x, 21, False, "kdtree")
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    idx = ctx.knn(x, 10, include_self=True)                                  # test/neighbors.jl:138-154 shapes
    assert idx.shape == (len(x), 10) and (idx[:, 0] == np.arange(len(x))).all()
    s = 25.0 / 8                                                             # _relative_spacing-like (coarse)
    for sp in (0.3, s):
        with ctx.relax(x, 20000, sp, FORCE, 21, sp / 2000, sp / 20) as sess:
            sess.step(True)
            p = sess.positions()
            pd = sess.point_data()
        ref = O.relax_sweep(x, 20000, sp, 2, 0.2, 1.0, 3.0, 21, sp / 2000, sp / 20)
        assert np.array_equal(pd["nn_id"], ref["nn_id"])
        assert np.abs(p - ref["p"]).max() <= 2e-5 * sp
