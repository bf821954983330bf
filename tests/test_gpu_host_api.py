"""The reference-shaped host API (PointCloud / set_topology / repel) end to end on the GPU,
checked with the assertions the reference's own tests make (test/topology.jl, test/repel.jl)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(wtp, nb=400, nv=2000, dtype=np.float64, seed=3):
    rng = np.random.default_rng(seed)
    # boundary: points on the faces of the unit cube; volume: uniform inside
    b = rng.random((nb, 3))
    f = rng.integers(0, 6, nb)
    b[np.arange(nb), f % 3] = (f // 3).astype(float)
    v = 0.05 + 0.9 * rng.random((nv, 3))
    return wtp.PointCloud(wtp.PointBoundary(b.astype(dtype)), wtp.PointVolume(v.astype(dtype)))


def test_set_topology_knn_and_radius(ctx, O, wtp):
    cloud = _cloud(wtp)
    c2 = wtp.set_topology(cloud, wtp.KNNTopology, 5, ctx=ctx)
    assert wtp.hastopology(c2) and not wtp.hastopology(cloud)          # functional: new cloud
    nb = wtp.neighbors(c2)
    assert nb.shape == (len(cloud), 5)
    assert not (nb == np.arange(len(cloud))[:, None]).any()             # test/topology.jl:40
    assert np.array_equal(nb, O.knn(wtp.points(cloud), 5, False, "kdtree", want_dist=False))
    assert len(wtp.neighbors(c2, 1)) == 5
    wtp.rebuild_topology(c2, ctx=ctx)
    assert c2.topology.k == 5 and wtp.neighbors(c2).shape == (len(cloud), 5)
    c3 = wtp.set_topology(cloud, wtp.RadiusTopology, 0.12, ctx=ctx)
    assert c3.topology.radius == 0.12
    off, idx = O.radius(wtp.points(cloud), 0.12)
    for i in (0, 17, len(cloud) - 1):
        assert np.array_equal(wtp.neighbors(c3, i), idx[off[i]:off[i + 1]])
    # surface- and volume-level topologies index locally (test/topology.jl:165-263)
    v2 = cloud.volume.set_topology(wtp.KNNTopology, 4, ctx=ctx)
    assert v2.neighbors().max() < len(cloud.volume)


def test_search_and_searchdists(ctx, wtp):
    N = 20
    th = np.linspace(0, 2 * np.pi, N + 1)[:-1]
    cloud = wtp.PointCloud(wtp.PointBoundary(np.stack([np.cos(th), np.sin(th)], 1)))
    m = wtp.KNearestSearch(cloud, 3)
    nb = wtp.search(cloud, m, ctx=ctx)
    assert nb.shape == (N, 3) and (nb[:, 0] == np.arange(N)).all()      # test/neighbors.jl:54-56
    idx, d = wtp.searchdists(cloud, m, ctx=ctx)
    assert (d >= 0).all() and np.allclose(d[:, 0], 0, atol=1e-10) and (np.diff(d, axis=1) >= 0).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_repel_stop_rules_and_types(ctx, wtp, dtype):
    cloud = _cloud(wtp, dtype=dtype)
    n_vol = len(cloud.volume)
    s = 0.06
    conv = []
    new = wtp.repel(cloud, wtp.ConstantSpacing(s), max_iters=3, convergence=conv, ctx=ctx)
    assert len(conv) == 3 and all(np.isfinite(conv)) and min(conv) >= 0  # test/float32_pipeline.jl:49-52
    assert new.volume.points().dtype == dtype and isinstance(new.topology, wtp.NoTopology)
    assert len(new.volume) == n_vol and np.array_equal(new.boundary.points(), cloud.boundary.points())
    conv = []
    wtp.repel(cloud, s, max_iters=30, tol=1e-12, stall_after=0, convergence=conv, ctx=ctx)
    assert len(conv) == 30                                               # test/repel.jl:292-298
    conv = []
    c_t = wtp.repel(cloud, s, max_iters=200, tol=1e-12, cv_target=10.0, convergence=conv, ctx=ctx)
    assert len(conv) == 1                                                # test/repel.jl:282-290
    assert np.array_equal(c_t.volume.points(), cloud.volume.points())    # comes back untouched
    conv = []
    wtp.repel(cloud, s, max_iters=400, tol=1e-12, stall_after=5, convergence=conv, ctx=ctx)
    assert 5 < len(conv) <= 400
    with pytest.raises(wtp.WtpArgumentError):
        wtp.repel(cloud, s, rebuild_every=0, ctx=ctx)                    # test/repel.jl:466


def test_repel_matches_oracle_loop(ctx, O, wtp):
    cloud = _cloud(wtp, dtype=np.float64)
    s = 0.06
    conv, trace = [], []
    new = wtp.repel(cloud, s, max_iters=8, tol=0.0, stall_after=0, convergence=conv, trace=trace, ctx=ctx)
    nb = len(cloud.boundary)
    ref = O.relax_loop(wtp.points(cloud), nb, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s / 20, max_iters=8, tol=0.0,
                       stall_after=0)
    assert np.allclose(conv, ref["conv"], rtol=1e-12)
    assert np.array_equal(new.volume.points(), ref["p"])                 # fp64 exact path: bit for bit
    assert len(trace) == 8 and all(t["idx_a"] < t["idx_b"] for t in trace)


def test_repel_callable_spacing_refreshed_every_sweep(ctx, O, wtp):
    """A host-callable (PER_POINT) spacing with rebuild_every > 1: the reference evaluates s = spacing(x_i) at
    the current position in every sweep (src/repel.jl:260), not only on rebuilds (:251).  fp64 -> exact path:
    the positions must equal the oracle loop's (which re-evaluates the same callable the same way) bit for bit."""
    cloud = _cloud(wtp, dtype=np.float64)
    nb = len(cloud.boundary)

    def graded(x):
        x = np.asarray(x)
        return 0.05 + 0.03 * x[:, 0]          # grows along x: a moving point's spacing changes every sweep

    conv = []
    new = wtp.repel(cloud, graded, max_iters=7, tol=0.0, stall_after=0, rebuild_every=3, convergence=conv, ctx=ctx,
                    alpha=0.05 / 20)
    ref = O.relax_loop(wtp.points(cloud), nb, graded, 2, 0.2, 1.0, 3.0, 21, 0.05 / 2000, 0.05 / 20, max_iters=7, tol=0.0,
                       rebuild_every=3, stall_after=0)
    assert len(conv) == 7
    assert np.array_equal(new.volume.points(), ref["p"])
    assert np.allclose(conv, ref["conv"], rtol=1e-12)


def test_repel_variable_spacing_and_kick(ctx, wtp):
    cloud = _cloud(wtp, dtype=np.float32)
    sp = wtp.BoundaryLayerSpacing(cloud.boundary.points(), at_wall=0.04, bulk=0.08, layer_thickness=0.3)
    conv = []
    new = wtp.repel(cloud, sp, max_iters=5, stall_after=0, kick_after=2, convergence=conv, ctx=ctx)
    assert len(conv) == 5 and np.isfinite(new.volume.points()).all()
    sp2 = wtp.LogLike(cloud.boundary.points(), 0.08, 1.3)
    assert wtp.repel(cloud, sp2, max_iters=2, stall_after=0, ctx=ctx).volume.points().shape == (len(cloud.volume), 3)


def test_near_duplicate_keep_mask(O, wtp, ctx):
    """_near_duplicate_keep_mask over the device radius search: the reference's known answers
    (test/repel.jl:301-325) and the oracle on a cloud seeded with near-duplicates."""
    from whatsthepoint_jl_amd.repel import near_duplicate_keep_mask

    pts = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [2.01, 0, 0], [3, 0, 0]], dtype=np.float64)
    keep = near_duplicate_keep_mask(pts, np.ones(5), 0.5, ctx=ctx)
    assert keep.tolist() == [True, True, True, False, True]
    assert near_duplicate_keep_mask(pts, np.ones(5), 0.0, ctx=ctx).all()
    cluster = np.array([[10.0 + 1.0e-3 * i, 0, 0] for i in range(1, 13)])
    cp = np.concatenate([pts, cluster])
    ck = near_duplicate_keep_mask(cp, np.ones(len(cp)), 0.5, ctx=ctx)
    assert ck[5] and ck[5:].sum() == 1
    rng = np.random.default_rng(12)
    for dtype in (np.float32, np.float64):
        x = wtp.synth.uniform(6000, 3, dtype, 9)
        dup = rng.choice(6000, 300, replace=False)
        x = np.concatenate([x, (x[dup] + rng.normal(0, 0.004, (300, 3))).astype(dtype)])
        sp = (0.04 + 0.03 * x[:, 0]).astype(dtype)                   # variable spacing: thresholds differ per point
        got = near_duplicate_keep_mask(x, sp, 0.4, ctx=ctx)
        want = O.cull_mask(x, sp, 0.4)
        assert np.array_equal(got, want) and 150 < (~got).sum() < 1500


def test_repel_cull_ratio(O, wtp, ctx):
    b = wtp.synth.uniform(300, 3, np.float32, 2)
    v = wtp.synth.uniform(2000, 3, np.float32, 4)
    cloud = wtp.PointCloud(wtp.PointBoundary(b), wtp.PointVolume(v))
    kw = dict(max_iters=2, stall_after=0, tol=0.0, inside=lambda p: np.ones(len(p), bool), ctx=ctx)
    everyone = wtp.repel(cloud, wtp.ConstantSpacing(0.08), **kw).volume.points()
    culled = wtp.repel(cloud, wtp.ConstantSpacing(0.08), cull_ratio=0.6, **kw).volume.points()
    want = O.cull_mask(everyone, np.full(len(everyone), 0.08, np.float32), 0.6)   # src/repel.jl:91-93
    assert np.array_equal(culled, everyone[want]) and 20 < (~want).sum() < 1500


def test_radius_offsets_on_device_equals_two_phase(ctx, O, wtp):
    """wtp_radius_offsets (scan on the device, fill with resident offsets) == the caller-side scan of
    wtp_radius_count / wtp_radius_fill == the oracle, incl. empty rows, rows past the brick kernel's 32
    entries and clouds smaller than one scan tile."""
    rng = np.random.default_rng(12)
    for n, dim, dtype, r in ((50000, 3, np.float32, 0.06), (1500, 2, np.float64, 0.05), (70000, 3, np.float32, 0.02),
                             (5, 3, np.float32, 0.5), (9000, 3, np.float32, 0.2)):
        x = rng.random((n, dim)).astype(dtype)
        off, idx = ctx.radius(x, r)
        off2, idx2 = ctx.radius_two_phase(x, r)
        ooff, oidx = O.radius(x, r)
        assert np.array_equal(off, off2) and np.array_equal(idx, idx2)
        assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("rules", [dict(tol=1e-6, stall_after=6, cv_target=0.0), dict(tol=0.05, stall_after=0, cv_target=0.0),
                                   dict(tol=1e-9, stall_after=0, cv_target=10.0), dict(tol=0.0, stall_after=0, cv_target=0.0)])
def test_stop_rules_on_the_device_equal_the_host_loop(ctx, wtp, dtype, rules):
    """wtp_relax_run_until (stop rules of src/repel.jl:305-334 evaluated on the device, sweeps enqueued in batches)
    against one wtp_relax_step per iteration with the rules applied on the host: the same number of iterations,
    the same convergence history and the same final positions, bit for bit — the sweeps enqueued behind the
    stopping one must not have touched the state."""
    import math

    n, k, max_iters = 30000, 21, 40
    x = wtp.synth.uniform(n, 3, dtype, 11)
    s = float(n) ** (-1.0 / 3.0)
    force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
    with ctx.relax(x, 2000, s, force, k, s / 2000, s / 20) as t:
        conv_d, reason, st = t.run_until(max_iters, 2, rules["tol"], rules["stall_after"], rules["cv_target"])
        p_d = t.positions()
        pd_d = t.point_data()
    conv_h, best, last_impr, why = [], math.inf, 0, 0
    with ctx.relax(x, 2000, s, force, k, s / 2000, s / 20) as t:
        i = 1
        while i <= max_iters:
            st = t.step((i - 1) % 2 == 0)
            conv_h.append(st["max_force"])
            if (rules["stall_after"] > 0 or rules["cv_target"] > 0) and st["n_move"] > 0:
                mu = st["sum_u"] / st["n_move"]
                cv = math.sqrt(max(st["sum_u2"] / st["n_move"] - mu * mu, 0.0)) / mu
                if rules["cv_target"] > 0 and cv <= rules["cv_target"]:
                    t.revert()
                    why = 2
                    break
                if rules["stall_after"] > 0:
                    if cv < best * (1 - 1.0e-3):
                        best, last_impr = cv, i
                    elif i - last_impr >= rules["stall_after"]:
                        why = 3
                        break
            if conv_h[-1] < rules["tol"]:
                why = 1
                break
            i += 1
        p_h = t.positions()
    assert reason == why and len(conv_d) == len(conv_h)
    assert np.array_equal(conv_d, np.asarray(conv_h))
    assert np.array_equal(p_d, p_h)
    if rules["cv_target"] > 0:
        assert len(conv_d) == 1 and np.array_equal(p_d, x[2000:])       # test/repel.jl:282-290: stopped after one, unchanged
    if rules["tol"] == 0.0 and rules["stall_after"] == 0:
        assert len(conv_d) == max_iters and reason == 0
    assert np.isfinite(pd_d["forces"]).all()
