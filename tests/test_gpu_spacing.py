"""Variable spacing laws on the device (include/wtp.h wtp_spacing_desc kinds 2/3,
wtp_spacing_eval): LogLike / BoundaryLayerSpacing against the oracle's restatement of
src/discretization/spacings.jl:17-22,67-72,121-133, stand-alone and inside the repel sweep."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _boundary(wtp, dtype, m=3000, dim=3, seed=31):
    b = wtp.synth.uniform(m, dim, dtype, seed)
    f = np.arange(m) % (2 * dim)
    b[np.arange(m), f % dim] = (f // dim).astype(dtype)      # points on the faces of the unit cube / square
    return b


@pytest.mark.parametrize("dtype,dim", [(np.float32, 3), (np.float64, 3), (np.float32, 2), (np.float64, 2)])
def test_spacing_eval_matches_oracle(O, wtp, ctx, dtype, dim):
    b = _boundary(wtp, dtype, 3000 if dim == 3 else 400, dim)
    x = np.concatenate([wtp.synth.uniform(20000, dim, dtype, 5) * dtype(1.4) - dtype(0.2), b[:7]])
    ll = wtp.LogLike(b, 0.08, 1.3)
    got = ll(x, ctx=ctx)
    want = O.spacing_loglike(x, b, 0.08, 1.3)
    assert got.dtype == dtype and np.array_equal(got, want)      # exact 1-NN distance, same arithmetic in T
    assert np.all(got[-7:] == 0)                                  # on a boundary point: x = 0
    bl = wtp.BoundaryLayerSpacing(b, at_wall=0.02, bulk=0.09, layer_thickness=0.3)
    got = bl(x, ctx=ctx)
    want = O.spacing_boundary_layer(x, b, 0.02, 0.09, 0.3)
    ulp = np.finfo(dtype).eps * 0.09
    assert np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))) <= 4 * ulp   # exp() differs by an ulp
    one = ll(x[3], ctx=ctx)
    assert np.isscalar(one) or one.shape == ()
    with pytest.raises(ValueError):
        ctx.spacing_eval(dict(kind=3, p0=0.02, p1=0.09, p2=0.0, boundary=b), x)     # layer_thickness must be > 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("rebuild_every", [1, 2])
def test_sweep_with_device_law_matches_oracle(O, wtp, ctx, dtype, rebuild_every):
    """The session evaluates the law at every point's current position before each sweep
    (src/repel.jl:251,260); the oracle is driven the same way, one sweep at a time."""
    n_fixed, n_move, k = 1500, 6000, 21
    b = _boundary(wtp, dtype, n_fixed)
    v = (wtp.synth.uniform(n_move, 3, dtype, 8) * dtype(0.9) + dtype(0.05))
    snap = np.concatenate([b, v])
    law = wtp.LogLike(b, 0.07, 1.2)
    alo, amax = 2e-5, 2e-3
    force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
    sess = ctx.relax(snap, n_fixed, law.desc(), force, k, alo, amax)
    try:
        sp0 = sess.spacings()
        assert np.array_equal(sp0, O.spacing_loglike(snap, b, 0.07, 1.2))           # spacing.(snap) at setup
        cur = snap.copy()
        tree = snap.copy()                                                          # positions the tree was built on
        for it in range(4):
            rebuild = it % rebuild_every == 0
            st = sess.step(rebuild)
            if rebuild:
                tree = cur.copy()
            sp = O.spacing_loglike(cur, b, 0.07, 1.2)
            sp[:n_fixed] = sp0[:n_fixed]
            r = O.relax_sweep(tree, n_fixed, sp, 2, 0.2, 1.0, 3.0, k, alo, amax, p_old=cur[n_fixed:])
            got = sess.positions()
            s_typ = 0.05
            tol = (1e-5 if dtype == np.float32 else 1e-12) * s_typ
            assert np.max(np.abs(got - r["p"])) <= tol, it
            assert abs(st["max_force"] - float(r["forces"].max())) <= 1e-4 * float(r["forces"].max())
            assert np.array_equal(sess.spacings()[n_fixed:], sp[n_fixed:])          # values this sweep used
            cur[n_fixed:] = got
    finally:
        sess.close()


def test_repel_with_boundary_layer_law_runs_on_device(O, wtp, ctx):
    b = _boundary(wtp, np.float32, 2000)
    v = wtp.synth.uniform(5000, 3, np.float32, 4) * np.float32(0.9) + np.float32(0.05)
    cloud = wtp.PointCloud(wtp.PointBoundary(b), wtp.PointVolume(v))
    law = wtp.BoundaryLayerSpacing(b, at_wall=0.03, bulk=0.07, layer_thickness=0.25)
    conv, trace = [], []
    new = wtp.repel(cloud, law, max_iters=5, stall_after=0, tol=0.0, convergence=conv, trace=trace, ctx=ctx)
    assert len(conv) == 5 and len(trace) == 5 and len(new.volume.points()) == 5000
    assert all(t["s"] > 0.03 - 1e-6 and t["s"] < 0.07 + 1e-6 for t in trace)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_law_values_stay_exact_when_points_barely_move_and_when_one_is_moved_by_hand(O, wtp, ctx, dtype):
    """The session skips the boundary-tree walk for a point whose previous walk still certifies its winner
    (wtp_spacing.hip); the value must be the brute-force one every sweep, also for a point thrown across the box."""
    n_fixed, n_move, k = 1200, 5000, 21
    b = _boundary(wtp, dtype, n_fixed)
    v = (wtp.synth.uniform(n_move, 3, dtype, 11) * dtype(0.9) + dtype(0.05))
    snap = np.concatenate([b, v])
    law = wtp.LogLike(b, 0.07, 1.2)
    force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
    with ctx.relax(snap, n_fixed, law.desc(), force, k, 1e-7, 1e-5) as sess:      # tiny steps: certificates hold
        cur = snap.copy()
        for it in range(10):
            if it == 6:                                                           # the kick of src/repel.jl:431
                far = np.array([0.93, 0.08, 0.51], dtype=dtype)
                sess.set_point(17, far)
                cur[n_fixed + 17] = far
            sess.step(it % 3 == 0)
            want = O.spacing_loglike(cur, b, 0.07, 1.2)
            assert np.array_equal(sess.spacings()[n_fixed:], want[n_fixed:]), it
            cur[n_fixed:] = sess.positions()
