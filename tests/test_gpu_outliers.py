"""Clouds whose bounding box is stretched by a few far outliers: the hash lays its grid over the
bulk (quantile box) and clamps the rest into the edge cells; results must stay exact and the run
must not degrade to the quadratic regime."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(wtp, n, dtype, far):
    x = wtp.synth.uniform(n, 3, dtype, 23)
    x[:5] = np.array([[far, 0.3, 0.3], [-far, 0.7, 0.2], [0.5, far, 0.5], [0.2, 0.2, -far], [far, far, far]], dtype=dtype)
    return x


@pytest.mark.parametrize("dtype,far", [(np.float32, 3e3), (np.float32, 1e7), (np.float64, 1e9)])
def test_knn_with_far_outliers_is_exact_and_fast(O, wtp, ctx, dtype, far):
    n = 60000
    x = _cloud(wtp, n, dtype, far)
    ctx.knn(x[:2000], 8)                                     # warm the context up
    t0 = time.perf_counter()
    idx, dist = ctx.knn(x, 21, return_dist=True)
    dt = time.perf_counter() - t0
    widx, wdist = O.knn(x, 21)
    assert np.array_equal(idx, widx) and np.array_equal(dist, wdist)
    assert dt < 0.5, dt                                      # the degenerate grid took seconds here


def test_sweep_with_far_outliers_matches_oracle(O, wtp, ctx):
    n = 40000
    x = _cloud(wtp, n, np.float32, 1e6)
    s = float(n) ** (-1.0 / 3.0)
    with ctx.relax(x, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20) as t:
        st = t.step(True)
        got = t.positions()
        pd = t.point_data()
    r = O.relax_sweep(x, 0, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s / 20)
    err = np.abs(got[5:] - r["p"][5:]).max() / s
    assert err < 1e-4                                         # bulk: fp32 tolerance at coordinates ~1
    assert np.array_equal(got[:5], r["p"][:5])               # the outliers feel no force and stay put
    assert np.array_equal(pd["nn_id"], r["nn_id"])
    assert st["n_move"] == n


def test_non_finite_points_do_not_disturb_the_rest(O, wtp, ctx):
    n = 20000
    x = wtp.synth.uniform(n, 3, np.float32, 29)
    bad = np.array([[np.nan, 0.5, 0.5], [0.5, np.inf, 0.5], [-np.inf, np.nan, 0.1]], dtype=np.float32)
    idx, dist = ctx.knn(np.concatenate([x, bad]), 21, return_dist=True)
    widx, wdist = O.knn(x, 21)
    assert np.array_equal(idx[:n], widx) and np.array_equal(dist[:n], wdist)   # never closer than anything finite


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_expanding_cloud_with_a_kept_grid(O, wtp, ctx, dtype):
    """A relax session recomputes its bounding box every 8th rebuild only; in between, points that leave the
    kept box are clamped into edge cells (unbounded outward).  A compressed cloud that expands by up to a
    spacing per sweep walks far out of its first box: 14 sweeps must still match the oracle loop (Float64:
    bit for bit, the summation order does not depend on the binning there)."""
    n = 6000
    s = 0.02
    rng = np.random.default_rng(77)
    x = (0.5 + (rng.random((n, 3)) - 0.5) * 8 * s).astype(dtype)        # 6000 points in a cube of 8 spacings
    force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
    iters = 14
    with ctx.relax(x, 0, s, force, 21, s / 2000, s) as t:
        conv, st = t.run(iters, 1)
        got = t.positions()
        t.step(True)
        nxt = t.positions()
    spread0 = np.ptp(x, axis=0).max()
    assert np.ptp(got, axis=0).max() > spread0 + 3 * s                 # it really left its first box (cells are ~1.5 s)
    if dtype == np.float64:
        ref = O.relax_loop(x, 0, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s, max_iters=iters, tol=0.0, rebuild_every=1,
                           stall_after=0, cv_target=0.0)
        assert np.array_equal(got, ref["p"])
        assert np.array_equal(conv, ref["conv"])
    # one more sweep from the evolved state, searched in the kept grid, against the oracle's sweep from the same
    # state (fp32 trajectories of a compressed cloud under full-length steps diverge chaotically; single sweeps
    # do not)
    one = O.relax_sweep(got, 0, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s)
    tol = 0.0 if dtype == np.float64 else max(2e-5, 4 * np.finfo(np.float32).eps * float(np.abs(got).max()) / s)
    assert np.abs(nxt - one["p"]).max() / s <= tol
