"""GPU parity of the k-selection kernels on the x-slowest layout (wtp_ksel.hip: fp32, 3-D, k + self <= 24, n >= 4096)
against the CPU oracle and against the 4 x 4 x 4-brick kernels they replace (WTP_KSEL=0).  Bar: neighbour rows and
distances bit-exact; repelled coordinates within 2e-5 spacings after one sweep (the summation order differs)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sweep_args(n):
    s = float(n) ** (-1.0 / 3.0)
    return s, s / 2000, s / 20


@pytest.mark.parametrize("n,k,inc", [(4096, 1, False), (5000, 5, False), (20000, 21, False), (20000, 22, True),
                                     (20000, 21, True), (30000, 12, False), (100000, 21, False), (40000, 23, False),
                                     (40000, 24, True), (40000, 24, False)])  # (the last: k + self = 25, the brick kernels)
def test_ksel_knn_matches_oracle(ctx, O, wtp, n, k, inc):
    x = wtp.synth.uniform(n, 3, np.float32, 20260821 + n)
    idx, dist = ctx.knn(x, k, include_self=inc, return_dist=True)
    oi, od = O.knn(x, k, inc, "kdtree")
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)


def test_ksel_lattice_ties(ctx, O):
    # a perfect 17^3 lattice: every distance shell is a mass tie, many reach past the window the kernel orders exactly
    # (those rows go to the exact path); canonical (d2, index) order must hold everywhere
    g = np.stack(np.meshgrid(*[np.arange(17, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    idx, dist = ctx.knn(g, 21, include_self=False, return_dist=True)
    oi, od = O.knn(g, 21, False, "kdtree")
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def test_ksel_coincident_cluster_and_slab(ctx, O, wtp):
    x = wtp.synth.uniform(12000, 3, np.float32, 5)
    x[100:140] = x[100]                                     # 40 coincident points: d2 = 0 ties, more than k of them
    x[2000:5000] = x[2000] + 1e-4 * (x[2000:5000] - 0.5)    # a dense cluster: runs longer than the hit masks, LDS overflow
    idx = ctx.knn(x, 21, include_self=True)
    assert np.array_equal(idx, O.knn(x, 21, True, "kdtree", want_dist=False))
    y = wtp.synth.uniform(20000, 3, np.float32, 6)
    y[:, 2] *= 0.01                                          # a thin slab: two or three cells along z
    idx = ctx.knn(y, 21, include_self=False)
    assert np.array_equal(idx, O.knn(y, 21, False, "kdtree", want_dist=False))


def test_ksel_equals_brick_kernels(wtp):
    x = wtp.synth.uniform(300000, 3, np.float32, 11)
    outs = {}
    for flag in ("1", "0"):
        os.environ["WTP_KSEL"] = flag
        try:
            c = wtp.Context(0)
            outs[flag] = c.knn(x, 21, include_self=False, return_dist=True)
            c.close()
        finally:
            os.environ.pop("WTP_KSEL", None)
    assert np.array_equal(outs["1"][0], outs["0"][0]) and np.array_equal(outs["1"][1], outs["0"][1])


@pytest.mark.parametrize("kind,beta,u0,gamma,k,n_fixed", [(0, 0.2, 1.0, 3.0, 21, 0), (1, 0.2, 1.0, 3.0, 21, 3000),
                                                          (3, 0.2, 1.0, 3.0, 12, 0), (3, 0.3, 1.0, 2.0, 22, 0), (0, 0.2, 1.0, 3.0, 24, 0)])
def test_ksel_sweep_force_models(ctx, O, wtp, kind, beta, u0, gamma, k, n_fixed):
    n = 30000
    x = wtp.synth.uniform(n, 3, np.float32, 20260821)
    s, alo, amax = _sweep_args(n)
    with ctx.relax(x, n_fixed, s, dict(kind=kind, beta=beta, u0=u0, gamma=gamma), k, alo, amax) as sess:
        st = sess.step(True)
        p = sess.positions()
        pd = sess.point_data()
    ref = O.relax_sweep(x, n_fixed, s, kind, beta, u0, gamma, k, alo, amax)
    assert np.array_equal(pd["nn_id"], ref["nn_id"]) and np.array_equal(pd["nn_dist"], ref["nn_dist"])
    assert np.abs(p - ref["p"]).max() <= 2e-5 * s
    assert np.allclose(pd["forces"], ref["forces"], rtol=2e-4, atol=1e-6)
    assert st["n_move"] == n - n_fixed


def test_ksel_full_select_clipped_equals_default(wtp, O):
    # ClippedSpacingForce through the explicit k-selection (WTP_FULL_SELECT=1) and through the default sweep: same step
    n = 50000
    x = wtp.synth.uniform(n, 3, np.float32, 3)
    s, alo, amax = _sweep_args(n)
    res = {}
    for flag in ("1", "0"):
        os.environ["WTP_FULL_SELECT"] = flag
        try:
            c = wtp.Context(0)
            with c.relax(x, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, alo, amax) as sess:
                sess.step(True)
                res[flag] = (sess.positions(), sess.point_data())
            c.close()
        finally:
            os.environ.pop("WTP_FULL_SELECT", None)
    assert np.array_equal(res["1"][1]["nn_id"], res["0"][1]["nn_id"])
    assert np.abs(res["1"][0] - res["0"][0]).max() <= 2e-5 * s
