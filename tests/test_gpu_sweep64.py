"""Float64 sweeps of the k-nearest laws (InverseDistance, SpacingEquilibrium, LennardJones) through fp32 candidates
(csrc/wtp_sweep64.hip; 3-D, k <= 22, n >= 4096, fresh snapshot): exact re-ranking in fp64, forces added in ascending
(d2, index) — positions bit-identical to the exact wave-per-query path (WTP_F64_KSEL=0) and to the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _args(n):
    s = float(n) ** (-1.0 / 3.0)
    return s, s / 2000, s / 20


@pytest.mark.parametrize("kind,beta,u0,gamma,k,n_fixed", [(0, 0.2, 1.0, 3.0, 21, 0), (1, 0.2, 1.0, 3.0, 21, 2000),
                                                          (3, 0.2, 1.0, 3.0, 12, 0), (3, 0.3, 1.0, 2.0, 22, 0)])
def test_one_sweep_equals_the_exact_path_and_the_oracle(wtp, O, monkeypatch, kind, beta, u0, gamma, k, n_fixed):
    n = 20000
    x = wtp.synth.uniform(n, 3, np.float64, 20261004 + kind)
    s, alo, amax = _args(n)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("WTP_F64_KSEL", flag)
        with wtp.Context(0) as c:
            with c.relax(x, n_fixed, s, dict(kind=kind, beta=beta, u0=u0, gamma=gamma), k, alo, amax) as sess:
                st = sess.step(True)
                res[flag] = (sess.positions(), sess.point_data(), st)
    p, pd, st = res["1"]
    assert np.array_equal(p, res["0"][0]), "the same bits as the exact wave-per-query path"
    assert np.array_equal(pd["forces"], res["0"][1]["forces"]) and np.array_equal(pd["nn_dist"], res["0"][1]["nn_dist"])
    assert st["n_move"] == n - n_fixed
    assert st["n_fallback"] < n // 50, "nearly every query is certified from its fp32 candidates"
    ref = O.relax_sweep(x, n_fixed, s, kind, beta, u0, gamma, k, alo, amax)
    assert np.array_equal(pd["nn_id"], ref["nn_id"]) and np.array_equal(pd["nn_dist"], ref["nn_dist"])
    if kind == 3:  # LennardJones: pow() of the device library and of the host's differ in the last place (the exact path's rows too)
        assert np.abs(p - ref["p"]).max() <= 1e-12 * s
    else:
        assert np.array_equal(p, ref["p"]) and np.array_equal(pd["forces"], ref["forces"])


def test_ties_coincident_points_and_a_cluster(ctx, O, wtp):
    n = 12000
    x = wtp.synth.uniform(n, 3, np.float64, 9)
    x[100:110] = x[100]                                   # coincident points: zero distances, directions from the index pair
    x[2000:4000] = x[2000] + 1e-5 * (x[2000:4000] - 0.5)  # a cluster far finer than the float copy resolves: uncertified, exact path
    s, alo, amax = _args(n)
    with ctx.relax(x, 0, s, dict(kind=1, beta=0.2, u0=1.0, gamma=3.0), 21, alo, amax) as sess:
        sess.step(True)
        p = sess.positions()
    ref = O.relax_sweep(x, 0, s, 1, 0.2, 1.0, 3.0, 21, alo, amax)
    assert np.array_equal(p, ref["p"])


def test_several_iterations_equal_the_exact_path(wtp, monkeypatch):
    n = 60000
    x = wtp.synth.uniform(n, 3, np.float64, 4)
    s, alo, amax = _args(n)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("WTP_F64_KSEL", flag)
        with wtp.Context(0) as c:
            with c.relax(x, 3000, s, dict(kind=0, beta=0.2, u0=1.0, gamma=3.0), 21, alo, amax) as sess:
                conv, last = sess.run(6, 1)
                res[flag] = (sess.positions(), conv, last)
    assert np.array_equal(res["1"][0], res["0"][0])
    assert np.array_equal(np.asarray(res["1"][1]), np.asarray(res["0"][1]))
    assert res["1"][2]["sum_u"] == pytest.approx(res["0"][2]["sum_u"], rel=1e-12)
