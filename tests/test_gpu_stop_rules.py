"""Stop rules of `_relax!` (src/repel.jl:305-334; the reference's own checks: test/repel.jl:262-299) — the device loop
(wtp_relax_run_until) against the ORACLE's loop, with the stall and cv_target rules switched on.

What is compared, and why it is split by element type:
  * Float64 clouds: the reference's `_dnn_cv` sums u and u^2 serially in T = Float64; the device sums the same values in
    double (block partials, fixed order).  Same iteration count, same reason, positions bit for bit.
  * Float32 clouds: the reference sums in Float32 (src/repel.jl:374-386), the device in double.  A serial Float32 sum over
    10^4..10^7 values carries 1e-4..1e-3 relative error in the variance — as large as the stall rule's 1e-3 margin — so
    the reference can stop at another iteration than the device.  The device value is the better number; what is
    asserted is equality with the oracle's loop evaluated ON DOUBLE SUMS (oracle.set_cv_double), and the test prints
    how often the Float32-sum loop differs from it."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORCE = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)


def _cv_history(ctx, x, n_fixed, s, k, iters):
    out = []
    with ctx.relax(x, n_fixed, s, FORCE, k, s / 2000, s / 20) as t:
        for _ in range(iters):
            st = t.step(True)
            mu = st["sum_u"] / st["n_move"]
            out.append(math.sqrt(max(st["sum_u2"] / st["n_move"] - mu * mu, 0.0)) / mu)
    return out


@pytest.mark.parametrize("dim", [2, 3])
def test_float64_stall_and_cv_target_stop_where_the_oracle_loop_stops(ctx, O, wtp, dim):
    n, k, n_fixed = 6000, 21, 400
    x = wtp.synth.uniform(n, dim, np.float64, 5)
    s = float(n) ** (-1.0 / dim)
    cvs = _cv_history(ctx, x, n_fixed, s, k, 12)
    assert cvs[7] < cvs[5] < cvs[3], "the cloud relaxes: the CV of d_NN / s falls"
    cases = [dict(tol=1e-12, stall_after=2, cv_target=0.0, max_iters=60),                       # stall
             dict(tol=1e-12, stall_after=0, cv_target=0.5 * (cvs[5] + cvs[6]), max_iters=60),   # cv_target at iteration 7
             dict(tol=1e-12, stall_after=4, cv_target=0.5 * (cvs[8] + cvs[9]), max_iters=60)]   # both armed
    O.set_cv_double(False)
    for c in cases:
        ref = O.relax_loop(x, n_fixed, s, 2, 0.2, 1.0, 3.0, k, s / 2000, s / 20, max_iters=c["max_iters"], tol=c["tol"],
                           rebuild_every=1, stall_after=c["stall_after"], cv_target=c["cv_target"])
        with ctx.relax(x, n_fixed, s, FORCE, k, s / 2000, s / 20) as t:
            conv, reason, _ = t.run_until(c["max_iters"], 1, c["tol"], c["stall_after"], c["cv_target"])
            p = t.positions()
        why = ("max_iters", "tol", "cv_target", "stall")[ref["stop_reason"]]
        assert reason == why and len(conv) == len(ref["conv"]), (c, reason, why, len(conv), len(ref["conv"]))
        assert reason in ("stall", "cv_target")
        assert np.array_equal(p, ref["p"]), "Float64: positions bit for bit, the reverted sweep of a cv_target stop included"
        assert np.array_equal(np.asarray(conv), ref["conv"])


def test_float32_stops_where_the_double_sum_oracle_stops(ctx, O, wtp):
    k = 21
    differs, total = 0, 0
    try:
        for seed, n, n_fixed in ((3, 20000, 0), (4, 30000, 1500), (5, 12000, 300), (6, 40000, 0)):
            x = wtp.synth.uniform(n, 3, np.float32, seed)
            s = float(n) ** (-1.0 / 3.0)
            cvs = _cv_history(ctx, x, n_fixed, s, k, 10)
            for c in (dict(stall_after=2, cv_target=0.0), dict(stall_after=0, cv_target=0.5 * (cvs[5] + cvs[6])),
                      dict(stall_after=5, cv_target=0.5 * (cvs[7] + cvs[8]))):
                with ctx.relax(x, n_fixed, s, FORCE, k, s / 2000, s / 20) as t:
                    conv, reason, _ = t.run_until(80, 1, 1e-12, c["stall_after"], c["cv_target"])
                    p = t.positions()
                O.set_cv_double(True)
                rd = O.relax_loop(x, n_fixed, s, 2, 0.2, 1.0, 3.0, k, s / 2000, s / 20, max_iters=80, tol=1e-12, rebuild_every=1,
                                  stall_after=c["stall_after"], cv_target=c["cv_target"])
                O.set_cv_double(False)
                rf = O.relax_loop(x, n_fixed, s, 2, 0.2, 1.0, 3.0, k, s / 2000, s / 20, max_iters=80, tol=1e-12, rebuild_every=1,
                                  stall_after=c["stall_after"], cv_target=c["cv_target"])
                why = ("max_iters", "tol", "cv_target", "stall")[rd["stop_reason"]]
                assert reason == why and len(conv) == len(rd["conv"]), (seed, c, reason, why, len(conv), len(rd["conv"]))
                # fast path: coordinates to rounding of the summation order, accumulated over the iterations run
                assert np.abs(p - rd["p"]).max() / s <= 1e-4
                total += 1
                differs += int(len(rf["conv"]) != len(rd["conv"]) or rf["stop_reason"] != rd["stop_reason"])
    finally:
        O.set_cv_double(False)
    print(f"[stop rules, Float32] the reference's Float32-sum CV stops elsewhere than the double-sum CV in {differs} of {total} runs")
