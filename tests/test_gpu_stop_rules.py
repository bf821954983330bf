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
    # stall: the relative improvement of the CV per sweep falls below 0.1 % after a few hundred sweeps of a random cloud
    # (tools/exp_cv_trajectory.py); Float64 trajectories are bit-identical to the oracle's, so the rule fires in the same sweep
    cases = [dict(x=x, n_fixed=n_fixed, s=s, tol=1e-12, stall_after=2, cv_target=0.0, max_iters=2000),               # stall
             dict(x=x, n_fixed=n_fixed, s=s, tol=1e-12, stall_after=0, cv_target=0.5 * (cvs[5] + cvs[6]), max_iters=60),   # cv_target at iteration 7
             dict(x=x, n_fixed=n_fixed, s=s, tol=1e-12, stall_after=40, cv_target=0.5 * (cvs[8] + cvs[9]), max_iters=60)]  # both armed
    O.set_cv_double(False)
    for c in cases:
        x, n_fixed, s = c.pop("x"), c.pop("n_fixed"), c.pop("s")
        ref = O.relax_loop(x, n_fixed, s, 2, 0.2, 1.0, 3.0, k, s / 2000, s / 20, max_iters=c["max_iters"], tol=c["tol"],
                           rebuild_every=1, stall_after=c["stall_after"], cv_target=c["cv_target"])
        with ctx.relax(x, n_fixed, s, FORCE, k, s / 2000, s / 20) as t:
            conv, reason, _ = t.run_until(c["max_iters"], 1, c["tol"], c["stall_after"], c["cv_target"])
            p = t.positions()
        why = ref["stop_reason"]  # 0 max_iters, 1 tol, 2 cv_target, 3 stall
        assert reason == why and len(conv) == len(ref["conv"]), (c, reason, why, len(conv), len(ref["conv"]))
        assert reason in (2, 3), (c, reason, len(conv))
        assert np.array_equal(p, ref["p"]), "Float64: positions bit for bit, the reverted sweep of a cv_target stop included"
        assert np.array_equal(np.asarray(conv), ref["conv"])


def test_float32_stops_where_the_double_sum_oracle_stops(ctx, O, wtp):
    """Float32.  Two parts: (a) on the library's exact path (WTP_FORCE_GENERIC=1: sums in the reference's order, positions
    bit-identical to the oracle's sweep by sweep) the stall and cv_target rules fire in exactly the sweep where the
    oracle's loop ON DOUBLE SUMS fires them; (b) on the default fast path (coordinates equal to rounding only) the
    cv_target rule, whose margin in the first sweeps is wide, fires in the same sweep too.  Printed: how often the
    reference's own Float32-sum CV would have stopped elsewhere."""
    import os

    k = 21
    differs, total = 0, 0

    def both_oracles(x, n_fixed, s, c, max_iters):
        O.set_cv_double(True)
        rd = O.relax_loop(x, n_fixed, s, 2, 0.2, 1.0, 3.0, k, s / 2000, s / 20, max_iters=max_iters, tol=1e-12, rebuild_every=1,
                          stall_after=c["stall_after"], cv_target=c["cv_target"])
        O.set_cv_double(False)
        rf = O.relax_loop(x, n_fixed, s, 2, 0.2, 1.0, 3.0, k, s / 2000, s / 20, max_iters=max_iters, tol=1e-12, rebuild_every=1,
                          stall_after=c["stall_after"], cv_target=c["cv_target"])
        return rd, rf

    try:
        # (a) exact path
        os.environ["WTP_FORCE_GENERIC"] = "1"
        try:
            cx = wtp.Context(0)
        finally:
            os.environ.pop("WTP_FORCE_GENERIC", None)
        with cx:
            for seed, n, n_fixed in ((3, 4000, 0), (4, 5000, 300)):
                x = wtp.synth.uniform(n, 3, np.float32, seed)
                s = float(n) ** (-1.0 / 3.0)
                cvs = _cv_history(cx, x, n_fixed, s, k, 10)
                for c, mi in ((dict(stall_after=2, cv_target=0.0), 2500), (dict(stall_after=0, cv_target=0.5 * (cvs[5] + cvs[6])), 60)):
                    with cx.relax(x, n_fixed, s, FORCE, k, s / 2000, s / 20) as t:
                        conv, reason, _ = t.run_until(mi, 1, 1e-12, c["stall_after"], c["cv_target"])
                        p = t.positions()
                    rd, rf = both_oracles(x, n_fixed, s, c, mi)
                    assert reason == rd["stop_reason"] and len(conv) == len(rd["conv"]), (n, c, reason, rd["stop_reason"], len(conv), len(rd["conv"]))
                    assert reason in (2, 3), (n, c, reason, len(conv))
                    assert np.array_equal(p, rd["p"]), "exact path: positions bit for bit"
                    total += 1
                    differs += int(len(rf["conv"]) != len(rd["conv"]) or rf["stop_reason"] != rd["stop_reason"])
        # (b) default fast path, cv_target
        for seed, n, n_fixed in ((5, 20000, 0), (6, 30000, 1500)):
            x = wtp.synth.uniform(n, 3, np.float32, seed)
            s = float(n) ** (-1.0 / 3.0)
            cvs = _cv_history(ctx, x, n_fixed, s, k, 10)
            c = dict(stall_after=50, cv_target=0.5 * (cvs[6] + cvs[7]))
            with ctx.relax(x, n_fixed, s, FORCE, k, s / 2000, s / 20) as t:
                conv, reason, _ = t.run_until(60, 1, 1e-12, c["stall_after"], c["cv_target"])
                p = t.positions()
            rd, rf = both_oracles(x, n_fixed, s, c, 60)
            assert reason == rd["stop_reason"] == 2 and len(conv) == len(rd["conv"]) == 8
            assert np.abs(p - rd["p"]).max() / s <= 1e-4
            total += 1
            differs += int(len(rf["conv"]) != len(rd["conv"]) or rf["stop_reason"] != rd["stop_reason"])
    finally:
        O.set_cv_double(False)
    print(f"[stop rules, Float32] the reference's Float32-sum CV stops elsewhere than the double-sum CV in {differs} of {total} runs")
