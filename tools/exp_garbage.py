#!/usr/bin/env python3
"""Robustness probe: hand libwtp hostile coordinates (NaN, inf, huge, all-equal) and make sure every
call returns (no hang, no fault).  Run under `timeout`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, wtp_amd
ctx = wtp_amd.Context(0)
n = 200000
rng = np.random.default_rng(0)
base = rng.random((n, 3)).astype(np.float32)
cases = {}
x = base.copy(); x[::1000] = np.nan; cases["some_nan"] = x
x = base.copy(); x[::777, 1] = np.inf; cases["some_inf"] = x
x = base.copy(); x[5] = 1e30; x[6] = -1e30; cases["huge_outliers"] = x
cases["all_equal"] = np.full((n, 3), 0.25, np.float32)
x = base.copy(); x[:, 2] = 0.5; cases["planar_in_3d"] = x
cases["all_nan"] = np.full((5000, 3), np.nan, np.float32)
x = rng.random((n, 3)).astype(np.float32) * 1e-30; cases["denormal_scale"] = x
for name, x in cases.items():
    s = 0.02
    try:
        idx = ctx.knn(x, 21, include_self=False)
        with ctx.relax(x, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20) as sess:
            st = sess.step(True); st = sess.step(True)
        with ctx.relax(x, 0, s, dict(kind=0, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20) as sess:
            st2 = sess.step(True)
        off, ridx = ctx.radius(x[:20000], 0.01)
        print(name, "ok", "fallback", st["n_fallback"], st2["n_fallback"], "nnz", int(off[-1]), flush=True)
    except Exception as e:
        print(name, "EXC", type(e).__name__, str(e)[:100], flush=True)
print("done", flush=True)
