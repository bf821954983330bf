"""fp64 repel trajectories, GPU vs oracle, many iterations: both evaluate every force term with IEEE
division / sqrt and add them in ascending (d2, index) order, so positions must agree bit for bit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
import wtp_amd as w
ctx = w.Context(0)
for (n, dim, nf, iters, kind) in ((60000, 3, 3000, 40, 2), (20000, 2, 0, 60, 2), (30000, 3, 0, 25, 1), (30000, 3, 2000, 25, 3)):
    x = w.synth.uniform(n, dim, np.float64, 77)
    s = float(n) ** (-1.0 / dim)
    force = dict(kind=kind, beta=0.2, u0=1.0, gamma=3.0)
    t0 = time.time()
    with ctx.relax(x, nf, s, force, 21, s / 2000, s / 20) as t:
        conv, st = t.run(iters, 1)
        got = t.positions()
    ref = O.relax_loop(x, nf, s, kind, 0.2, 1.0, 3.0, 21, s / 2000, s / 20, max_iters=iters, tol=0.0, rebuild_every=1, stall_after=0)
    same = np.array_equal(got, ref["p"])
    print(f"n={n} dim={dim} n_fixed={nf} law={kind} iters={iters}: positions identical={same} conv identical={np.array_equal(conv, ref['conv'])} "
          f"max|diff|/s={np.abs(got-ref['p']).max()/s:.2e}  ({time.time()-t0:.1f} s)", flush=True)
