import sys, math, numpy as np
sys.path.insert(0, "/root/repo")
import wtp_amd as w
ctx = w.Context(0)
F = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
for dim, n, nf, dt in ((2, 6000, 400, np.float64), (3, 6000, 400, np.float64), (3, 20000, 0, np.float32)):
    x = w.synth.uniform(n, dim, dt, 5)
    s = float(n) ** (-1.0 / dim)
    cv = []
    with ctx.relax(x, nf, s, F, 21, s / 2000, s / 20) as t:
        for i in range(400):
            st = t.step(True)
            mu = st["sum_u"] / st["n_move"]
            cv.append(math.sqrt(max(st["sum_u2"] / st["n_move"] - mu * mu, 0)) / mu)
    best, last = float("inf"), 0
    first_noimp = None
    for i, c in enumerate(cv, 1):
        if c < best * (1 - 1e-3):
            best, last = c, i
        elif first_noimp is None:
            first_noimp = i
    print(dim, n, "cv[0,10,50,100,200,399]", [round(cv[i], 5) for i in (0, 10, 50, 100, 200, 399)], "first iteration without 0.1% improvement:", first_noimp, "last improvement at", last)
