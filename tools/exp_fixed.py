import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, wtp_amd
n = 2_400_000
ctx = wtp_amd.Context(0)
x = wtp_amd.synth.uniform(n, 3, np.float32)
x[:, 2] *= 0.6   # slab-like box
s = (0.6 / n) ** (1 / 3)
for nf, order in ((0, "orig"), (400000, "orig"), (400000, "ghost_slab")):
    y = x.copy()
    if order == "ghost_slab":
        # ghosts = the points with z in the outer 0.05 layers, moved to the head (like sharded.py)
        m = (y[:, 2] < 0.05) | (y[:, 2] > 0.55)
        y = np.concatenate([y[m], y[~m]]); nf = int(m.sum())
    sess = ctx.relax(y, nf, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20)
    st = sess.step(True); ctx.timers_reset()
    t0 = time.perf_counter(); st = sess.step(True); dt = time.perf_counter() - t0
    print(order, nf, "ms", round(dt * 1e3, 2), "fb", st["n_fallback"], ctx.timers())
    sess.close()
