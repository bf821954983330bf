"""What the compact-support sweep hands back on the benchmark cloud: count vs the number of points whose
nearest neighbour lies beyond the ring margin."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
ctx = wtp_amd.Context(0)
n = 4_000_000
x = wtp_amd.synth.uniform(n, 3, np.float32)
s = float(n) ** (-1.0 / 3.0)
with ctx.relax(x, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20) as t:
    for it in range(3):
        st = t.step(True)
        u = t.point_data()["nn_dist"] / s
        print(f"iter {it}: hand-backs {st['n_fallback']} ({st['n_fallback']/n:.2e}); nn/s > 1.0: {(u>1.0).sum()}, > 1.1: {(u>1.1).sum()}, "
              f"> 1.2: {(u>1.2).sum()}, > 1.25: {(u>1.25).sum()}", flush=True)
ctx.close()
