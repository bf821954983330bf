import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
ctx = wtp_amd.Context(0)
n = 46786
s = float(n) ** (-1.0 / 3.0)
x = wtp_amd.synth.uniform(n, 3, np.float32, 7)
with ctx.relax(x, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20) as t:
    t.run_async_free(20, 1)
    t0 = time.perf_counter()
    t.run_async_free(200, 1)
    print("us/step", (time.perf_counter() - t0) / 200 * 1e6)
