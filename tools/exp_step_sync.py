"""Per-iteration host synchronisation: sess.step() in a loop (what the host stop rules need) against
sess.run() (no read-back until the end)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
ctx = wtp_amd.Context(0)
for n in (50_000, 200_000, 1_000_000, 10_000_000):
    x = wtp_amd.synth.uniform(n, 3, np.float32)
    s = float(n) ** (-1.0 / 3.0)
    with ctx.relax(x, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20) as t:
        t.run(5, 1)
        it = 200 if n <= 1_000_000 else 40
        t0 = time.perf_counter(); t.run(it, 1); a = (time.perf_counter() - t0) / it
        t0 = time.perf_counter()
        for _ in range(it):
            t.step(True)
        b = (time.perf_counter() - t0) / it
    print(f"n={n:9d}: run() {a*1e3:.3f} ms/iter   step() loop {b*1e3:.3f} ms/iter   (+{(b-a)*1e6:.0f} us per iteration)", flush=True)
ctx.close()
