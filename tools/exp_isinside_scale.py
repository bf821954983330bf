"""Throughput of the Green's-function isinside filter (wtp_isinside_greens): N test points x the
46 786 boundary elements of the box fixture."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
z = np.load(os.path.join(ROOT, "tests", "golden", "box_surface.npz"))
c, nrm, a = z["centroid"], z["normal"], z["area"]
ctx = wtp_amd.Context(0)
for dtype in (np.float32, np.float64):
    for n in (1000, 100_000, 2_000_000):
        t = (wtp_amd.synth.uniform(n, 3, np.float32, 3) * 30 - 2.5).astype(dtype)
        ctx.isinside_greens(t[:1000], c, nrm, a)
        ctx.timers_reset()
        t0 = time.perf_counter()
        ins = ctx.isinside_greens(t, c, nrm, a)
        dt = time.perf_counter() - t0
        dev = ctx.timers()["other_ms"] * 1e-3
        pairs = n * len(c)
        print(f"{np.dtype(dtype).name} n={n:8d}: wall {dt*1e3:9.2f} ms  device {dev*1e3:9.2f} ms  "
              f"{pairs/dev/1e12:6.3f} Tpairs/s  ({pairs/dev*16/1e12:5.1f} T lane-ops/s at 16/pair)  inside {ins.mean():.3f}",
              flush=True)
