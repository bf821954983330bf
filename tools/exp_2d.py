"""2-D clouds: sweep and k-NN throughput (bricks are 4x4x4 cells: a 2-D grid fills one layer of them)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd as w
ctx = w.Context(0)
force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
for dtype in (np.float32, np.float64):
    for n in (1_000_000, 4_000_000):
        x = w.synth.uniform(n, 2, dtype, 7)
        s = float(n) ** (-0.5)
        with ctx.relax(x, 0, s, force, 21, s / 2000, s / 20) as t:
            t.run_async_free(3, 1)
            ctx.timers_reset()
            t0 = time.perf_counter()
            conv, st = t.run(10, 1)
            dt = (time.perf_counter() - t0) / 10
            tm = ctx.timers()
        print(f"2-D {np.dtype(dtype).name} n={n}: {dt*1e3:7.3f} ms/iter {n/dt/1e6:8.1f} Mpts/s  hash {tm['hash_ms']/10:.3f} sweep {tm['sweep_ms']/10:.3f} "
              f"other {tm['other_ms']/10:.3f} fallback {st['n_fallback']}", flush=True)
    if dtype == np.float32:
        x = w.synth.uniform(1_000_000, 2, dtype, 7)
        ctx.knn(x[:1000], 21)
        t0 = time.perf_counter(); ctx.knn(x, 21); print(f"2-D knn 1M host arrays: {(time.perf_counter()-t0)*1e3:.2f} ms")
