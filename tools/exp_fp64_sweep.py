"""fp64 repel sweep throughput (exact wave-per-query path) and graded fp32 sweep."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd as w
ctx = w.Context(0)
force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
for dtype, n in ((np.float64, 1_000_000), (np.float64, 4_000_000), (np.float32, 1_000_000)):
    s = float(n) ** (-1.0 / 3.0)
    x = w.synth.uniform(n, 3, dtype, 7)
    for fk in (2, 1):
        os.environ.pop("WTP_FORCE_GENERIC", None)
        c = ctx
        if dtype == np.float32:
            os.environ["WTP_FORCE_GENERIC"] = "1"    # fp32 through the wave kernel too, for comparison
            c = w.Context(0)
        with c.relax(x, 0, s, dict(kind=fk, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20) as t:
            t.run_async_free(3, 1)
            t0 = time.perf_counter()
            t.run_async_free(10, 1)
            dt = (time.perf_counter() - t0) / 10
        print(f"{np.dtype(dtype).name} n={n} force_kind={fk} (wave path): {dt*1e3:8.2f} ms/iter  {n/dt/1e6:8.1f} Mpts/s  "
              f"{dt/n*1e9:.2f} ns/query", flush=True)
os.environ.pop("WTP_FORCE_GENERIC", None)
