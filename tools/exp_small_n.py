"""Step time vs cloud size (launch-bound regime): wtp_relax_run without read-back."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
ctx = wtp_amd.Context(0)
force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
timers = os.environ.get("WTP_TIMING", "1") != "0"
for n in (2000, 10000, 46786, 200000, 1000000, 4000000):
    s = float(n) ** (-1.0 / 3.0)
    x = wtp_amd.synth.uniform(n, 3, np.float32, 7)
    with ctx.relax(x, 0, s, force, 21, s / 2000, s / 20) as t:
        t.run_async_free(20, 1)
        if timers:
            ctx.timers_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = int(os.environ.get("ITERS", "200"))
        t.run_async_free(iters, 1)
        t_enq = (time.perf_counter() - t0) / iters
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        tm = ctx.timers() if timers else dict(hash_ms=0.0, sweep_ms=0.0, other_ms=0.0)
        dev = (tm["hash_ms"] + tm["sweep_ms"] + tm["other_ms"]) / iters
        print(f"n={n:8d}  wall {dt*1e6:8.1f} us/step   device spans {dev*1e3:8.1f} us/step   {n/dt/1e6:8.1f} Mpts/s   host enqueue {t_enq*1e6:6.1f} us/step", flush=True)
