"""Paths that run on the exact wave kernel for every query: a Float64 sweep with a non-default law, a fp32 sweep against a
stale snapshot (rebuild_every = 2).  ms per iteration."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wtp_amd as w
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ctx = w.Context(0)
s = float(n) ** (-1.0 / 3.0)
for dt, kind, re, label in ((np.float64, 1, 1, "Float64, SpacingEquilibriumForce"), (np.float32, 2, 2, "fp32, default law, rebuild_every=2")):
    x = torch.empty((n, 3), dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda")
    ctx.gen_uniform_dev(w.synth.SEED, 0, n, 3, dt, x.data_ptr())
    with ctx.relax(None, 0, s, dict(kind=kind, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20, device_ptr=(x.data_ptr(), n, 3, dt)) as t:
        t.run_async_free(4, re); torch.cuda.synchronize(); t0 = time.perf_counter()
        t.run_async_free(10, re); torch.cuda.synchronize()
        print(f"{label}, n={n}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per iteration", flush=True)
    del x
