"""Float64 repel (ClippedSpacingForce, uniform cloud): ms per iteration, for kernel traces."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wtp_amd as w
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
ctx = w.Context(0)
s = float(n) ** (-1.0 / 3.0)
x = torch.empty((n, 3), dtype=torch.float64, device="cuda")
ctx.gen_uniform_dev(w.synth.SEED, 0, n, 3, np.float64, x.data_ptr())
with ctx.relax(None, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20, device_ptr=(x.data_ptr(), n, 3, np.float64)) as t:
    t.run_async_free(5, 1); torch.cuda.synchronize(); t0 = time.perf_counter()
    t.run_async_free(20, 1); torch.cuda.synchronize()
    print(f"Float64 repel n={n}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per iteration")
