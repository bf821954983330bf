"""One small cloud (the reference's own test size, 46 786 points), 300 repel iterations enqueued back to back: for kernel traces."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wtp_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 46786
ctx = wtp_amd.Context(0)
s = float(n) ** (-1.0 / 3.0)
x = wtp_amd.synth.uniform(n, 3, np.float32, 7)
with ctx.relax(x, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20) as t:
    t.run_async_free(20, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t.run_async_free(300, 1)
    torch.cuda.synchronize()
    print(f"n={n}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per iteration")
