"""RadiusTopology on bench.py's graded cloud alone (10 M points, r = 2.5 h_wall; DT=f64 for Float64): device time per call.
For a kernel trace: rocprofv3 --kernel-trace -- python3 tools/exp_radius_graded.py, then tools/rocpd_stats.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import wtp_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ctx = wtp_amd.Context(0)
xg = bench.graded_dev(ctx, torch, np, wtp_amd, n).cpu().numpy()
shell = int((np.minimum(xg, 1 - xg).min(axis=1) < 0.02).sum())
hw = float(((1 - 0.96 ** 3) / shell) ** (1.0 / 3.0))
if os.environ.get("DT", "f32") == "f64":
    xg = xg.astype(np.float64)
for rep in range(3):
    ctx.timers_reset()
    t0 = time.perf_counter()
    off, idx = ctx.radius(xg, 2.5 * hw)
    wall = time.perf_counter() - t0
    tm = ctx.timers()
    print("n=%d %s pairs %d wall %.1f ms device %.3f ms (hash %.3f)" % (n, xg.dtype, int(off[-1]), wall * 1e3,
          tm["hash_ms"] + tm["sweep_ms"] + tm["other_ms"], tm["hash_ms"]), flush=True)
