"""Graded RadiusTopology (BASELINE config 5): device time of wtp_radius_offsets + wtp_radius_fill on the 64x-graded cloud,
for a kernel trace (rocprofv3 --kernel-trace --stats -- python3 tools/exp_radius_graded.py [n])."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wtp_amd as w
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
dt = np.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else np.float32
ctx = w.Context(0)
x = w.synth.graded(n, 4.0, 0.2, dt)
shell = (np.minimum(x, 1 - x).min(axis=1) < 0.02).sum()
hw = float(((1 - 0.96 ** 3) / shell) ** (1 / 3))
ctx.radius(x[:100000], 2.5 * hw)
for rep in range(3):
    ctx.timers_reset(); t0 = time.perf_counter()
    off, idx = ctx.radius(x, 2.5 * hw)
    dtt = time.perf_counter() - t0; tm = ctx.timers()
    print(f"n={n} {dt.__name__} pairs={int(off[-1])} wall {dtt*1e3:.2f} ms, device {tm['hash_ms']+tm['sweep_ms']+tm['other_ms']:.3f} ms "
          f"(hash {tm['hash_ms']:.3f}, brick {tm['sweep_ms']:.3f}, rest {tm['other_ms']:.3f}); max row {int(np.diff(off).max())}, mean {off[-1]/n:.1f}", flush=True)
