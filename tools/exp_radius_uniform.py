"""RadiusTopology on a uniform cloud at a radius holding ~21 neighbours (the bench's radius leg): device time per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wtp_amd as w
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
mult = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ctx = w.Context(0)
x = w.synth.uniform(n, 3, np.float32, 7)
r = mult * (21.0 / (4.0 / 3.0 * np.pi * n)) ** (1.0 / 3.0)
ctx.radius(x[:100000], r)
for rep in range(3):
    ctx.timers_reset(); t0 = time.perf_counter()
    off, idx = ctx.radius(x, r)
    dtt = time.perf_counter() - t0; tm = ctx.timers()
    print(f"n={n} r={mult} r21 pairs/pt {off[-1]/n:.1f} wall {dtt*1e3:.2f} ms, device {tm['hash_ms']+tm['sweep_ms']+tm['other_ms']:.3f} ms", flush=True)
