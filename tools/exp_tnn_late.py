"""Does a smaller nearest-neighbour margin (WTP_TNN) pay once the cloud has relaxed?  Step time and
hand-backs early (iterations 1-50) and late (after 600 iterations) for one margin per process."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
ctx = wtp_amd.Context(0)
x = wtp_amd.synth.uniform(n, 3, np.float32)
s = float(n) ** (-1.0 / 3.0)
with ctx.relax(x, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20) as t:
    t.run(5, 1)
    t0 = time.perf_counter(); _, st = t.run(50, 1); early = (time.perf_counter() - t0) / 50
    fb_early = st["n_fallback"]
    t.run(600, 1)
    t0 = time.perf_counter(); _, st = t.run(50, 1); late = (time.perf_counter() - t0) / 50
    pd = t.point_data()
    u = pd["nn_dist"] / s
print(f"WTP_TNN={os.environ.get('WTP_TNN', '0.8')} n={n}: early {early*1e3:.3f} ms/iter (hand-backs {fb_early}), late {late*1e3:.3f} ms/iter "
      f"(hand-backs {st['n_fallback']}); late nn/s: mean {u.mean():.3f} p99.9 {np.quantile(u, 0.999):.3f} max {u.max():.3f}", flush=True)
ctx.close()
