"""Cost of the device-evaluated spacing law inside the sweep: repel iteration with ConstantSpacing
vs BoundaryLayerSpacing / LogLike over the box fixture's 46 786 boundary points."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd as w
z = np.load(os.path.join(ROOT, "tests", "golden", "box_surface.npz"))
b = (z["centroid"] / 25.0).astype(np.float32)            # unit cube walls
ctx = w.Context(0)
force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
for n in (1_000_000, 8_000_000):
    s = float(n) ** (-1.0 / 3.0)
    v = w.synth.uniform(n, 3, np.float32, 7) * np.float32(0.96) + np.float32(0.02)
    snap = np.concatenate([b, v])
    for name, sp in (("constant", s), ("loglike", w.LogLike(b, 1.3 * s, 1.2).desc()),
                     ("boundary_layer", w.BoundaryLayerSpacing(b, at_wall=0.8 * s, bulk=1.2 * s, layer_thickness=0.2).desc())):
        with ctx.relax(snap, len(b), sp, force, 21, s / 2000, s / 20) as t:
            t.run_async_free(5, 1)
            ctx.timers_reset()
            t0 = time.perf_counter()
            t.run_async_free(20, 1)
            dt = (time.perf_counter() - t0) / 20
            tm = ctx.timers()
            print(f"n={n:9d} {name:15s} {dt*1e3:8.3f} ms/iter  hash {tm['hash_ms']/20:.3f} sweep {tm['sweep_ms']/20:.3f} "
                  f"other(spacing law, fallbacks, reduce) {tm['other_ms']/20:.3f}", flush=True)
