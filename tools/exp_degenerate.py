"""The reference's own test configuration: all 46 786 face centres of box.stl (0.22 apart) as boundary, a
handful of volume points, spacing = bbox / 8 — a spacing far coarser than the cloud."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import wtp_amd
import oracle as O
z = np.load(os.path.join(ROOT, "tests", "golden", "box_surface.npz"))
cen = z["centroid"]
rng = np.random.default_rng(0)
vol = (rng.random((50, 3)) * 24 + 0.5).astype(np.float32)
snap = np.concatenate([cen, vol])
s = 25.0 / 8
ctx = wtp_amd.Context(0)
for label, nf in (("volume-only (boundary fixed)", len(cen)), ("all movable", 0)):
    with ctx.relax(snap, nf, s, dict(kind=2, beta=0.2, u0=1.0), 21, s / 2000, s / 20) as t:
        st = t.step(True)
        got = t.positions()
        t0 = time.perf_counter()
        for _ in range(10):
            st = t.step(True)
        dt = (time.perf_counter() - t0) / 10
    ref = O.relax_sweep(snap, nf, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s / 20)
    err = np.abs(got - ref["p"]).max() / s
    print(f"{label}: {dt*1e3:.2f} ms per iteration, hand-backs {st['n_fallback']}, first sweep vs oracle {err:.2e} spacings", flush=True)
ctx.close()
