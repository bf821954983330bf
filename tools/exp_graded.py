"""Graded cloud (C5-like): repel sweep with the matching BoundaryLayerSpacing law, and k-NN
topology — how the uniform hash copes with a 64x density contrast."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd as w
ctx = w.Context(0)
force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
for n in (1_000_000, 4_000_000):
    x = w.synth.graded(n, 4.0, 0.2, np.float32)
    # wall spacing from the density near the wall: count in the outer 2% shell
    shell = (np.minimum(x, 1 - x).min(axis=1) < 0.02).sum()
    vol = 1 - 0.96 ** 3
    hw = (vol / shell) ** (1 / 3)
    # boundary points for the law: a lattice on the six faces at spacing hw
    m = max(int(1 / hw), 8)
    g = (np.arange(m, dtype=np.float32) + 0.5) / m
    u, v = np.meshgrid(g, g, indexing="ij")
    faces = []
    for axis in range(3):
        for side in (0.0, 1.0):
            c = np.zeros((m * m, 3), np.float32); c[:, axis] = side
            c[:, (axis + 1) % 3] = u.ravel(); c[:, (axis + 2) % 3] = v.ravel(); faces.append(c)
    b = np.concatenate(faces)
    law = w.BoundaryLayerSpacing(b, at_wall=hw, bulk=4 * hw, layer_thickness=0.2)
    snap = np.concatenate([b, x])
    for name, sp in (("law", law.desc()), ("const(h_wall)", float(hw))):
        with ctx.relax(snap, len(b), sp, force, 21, hw / 2000, hw / 20) as t:
            st = t.step(True); t.run_async_free(3, 1)
            ctx.timers_reset(); t0 = time.perf_counter()
            conv, st = t.run(10, 1)
            dt = (time.perf_counter() - t0) / 10; tm = ctx.timers()
            print(f"n={n} boundary={len(b)} hw={hw:.4f} spacing={name:14s}: {dt*1e3:8.2f} ms/iter  hash {tm['hash_ms']/10:.2f} "
                  f"sweep {tm['sweep_ms']/10:.2f} other {tm['other_ms']/10:.2f}  n_fallback {st['n_fallback']}", flush=True)
    t0 = time.perf_counter(); idx = ctx.knn(x, 21); dt = time.perf_counter() - t0
    print(f"n={n} knn k=21 (host arrays): {dt*1e3:.1f} ms", flush=True)
