"""One graded session (1 M points, BoundaryLayerSpacing on the device) for a rocprofv3 kernel trace."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd as w
ctx = w.Context(0)
force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
dt = np.float64 if os.environ.get("DT") == "f64" else np.float32
x = w.synth.graded(n, 4.0, 0.2, dt)
shell = (np.minimum(x, 1 - x).min(axis=1) < 0.02).sum()
hw = ((1 - 0.96 ** 3) / shell) ** (1 / 3)
m = max(int(1 / hw), 8)
g = (np.arange(m, dtype=dt) + 0.5) / m
u, v = np.meshgrid(g, g, indexing="ij")
faces = []
for axis in range(3):
    for side in (0.0, 1.0):
        c = np.zeros((m * m, 3), dt); c[:, axis] = side
        c[:, (axis + 1) % 3] = u.ravel(); c[:, (axis + 2) % 3] = v.ravel(); faces.append(c)
b = np.concatenate(faces)
law = w.BoundaryLayerSpacing(b, at_wall=hw, bulk=4 * hw, layer_thickness=0.2)
snap = np.concatenate([b, x])
with ctx.relax(snap, len(b), law.desc(), force, 21, hw / 2000, hw / 20) as t:
    t.step(True); t.run_async_free(3, 1)
    for stage in range(int(os.environ.get("STAGES", "1"))):   # later stages: the cloud has relaxed, points move less per sweep
        t0 = time.perf_counter()
        conv, st = t.run(20, 1)
        print("iterations", 4 + 120 * stage, "ms/iter", (time.perf_counter() - t0) / 20 * 1e3, "n_fallback", st["n_fallback"],
              "max_force", conv[-1], flush=True)
        if os.environ.get("WTP_DEBUG_KD"):
            import ctypes as C
            out = (C.c_ulonglong * 16)()
            w.load_library().wtp_debug_diag(ctx._h, out)
        t.run_async_free(100, 1)
