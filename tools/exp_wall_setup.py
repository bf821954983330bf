"""One-off costs of the octree method's setup: mesh upload + host build, class grid."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
z = np.load(os.path.join(ROOT, "tests", "golden", "box_mesh.npz"))
v, t = z["vertices"], z["triangles"]
cen = v[t].mean(axis=1).astype(np.float32)
ctx = wtp_amd.Context(0)
oc = wtp_amd.TriangleOctree(v, t, ctx=ctx)
oc._resident(ctx)
for n in (1_000_000, 10_000_000):
    vol = (wtp_amd.synth.uniform(n, 3, np.float32, 5) * 24.8 + 0.1).astype(np.float32)
    s = 25.0 / n ** (1 / 3)
    sess = ctx.relax(np.concatenate([cen, vol]), 0, s, dict(kind=2, beta=0.2, u0=1.0), 21, s / 2000, s / 20)
    ctx.mesh_set(v, t)  # drop the class grid
    ctx._mesh_owner = oc._token
    t0 = time.perf_counter()
    sess.set_wall(len(cen), 4e-5)
    sess.step(True)
    print(f"n={n}: set_wall + first step {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
    t0 = time.perf_counter()
    sess.step(True)
    print(f"        second step {1e3*(time.perf_counter()-t0):.2f} ms", flush=True)
    sess.close()
ctx.close()
