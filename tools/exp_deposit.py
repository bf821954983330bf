"""Octree method with deposition at scale: a sparse boundary on the box surface, N volume points, the
host-serial deposit pass (projections and site k-NN lists from the device)."""
import os, sys, time, logging
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
z = np.load(os.path.join(ROOT, "tests", "golden", "box_mesh.npz"))
v, t = z["vertices"], z["triangles"]
tri = v[t].astype(np.float64)
cen = tri.mean(axis=1).astype(np.float32)
cr = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
area = (0.5 * np.linalg.norm(cr, axis=1)).astype(np.float32)
nrm = (cr / np.maximum(np.linalg.norm(cr, axis=1), 1e-300)[:, None]).astype(np.float32)
ctx = wtp_amd.Context(0)
oc = wtp_amd.TriangleOctree(v, t, ctx=ctx)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sel = np.arange(0, len(cen), 200)
vol = (wtp_amd.synth.uniform(n, 3, np.float32, 5) * 24.9 + 0.05).astype(np.float32)
s = 25.0 / n ** (1 / 3)
cloud = wtp_amd.PointCloud(wtp_amd.PointBoundary(cen[sel], nrm[sel], area[sel]), wtp_amd.PointVolume(vol))
for iters in (1, 6):
    t0 = time.perf_counter()
    out = wtp_amd.repel(cloud, wtp_amd.ConstantSpacing(s), oc, max_iters=iters, deposit_ratio=0.5, stall_after=0, ctx=ctx)
    dt = time.perf_counter() - t0
    print(f"n={n} iters={iters}: {dt:.2f} s, boundary {len(cloud.boundary)} -> {len(out.boundary)}, "
          f"volume inside: {bool(oc.isinside(out.volume.points()).all())}", flush=True)
ctx.close()
