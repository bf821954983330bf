"""Cost of the octree wall rule (wtp_relax_set_wall) per repel iteration: box fixture (46 786
triangles, boundary = its face centroids) + N volume points, with and without the wall."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
z = np.load(os.path.join(ROOT, "tests", "golden", "box_mesh.npz"))
v, t = z["vertices"], z["triangles"]
if os.environ.get("WALL_F64"):
    v = v.astype(np.float64)   # a Float64 index under Float32 points (the reference's usual pairing)
cen = v[t].mean(axis=1).astype(np.float32)
ctx = wtp_amd.Context(0)
oc = wtp_amd.TriangleOctree(v, t, ctx=ctx)
t0 = time.perf_counter(); oc._resident(ctx); print(f"mesh upload + host build {1e3*(time.perf_counter()-t0):.1f} ms")
for n in [int(a) for a in (sys.argv[1:] or ["1000000", "10000000"])]:
    vol = (wtp_amd.synth.uniform(n, 3, np.float32, 5) * 24.8 + 0.1).astype(np.float32)
    s = 25.0 / n ** (1 / 3)
    snap = np.concatenate([cen, vol])
    for wall in (False, True):
        sess = ctx.relax(snap, 0, s, dict(kind=2, beta=0.2, u0=1.0), 21, s / 2000, s / 20)
        if wall:
            sess.set_wall(len(cen), 1e-6 * 43.3)
        sess.step(True)
        ctx.timers_reset()
        t0 = time.perf_counter()
        sess.run_async_free(20, 1)
        st = sess.step(True)
        dt = (time.perf_counter() - t0) / 21
        tm = ctx.timers()
        if wall and os.environ.get("WTP_LIB"):
            import ctypes
            c = (ctypes.c_ulonglong * 4)()
            ctypes.CDLL(os.environ["WTP_LIB"]).wtp_mesh_count(c)
            print(f"   traversal: {c[0]/max(c[2],1):.0f} node steps and {c[1]/max(c[2],1):.0f} evaluated nodes per searching wave, "
                  f"{c[2]/22:.0f} waves/iter, longest walk {c[3]} steps")
        print(f"n={n:9d} wall={wall}: {dt*1e3:7.3f} ms/iter  hash {tm['hash_ms']/21:.3f} sweep {tm['sweep_ms']/21:.3f} "
              f"other {tm['other_ms']/21:.3f}  escaped {st['n_escaped']}", flush=True)
        sess.close()
# standalone queries
for n in (1_000_000,):
    pts = (wtp_amd.synth.uniform(n, 3, np.float32, 7) * 30 - 2.5).astype(np.float32)
    oc.isinside(pts[:1000])
    ctx.timers_reset()
    ins = oc.isinside(pts)
    print(f"isinside {n} random points: device {ctx.timers()['other_ms']:.2f} ms, inside {ins.mean():.3f}")
    srt = pts[np.lexsort((pts[:, 0] // 1, pts[:, 1] // 1, pts[:, 2] // 1))]
    ctx.timers_reset()
    oc.isinside(srt)
    print(f"isinside {n} cell-sorted points: device {ctx.timers()['other_ms']:.2f} ms")
ctx.close()
