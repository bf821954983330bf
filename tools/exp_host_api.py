"""Host-array API cost breakdown: wtp_knn / wtp_radius on 1 M points (what a set_topology call pays)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
ctx = wtp_amd.Context(0)
n = 1_000_000
x = wtp_amd.synth.uniform(n, 3, np.float32)
ctx.knn(x[:10000], 21)
for rep in range(3):
    ctx.timers_reset()
    t0 = time.perf_counter()
    idx = ctx.knn(x, 21)
    dt = time.perf_counter() - t0
    tm = ctx.timers()
    print(f"knn 1M k=21 host arrays: {dt*1e3:.2f} ms total, device {tm['hash_ms']+tm['sweep_ms']+tm['other_ms']:.2f} ms, rows {idx.nbytes/1e6:.0f} MB", flush=True)
# reuse of an already-touched output buffer (page faults excluded)
import ctypes as C
L = wtp_amd.load_library()
out = np.empty((n, 21), dtype=np.int32); out[:] = 0
for rep in range(3):
    t0 = time.perf_counter()
    rc = L.wtp_knn(ctx._h, x.ctypes.data_as(C.c_void_p), n, 3, 0, 21, 0, out.ctypes.data_as(C.c_void_p), None)
    print(f"  into a pre-touched buffer: {(time.perf_counter()-t0)*1e3:.2f} ms (rc {rc})", flush=True)
ctx.close()
