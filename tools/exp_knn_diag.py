#!/usr/bin/env python3
"""KNNTopology timing (and, with a -DWTP_DIAG build through WTP_LIB, the brick kernel's per-phase wave-cycle shares).
usage: exp_knn_diag.py [n] [k] [reps]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import wtp_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = wtp_amd.Context(0)
xyz = torch.empty((n, 3), dtype=torch.float32, device="cuda")
ctx.gen_uniform_dev(wtp_amd.synth.SEED, 0, n, 3, np.float32, xyz.data_ptr())
idx = torch.empty((n, k), dtype=torch.int32, device="cuda")
lib = wtp_amd.load_library()
out = (C.c_ulonglong * 16)()
ctx.knn_dev(xyz.data_ptr(), n, 3, np.float32, k, False, idx.data_ptr())
torch.cuda.synchronize()
lib.wtp_debug_diag(ctx._h, out)  # reset
t0 = time.perf_counter()
for _ in range(reps):
    ctx.knn_dev(xyz.data_ptr(), n, 3, np.float32, k, False, idx.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("knn n=%d k=%d: %.3f ms per call, %.1f Mpoints/s" % (n, k, dt * 1e3, n / dt / 1e6))
lib.wtp_debug_diag(ctx._h, out)
names = ["stage", "query_setup", "scan", "select", "prune_compact", "sort_out", "force_loop"]
tot = sum(out[i] for i in range(7)) or 1
if out[7]:
    print({names[i]: round(out[i] / tot, 4) for i in range(7)}, "waves", out[7], "cycles/wave", tot // max(out[7], 1))
tm = ctx.timers() if hasattr(ctx, "timers") else None
if tm:
    print(tm)
