#!/usr/bin/env python3
"""Per-phase wave-cycle shares of brick_kernel<0,21,0> (KNNTopology) / brick_kernel<1,21,0> (WTP_FULL_SELECT=1 sweep)
from a -DWTP_DIAG build (WTP_LIB=.../libwtp_diag.so).  Read SHARES, never the diagnostic build's run time."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import wtp_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "knn"
k = 21
ctx = wtp_amd.Context(0)
lib = wtp_amd.load_library()
out = (C.c_ulonglong * 16)()
x = torch.empty((n, 3), dtype=torch.float32, device="cuda")
ctx.gen_uniform_dev(wtp_amd.synth.SEED, 0, n, 3, np.float32, x.data_ptr())
reps = 4
if mode == "knn":
    idx = torch.empty((n, k), dtype=torch.int32, device="cuda")
    ctx.knn_dev(x.data_ptr(), n, 3, np.float32, k, False, idx.data_ptr())
    lib.wtp_debug_diag(ctx._h, out)
    for _ in range(reps):
        ctx.knn_dev(x.data_ptr(), n, 3, np.float32, k, False, idx.data_ptr())
else:
    s = float(n) ** (-1.0 / 3.0)
    sess = ctx.relax(None, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20,
                     device_ptr=(x.data_ptr(), n, 3, np.float32))
    sess.run_async_free(2, 1)
    lib.wtp_debug_diag(ctx._h, out)
    sess.run_async_free(reps, 1)
lib.wtp_debug_diag(ctx._h, out)
if out[15]:  # wtp_ksel.hip: eight phases, wave count in slot 15
    names = ["cell_table_to_barrier", "last_barrier", "query_setup", "scan", "extraction", "keys", "network_window", "rows_force_out",
             "wait_slowest_wave", "prefix_tables", "staging"]
    tot = sum(out[i] for i in range(11)) or 1
    print(mode, n, "ksel", {names[i]: round(out[i] / tot, 4) for i in range(11)}, "waves", out[15], "cycles/wave", tot // out[15])
else:
    names = ["stage", "query_setup", "scan", "select", "prune_compact", "topo_out", "force_loop"]
    tot = sum(out[i] for i in range(7)) or 1
    print(mode, n, {names[i]: round(out[i] / tot, 4) for i in range(7)}, "waves", out[7], "cycles/wave", tot // max(out[7], 1))
    print("cycles per wave per call by phase", {names[i]: int(out[i] // max(out[7], 1) // reps) for i in range(7)})
