#!/usr/bin/env python3
"""Per-phase wave-cycle shares of the brick kernel from a -DWTP_DIAG build (WTP_LIB=.../libwtp_diag.so).
Read SHARES, never the diagnostic build's run time."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import wtp_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
s = float(n) ** (-1.0 / 3.0)
ctx = wtp_amd.Context(0)
xyz = torch.empty((n, 3), dtype=torch.float32, device="cuda")
ctx.gen_uniform_dev(wtp_amd.synth.SEED, 0, n, 3, np.float32, xyz.data_ptr())
sess = ctx.relax(None, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20,
                 device_ptr=(xyz.data_ptr(), n, 3, np.float32))
sess.run_async_free(2, 1)
lib = wtp_amd.load_library()
out = (C.c_ulonglong * 16)()
lib.wtp_debug_diag(ctx._h, out)
sess.run_async_free(3, 1)
lib.wtp_debug_diag(ctx._h, out)
names = ["stage", "query_setup", "scan", "select", "prune_compact", "step_out", "force_loop"]
tot = sum(out[i] for i in range(7)) or 1
print({names[i]: round(out[i] / tot, 4) for i in range(7)}, "waves", out[7], "cycles/wave", tot // max(out[7], 1))
print("raw", [int(out[i]) for i in range(16)], "scan lane utilisation (WTP_DIAG=2 builds)", round(out[4] / max(out[3], 1), 3))
if out[15]:  # round-2 sweep (wtp_cs2.hip): 0 stage, 1 query setup, 2 scan, 3 ring pass, 4 step/out, 5 follow-up; 8.. trip counts
    cn = ["cell_table", "query_setup", "scan", "ring_pass", "step_out", "followup", "table_scan", "staging"]
    ct = sum(out[i] for i in range(8)) or 1
    w = out[15]
    print("cs2 shares", {cn[i]: round(out[i] / ct, 4) for i in range(8)}, "waves", int(w), "cycles/wave", ct // w)
    print("cs2 per wave-round: scan steps %.2f (busy lanes %.3f), ring batches %.2f; queries/round %.1f; rounds/brick %.3f; misses/wave/brick %.3f"
          % (out[11] / max(out[9], 1), out[12] / max(out[11] * 64, 1), out[13] / max(out[9], 1), out[10] / max(out[9], 1),
             out[9] / max(out[8], 1), out[14] / max(out[8], 1)))
