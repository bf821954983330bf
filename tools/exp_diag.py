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
out = (C.c_ulonglong * 8)()
lib.wtp_debug_diag(ctx._h, out)
sess.run_async_free(3, 1)
lib.wtp_debug_diag(ctx._h, out)
names = ["stage", "query_setup", "scan", "select", "prune_compact", "step_out", "force_loop"]
tot = sum(out[i] for i in range(7)) or 1
print({names[i]: round(out[i] / tot, 4) for i in range(7)}, "waves", out[7], "cycles/wave", tot // max(out[7], 1))
print("raw", [int(out[i]) for i in range(8)], "scan lane utilisation (WTP_DIAG=2 builds)", round(out[4] / max(out[3], 1), 3))
