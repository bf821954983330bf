#!/usr/bin/env python3
"""Experiment driver: time the repel step at N points for the current tunables (env WTP_RHO,
WTP_GAMMA_CAP) and print fallback counts + per-phase device times."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import wtp_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
k = 21
s = float(n) ** (-1.0 / 3.0)
ctx = wtp_amd.Context(0)
xyz = torch.empty((n, 3), dtype=torch.float32, device="cuda")
ctx.gen_uniform_dev(wtp_amd.synth.SEED, 0, n, 3, np.float32, xyz.data_ptr())
sess = ctx.relax(None, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), k, s / 2000, s / 20,
                 device_ptr=(xyz.data_ptr(), n, 3, np.float32))
st = sess.step(True)
fb0 = st["n_fallback"]
sess.run_async_free(2, 1)
ctx.timers_reset()
t0 = time.perf_counter()
conv, last = sess.run(steps, 1)
dt = time.perf_counter() - t0
tm = ctx.timers()
print(json.dumps(dict(rho=os.environ.get("WTP_RHO"), gamma=os.environ.get("WTP_GAMMA_CAP"), n=n,
                      ms_per_step=round(dt / steps * 1e3, 3), fb_first=fb0, fb_last=last["n_fallback"],
                      hash=round(tm["hash_ms"] / steps, 3), sweep=round(tm["sweep_ms"] / steps, 3),
                      other=round(tm["other_ms"] / steps, 3), max_force=last["max_force"])))
