#!/usr/bin/env python3
"""Secondary configs of BASELINE.json on one GPU: C2 (1 M uniform, KNNTopology k=21, device
resident) and C5 (graded cloud, RadiusTopology, fp32 vs fp64 set agreement)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import wtp_amd

ctx = wtp_amd.Context(0)
out = {}

# ---- C2: k-NN topology, device resident --------------------------------------------------------
for n in (1_000_000, 10_000_000):
    k = 21
    xyz = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    idx = torch.empty((n, k), dtype=torch.int32, device="cuda")
    ctx.gen_uniform_dev(wtp_amd.synth.SEED, 0, n, 3, np.float32, xyz.data_ptr())
    for _ in range(2):
        ctx.knn_dev(xyz.data_ptr(), n, 3, np.float32, k, False, idx.data_ptr())
    ctx.timers_reset()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.knn_dev(xyz.data_ptr(), n, 3, np.float32, k, False, idx.data_ptr())
    dt = (time.perf_counter() - t0) / reps
    tm = ctx.timers()
    out[f"C2_knn_{n}"] = dict(ms=round(dt * 1e3, 3), mpts=round(n / dt / 1e6, 1), hash_ms=round(tm["hash_ms"] / reps, 3),
                              search_ms=round(tm["sweep_ms"] / reps, 3),
                              alg_gbs=round(151.0 * n / dt / 1e9, 1))
    del xyz, idx

# ---- C5: graded cloud, radius stencils, fp32 vs fp64 ------------------------------------------------
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
g64 = wtp_amd.synth.graded(n, dtype=np.float64)
g32 = g64.astype(np.float32)
h_w = (1.0 / (n * 2.2)) ** (1.0 / 3.0)  # rough wall spacing of the thinned stream
r = 2.5 * h_w
res = {}
for name, pts in (("f32", g32), ("f64", g64.astype(np.float32).astype(np.float64))):
    ctx.timers_reset()
    t0 = time.perf_counter()
    off, idx = ctx.radius(pts, r)
    dt = time.perf_counter() - t0
    tm = ctx.timers()
    res[name] = (off, idx)
    out[f"C5_radius_{name}"] = dict(n=n, r=r, nnz=int(off[-1]), mean_nbrs=round(off[-1] / n, 2), max_nbrs=int(np.diff(off).max()),
                                    wall_ms=round(dt * 1e3, 1), hash_ms=round(tm["hash_ms"], 2), kernels_ms=round(tm["sweep_ms"], 2),
                                    mpts_kernels=round(n / (tm["sweep_ms"] + tm["hash_ms"]) / 1e3, 1),
                                    mpairs_kernels=round(off[-1] / (tm["sweep_ms"] + tm["hash_ms"]) / 1e3, 1))
same = np.array_equal(res["f32"][0], res["f64"][0]) and np.array_equal(res["f32"][1], res["f64"][1])
out["C5_f32_vs_f64_identical_sets"] = bool(same)
if not same:
    d = np.abs(np.diff(res["f32"][0]) - np.diff(res["f64"][0]))
    out["C5_rows_with_different_counts"] = int((d > 0).sum())
print(json.dumps(out, indent=1))
