#!/usr/bin/env python3
"""Reproduction attempt of a one-off stall: sharded driver at world=1 without explicit syncs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, wtp_amd
from whatsthepoint_jl_amd import sharded

class FakeDist:
    def get_rank(self): return 0
    def get_world_size(self): return 1

n = 10_000_000
s = float(n) ** (-1 / 3)
ctx = wtp_amd.Context(0)
def gen(first, m):
    t = torch.empty((m, 3), dtype=torch.float32, device="cuda")
    ctx.gen_uniform_dev(wtp_amd.synth.SEED, first, m, 3, np.float32, t.data_ptr())
    return t
print("a", flush=True)
xyz, gid, cuts = sharded.uniform_shard(gen, 0, 1, n, wtp_amd.synth.SEED, "cuda")
print("b", flush=True)
force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
drv = sharded.ShardedRelax(sharded.GpuEngine(ctx, s, force, 21, s / 2000, s / 20), FakeDist(), xyz, gid, cuts,
                           sharded.ghost_width(n, 21))
drv.run(3)
print("c", flush=True)
torch.cuda.synchronize(); ctx.timers_reset()
t0 = time.perf_counter(); drv.run(10); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("sharded driver world=1: ms/step", round(dt * 1e3, 3), ctx.timers(), flush=True)
