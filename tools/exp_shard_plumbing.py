#!/usr/bin/env python3
"""Time the torch plumbing of one sharded iteration (masks, packing, concatenation) at 10 M owned
points, without communication: what the driver adds on top of the libwtp sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, wtp_amd
from whatsthepoint_jl_amd import sharded

n = 10_000_000
ctx = wtp_amd.Context(0)
def gen(first, m):
    t = torch.empty((m, 3), dtype=torch.float32, device="cuda")
    ctx.gen_uniform_dev(wtp_amd.synth.SEED, first, m, 3, np.float32, t.data_ptr())
    return t
xyz, gid, _ = sharded.uniform_shard(gen, 0, 1, n, wtp_amd.synth.SEED, "cuda")
class D:  # stand-in exposing what _pack/_unpack need
    pass
drv = sharded.ShardedRelax.__new__(sharded.ShardedRelax)
drv.xyz, drv.gid, drv.dev = xyz, gid, xyz.device
lo, hi, w = 0.0, 1.0, 2 * 2 * n ** (-1 / 3)
def plumbing():
    z = drv.xyz[:, 2]
    go_lo, go_hi = z < lo + 1e-4, z >= hi - 1e-4      # a few migrants
    keep = ~(go_lo | go_hi)
    gl_lo, gl_hi = keep & (z < lo + w), keep & (z >= hi - w)
    a = torch.cat([drv._pack(go_lo), drv._pack(gl_lo)]); b = torch.cat([drv._pack(go_hi), drv._pack(gl_hi)])
    xa, ga = drv._unpack(a, torch.float32); xb, gb = drv._unpack(b, torch.float32)
    own = torch.cat([drv.xyz[keep], xa[:10]]); g2 = torch.cat([drv.gid[keep], ga[:10]])
    local = torch.cat([xa[10:], xb, own]).contiguous()
    return local, g2
for _ in range(3): plumbing()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): local, g2 = plumbing()
torch.cuda.synchronize(); print("torch plumbing ms/iter", round((time.perf_counter() - t0) / 10 * 1e3, 3), local.shape)
