"""Sanity check of the RCCL path on a 1-GPU box: a world of one rank over the "nccl" backend running the
sharded driver's collective (all_gather of float64 device tensors) — catches environment problems (IPC
mode, missing librccl) that the gloo rehearsals cannot."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.arange(9, dtype=torch.float64, device="cuda")
out = [torch.zeros_like(x)]
t0 = time.perf_counter()
dist.all_gather(out, x)
torch.cuda.synchronize()
print("all_gather ok", bool((out[0] == x).all()), f"{1e3*(time.perf_counter()-t0):.1f} ms (first call)")
t0 = time.perf_counter()
for _ in range(100):
    dist.all_gather(out, x)
torch.cuda.synchronize()
print(f"all_gather steady state {1e4*(time.perf_counter()-t0):.1f} us per call")
# the sharded driver itself over this backend (one slab, no neighbours): device-resident payloads
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import wtp_amd
from whatsthepoint_jl_amd import sharded
n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 21
s = float(n) ** (-1.0 / 3.0)
force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
ctx = wtp_amd.Context(0)

def gen(first, m):
    t = torch.empty((m, 3), dtype=torch.float32, device="cuda")
    ctx.gen_uniform_dev(wtp_amd.synth.SEED, first, m, 3, np.float32, t.data_ptr())
    return t

own_xyz, own_gid, cuts = sharded.uniform_shard(gen, 0, 1, n, wtp_amd.synth.SEED, "cuda")
drv = sharded.ShardedRelax(sharded.GpuEngine(ctx, s, force, k, s / 2000, s / 20), dist, own_xyz, own_gid, cuts,
                           sharded.ghost_width(n, k, 8.0))
drv.run(3)
torch.cuda.synchronize()
ctx.timers_reset()
t0 = time.perf_counter()
drv.run(20)
torch.cuda.synchronize()
tm = ctx.timers()
print(f"sharded driver, 1 rank over nccl: {1e3*(time.perf_counter()-t0)/20:.3f} ms per iteration at {n} points "
      f"(device: hash {tm['hash_ms']/20:.3f} sweep {tm['sweep_ms']/20:.3f} other {tm['other_ms']/20:.3f})")
dist.barrier()
dist.destroy_process_group()
print("rccl single-rank path OK")
