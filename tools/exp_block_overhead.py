"""Host-side cost of one block-driver iteration without any neighbour (one rank, 1 x 1 x 1 grid): what the Python
driver adds to the resident session's step (statistics collective, layer bookkeeping), per transport."""
import os, socket, sys, time
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
from whatsthepoint_jl_amd import blocks, sharded

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
with socket.socket() as so:
    so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
backend = os.environ.get("BACKEND", "nccl")
dist.init_process_group(backend, rank=0, world_size=1)
ctx = wtp_amd.Context(0)
s = float(n) ** (-1.0 / 3.0)
def gen(first, m):
    t = torch.empty((m, 3), dtype=torch.float32, device="cuda")
    ctx.gen_uniform_dev(wtp_amd.synth.SEED, first, m, 3, np.float32, t.data_ptr())
    return t
xyz, gid, cuts = blocks.uniform_block_shard(gen, 0, (1, 1, 1), n, "cuda")
eng = sharded.GpuEngine(ctx, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20)
drv = blocks.BlockShardedRelax(eng, dist, xyz, gid, (1, 1, 1), cuts, sharded.ghost_width(n, 21),
                               comm_device="cuda" if backend == "nccl" else "cpu")
for _ in range(5):
    drv.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
iters = 30
for _ in range(iters):
    drv.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
print(f"block driver, one rank, {backend}, WTP_COMM={os.environ.get('WTP_COMM','torch')}: {dt*1e3:.3f} ms per iteration at n={n}")
eng.close()
with ctx.relax(None, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20,
               device_ptr=(gen(0, n).data_ptr(), n, 3, np.float32)) as t:
    t.run_async_free(5, 1); torch.cuda.synchronize(); t0 = time.perf_counter()
    t.run_async_free(iters, 1); torch.cuda.synchronize()
    print(f"plain session: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration")
ctx.close(); dist.destroy_process_group()
