"""What the block driver adds to the plain session's iteration when no row travels (one rank, no neighbour): the
classification's bookkeeping, the statistics read-back and its host synchronisation — per transport path of the C
driver (wtp_block_*): the degenerate RCCL path (one rank) and the host-callback path.  VERDICT r2: <= 0.05 ms."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
from whatsthepoint_jl_amd import blockc

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
iters = 30
FORCE = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
s = float(n) ** (-1.0 / 3.0)
ctx = wtp_amd.Context(0)
def gen(first, m):
    t = torch.empty((m, 3), dtype=torch.float32, device="cuda")
    ctx.gen_uniform_dev(wtp_amd.synth.SEED, first, m, 3, np.float32, t.data_ptr())
    return t
boxes = blockc.orthtree_boxes(None, 1, False)
xyz, gid = blockc.shard_stream(gen, boxes, 0, n)
for name, tr in (("rccl path (one rank)", None), ("host-callback path", blockc.loopback_transport(blockc.LoopbackHub(1), 0))):
    drv = blockc.BlockRelax(ctx, 0, 1, boxes, xyz, gid, 2.0 * s, s, FORCE, 21, s / 2000, s / 20, transport=tr)
    drv.run(5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = drv.run(iters)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"block driver (C), {name}: {dt*1e3:.3f} ms per iteration at n={n}, host syncs per iteration {out['host_syncs']/iters:.2f}")
    drv.close()
with ctx.relax(None, 0, s, FORCE, 21, s / 2000, s / 20, device_ptr=(xyz.data_ptr(), n, 3, np.float32)) as t:
    t.run_async_free(5, 1); torch.cuda.synchronize(); t0 = time.perf_counter()
    t.run_async_free(iters, 1); torch.cuda.synchronize()
    print(f"plain session, {iters} iterations enqueued back to back: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration")
    t0 = time.perf_counter()
    for _ in range(iters):
        t.step(True)
    print(f"plain session, one wtp_relax_step (one read-back) per iteration: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration")
ctx.close()
