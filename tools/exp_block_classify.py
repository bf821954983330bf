"""Cost of the block driver's classification at scale on ONE GPU: two ranks as threads, rank 0 owns x < 0.9 of a 10 M-point
uniform cloud (9 M points, one peer), rank 1 the thin rest — so rank 0's kernels run nearly alone.  For a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 tools/exp_block_classify.py   (WTP_BLOCK_ROWS=0: every slot is looked at)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wtp_amd
from whatsthepoint_jl_amd import blockc

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
s = float(n) ** (-1.0 / 3.0)
FORCE = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
inf = float("inf")
boxes = np.array([[-inf, -inf, -inf, 0.9, inf, inf], [0.9, -inf, -inf, inf, inf, inf]])

def worker(rank, hub):
    torch.cuda.set_device(0)
    ctx = wtp_amd.Context(0)
    def gen(first, m):
        t = torch.empty((m, 3), dtype=torch.float32, device="cuda")
        ctx.gen_uniform_dev(wtp_amd.synth.SEED, first, m, 3, np.float32, t.data_ptr())
        return t
    xyz, gid = blockc.shard_stream(gen, boxes, rank, n)
    drv = blockc.BlockRelax(ctx, rank, 2, boxes, xyz, gid, 2.1 * s, s, FORCE, 21, s / 2000, s / 20, transport=blockc.loopback_transport(hub, rank))
    drv.run(4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = drv.run(12)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 12
    drv.close(); ctx.close()
    return rank, dt, out["n_owned"], out["n_ghost"]

for r in blockc.run_threads(2, worker):
    print("rank %d: %.3f ms per iteration, owned %d ghosts %d" % (r[0], r[1] * 1e3, r[2], r[3]))
