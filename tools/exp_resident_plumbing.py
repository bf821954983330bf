"""Cost of the resident sharded iteration without any communication: one process, one slab
[0,1) in z, ghost layers = periodic images of the rank's own boundary layers.  Same library
calls and torch ops as ShardedRelax._step_resident minus the exchange.  GPU only."""
import sys, time, os
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wtp_amd
from whatsthepoint_jl_amd import sharded

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8      # emulated world size (sets density / ghost width)
iters = 20
torch.cuda.set_device(0)
ctx = wtp_amd.Context(0)
n_total = n * world
s = float(n_total) ** (-1.0 / 3.0)
k = 21
force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
# slab of thickness 1/world: generate n points in [0,1)^2 x [0, 1/world)
x = torch.empty((n, 3), dtype=torch.float32, device="cuda")
ctx.gen_uniform_dev(7, 0, n, 3, np.float32, x.data_ptr())
x[:, 2] /= world
L = 1.0 / world
gc = float(sys.argv[3]) if len(sys.argv) > 3 else 1.25
w = sharded.ghost_width(n_total, k, ghost_cells=gc)
margin = 0.25 * w
w_eff = w + margin
eng = sharded.GpuEngine(ctx, s, force, k, s / 2000, s / 20)
eng.set_coverage(2, -w_eff, L + w_eff)
eng.open(x)
shift = torch.tensor([0, 0, L, 0], dtype=torch.float32, device="cuda")


nxt = [None]


def iteration(sync_each=False, T=None):
    t0 = time.perf_counter()
    if nxt[0] is None:
        lo_rows, hi_rows, stray = eng.layers(2, w_eff, L - w_eff, -margin, L + margin)
    else:
        lo_rows, hi_rows, stray = nxt[0]
    if T is not None:
        torch.cuda.synchronize(); t1 = time.perf_counter(); T[0] += t1 - t0
    g = torch.cat([hi_rows.view(torch.float32) - shift, lo_rows.view(torch.float32) + shift]).view(torch.int32)
    eng.set_ghosts(g)
    if T is not None:
        torch.cuda.synchronize(); t2 = time.perf_counter(); T[1] += t2 - t1
    st, nxt[0] = eng.step_and_layers(2, w_eff, L - w_eff, -margin, L + margin)
    if T is not None:
        t3 = time.perf_counter(); T[2] += t3 - t2
    return st, g.shape[0]


unc = []
for _ in range(3):
    st, ng = iteration()
    unc.append(st['n_uncovered'])
print('ghost_cells', gc, 'w/s', w / s, 'n_uncovered in the first iterations', unc, flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    st, ng = iteration()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
T = [0.0, 0.0, 0.0]
for _ in range(iters):
    iteration(T=T)
print(f"n_own={n} n_ghost={ng} ({100.0*ng/n:.1f}%)  resident iteration {dt*1e3:.3f} ms  "
      f"[layers {T[0]/iters*1e3:.3f}  ghosts+refix {T[1]/iters*1e3:.3f}  step {T[2]/iters*1e3:.3f}]  "
      f"n_fallback={st['n_fallback']}", flush=True)
ctx.timers_reset()
for _ in range(iters):
    iteration()
tm = ctx.timers()
print("lib phases per iter: hash %.3f sweep %.3f other %.3f" % (tm["hash_ms"]/iters, tm["sweep_ms"]/iters, tm["other_ms"]/iters))
eng.close()
# plain session of the same size for reference
x = torch.empty((n + ng, 3), dtype=torch.float32, device="cuda")
ctx.gen_uniform_dev(7, 0, n + ng, 3, np.float32, x.data_ptr())
x[:, 2] /= world
sess = ctx.relax(None, ng, s, force, k, s / 2000, s / 20, device_ptr=(x.data_ptr(), n + ng, 3, np.float32))
for _ in range(3):
    sess.step(True)
t0 = time.perf_counter()
for _ in range(iters):
    sess.step(True)
dt2 = (time.perf_counter() - t0) / iters
print(f"plain step (n={n+ng}, n_fixed={ng}) with stats read-back: {dt2*1e3:.3f} ms", flush=True)
sess.close()
