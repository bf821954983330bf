"""Block (orthtree) decomposition with the product engine: four gloo ranks sharing the one GPU of the test box
(1 x 2 x 2 blocks, payloads staged through host memory) against the single-domain session; and the box form
of the coverage check (wtp_relax_set_coverage_box)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORCE = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)


def _worker(rank, grid, port, n_total, iters, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    world = grid[0] * grid[1] * grid[2]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wtp_amd
    from whatsthepoint_jl_amd import blocks, sharded

    torch.cuda.set_device(0)
    ctx = wtp_amd.Context(0)
    k = 21
    s = float(n_total) ** (-1.0 / 3.0)

    def gen(first, n):
        t = torch.empty((n, 3), dtype=torch.float32, device="cuda")
        ctx.gen_uniform_dev(wtp_amd.synth.SEED, first, n, 3, np.float32, t.data_ptr())
        return t

    xyz, gid, cuts = blocks.uniform_block_shard(gen, rank, grid, n_total, "cuda")
    eng = sharded.GpuEngine(ctx, s, FORCE, k, s / 2000, s / 20)
    drv = blocks.BlockShardedRelax(eng, dist, xyz, gid, grid, cuts, sharded.ghost_width(n_total, k), comm_device="cpu")
    conv = [drv.step()["max_force"] for _ in range(iters)]
    allp = drv.gather_global(n_total)
    if rank == 0:
        q.put((conv, allp.cpu().numpy(), [h["n_ghost"] for h in drv.history], drv.migrations, drv.widened))
    dist.barrier()
    eng.close()
    ctx.close()
    dist.destroy_process_group()


def test_blocks_on_the_gpu_match_single_domain(wtp):
    import torch.multiprocessing as mp

    grid, n_total, iters = (1, 2, 2), 400_000, 6
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, grid, port, n_total, iters, q)) for r in range(4)]
    for p in procs:
        p.start()
    conv, allp, n_ghost, migrations, widened = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x = wtp.synth.uniform(n_total, 3, np.float32)
    s = float(n_total) ** (-1.0 / 3.0)
    with wtp.Context(0) as c, c.relax(x, 0, s, FORCE, 21, s / 2000, s / 20) as t:
        ref_conv, _ = t.run(iters, 1)
        ref = t.positions()
    # same points, same ids; each rank lays its own grid over its block, so sums run in another order: rounding
    assert np.abs(allp - ref).max() <= 2e-5 * s
    assert np.allclose(conv, ref_conv, rtol=1e-4)
    assert all(g > 0 for g in n_ghost)


def test_coverage_box_counts(wtp):
    n, k = 80000, 21
    s = n ** (-1.0 / 3.0)
    x = wtp.synth.uniform(n, 3, np.float32, 21)
    c = wtp.Context(0)
    try:
        _, d = c.knn(x, k, include_self=True, return_dist=True)
        rk, nn = d[:, k - 1].astype(np.float64), d[:, 1].astype(np.float64)
        xd = x.astype(np.float64)
        for lo3, hi3 in (([0, 0, 0], [1, 1, 1]), ([-2 * s] * 3, [1 + 2 * s] * 3), ([-np.inf, 0.2, -np.inf], [np.inf, 0.9, 0.6]),
                         ([-6 * s] * 3, [1 + 6 * s] * 3)):
            with c.relax(x, 0, s, FORCE, k, s / 2000, s / 20) as t:
                t.set_coverage_box(lo3, hi3)
                st = t.step(True)
            dd = np.minimum(xd - np.array(lo3), np.array(hi3) - xd).min(axis=1)   # distance to the nearest face
            most = int((np.maximum(rk, s) > dd * (1 - 1e-5)).sum())
            least = int((np.maximum(nn, s) > dd * (1 + 1e-5)).sum())
            assert least <= st["n_uncovered"] <= most, ((lo3, hi3), least, st["n_uncovered"], most)
        with c.relax(x, 0, s, FORCE, k, s / 2000, s / 20) as t:
            with pytest.raises(ValueError):
                t.set_coverage_box([0, 0, 1], [1, 1, 0])
    finally:
        c.close()
