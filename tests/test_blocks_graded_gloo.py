"""A GRADED cloud over count-quantile cuts on CPU ranks (gloo): the block decomposition with equal-count cut planes and a
variable spacing law (BoundaryLayerSpacing, src/discretization/spacings.jl:121-133) evaluated at every owned point's
current position.  The local sweep is the CPU oracle; under test is the distributed logic (quantile cuts, ghost
exchange, migration, global ids): the run must reproduce the single-domain run bit for bit.  The product path for
the same configuration is the C block driver (tests/test_gpu_blockc.py: graded cloud, device-evaluated law)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HW, HB, DELTA = 0.03, 0.12, 0.2


def _wall_points():
    g = np.linspace(0.0, 1.0, 12, dtype=np.float32)
    u, v = np.meshgrid(g, g, indexing="ij")
    faces = []
    for a in range(3):
        for c in (0.0, 1.0):
            f = np.empty((u.size, 3), dtype=np.float32)
            f[:, a] = c
            f[:, (a + 1) % 3] = u.ravel()
            f[:, (a + 2) % 3] = v.ravel()
            faces.append(f)
    return np.concatenate(faces)


def _quantile_cuts(x, grid):
    cuts = []
    for a in range(3):
        v = np.sort(x[:, a].astype(np.float64))
        cuts.append([float(np.float32(0.5 * (v[len(v) * i // grid[a] - 1] + v[len(v) * i // grid[a]]))) for i in range(1, grid[a])])
    return cuts


def _worker(rank, grid, port, n_total, iters, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    world = grid[0] * grid[1] * grid[2]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    import wtp_amd
    from test_sharded_gloo import ResidentOracleEngine
    from whatsthepoint_jl_amd import blocks

    wall = _wall_points()

    class GradedEngine(ResidentOracleEngine):
        def sweep(self, local_xyz, n_ghost):
            x = local_xyz.numpy()
            sp = O.spacing_boundary_layer(x, wall, HW, HB, DELTA).astype(np.float32)  # s = spacing(x_i), src/repel.jl:260
            r = O.relax_sweep(x, n_ghost, sp, 2, 0.2, 1.0, 3.0, self.k, self.alo, self.amax)
            cv, s1, s2 = O.dnn_cv(r["nn_dist"], sp, n_ghost)
            st = dict(max_force=float(r["forces"].max()) if len(r["forces"]) else 0.0, sum_u=s1, sum_u2=s2,
                      n_move=len(r["forces"]), n_fallback=0)
            return torch.from_numpy(r["p"]), st

    x = wtp_amd.synth.graded(n_total, HB / HW, DELTA, np.float32, 7)
    cuts = _quantile_cuts(x, grid)
    idx = blocks.block_of_rank(rank, grid)
    m = np.ones(n_total, dtype=bool)
    for a in range(3):
        lo = -np.inf if idx[a] == 0 else np.float32(cuts[a][idx[a] - 1])
        hi = np.inf if idx[a] == grid[a] - 1 else np.float32(cuts[a][idx[a]])
        m &= (x[:, a] >= lo) & (x[:, a] < hi)
    gid = np.nonzero(m)[0]
    eng = GradedEngine(0.0, 21, HW / 2000, HW / 20)
    drv = blocks.BlockShardedRelax(eng, dist, torch.from_numpy(x[gid]), torch.from_numpy(gid.astype(np.int64)), grid, cuts,
                                   0.42, margin=0.01)
    conv = [drv.step()["max_force"] for _ in range(iters)]
    allp = drv.gather_global(n_total)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(gid)]))
    if rank == 0:
        q.put((conv, allp.numpy(), [int(c) for c in counts], drv.migrations))
    dist.barrier()
    dist.destroy_process_group()


def test_graded_cloud_quantile_cuts_match_single_domain(O, wtp):
    grid, n_total, iters = (1, 2, 2), 4000, 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, grid, port, n_total, iters, q)) for r in range(4)]
    for p in procs:
        p.start()
    conv, allp, counts, migrations = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # per-axis quantiles (blocks.py's cuts are one list per axis): equal counts to a few per cent; the nested cuts of
    # blockc.orthtree_boxes (the C driver's partition) are equal to a point (tests/test_gpu_blockc.py)
    assert max(counts) - min(counts) <= 0.05 * n_total / 4 and sum(counts) == n_total, counts
    x = wtp.synth.graded(n_total, HB / HW, DELTA, np.float32, 7)
    wall = _wall_points()
    law = lambda pts: O.spacing_boundary_layer(np.ascontiguousarray(pts, dtype=np.float32), wall, HW, HB, DELTA)
    ref = O.relax_loop(x, 0, law, 2, 0.2, 1.0, 3.0, 21, HW / 2000, HW / 20, max_iters=iters, tol=0.0, stall_after=0)
    assert np.array_equal(allp, ref["p"])
    assert np.array_equal(np.asarray(conv, dtype=np.float32), ref["conv"])
