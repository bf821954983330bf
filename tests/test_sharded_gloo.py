"""N>1 path on CPU ranks (gloo, world_size 2 and 3): slab partition, migration, ghost exchange
and the all-reduced stop scalars.  The local sweep is the CPU oracle here (the product engine is
libwtp on a GPU); what is under test is the distributed logic of whatsthepoint.jl_amd/sharded.py:
a sharded run must reproduce the single-domain run point for point."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    def __init__(self, s, k, alo, amax):
        self.s, self.k, self.alo, self.amax = s, k, alo, amax

    def sweep(self, local_xyz, n_ghost):
        import oracle as O

        r = O.relax_sweep(local_xyz.numpy(), n_ghost, self.s, 2, 0.2, 1.0, 3.0, self.k, self.alo, self.amax)
        cv, s1, s2 = O.dnn_cv(r["nn_dist"], np.full(len(local_xyz), self.s, np.float32), n_ghost)
        st = dict(max_force=float(r["forces"].max()) if len(r["forces"]) else 0.0, sum_u=s1, sum_u2=s2,
                  n_move=len(r["forces"]), n_fallback=0)
        return torch.from_numpy(r["p"]), st


class ResidentOracleEngine(OracleEngine):
    """CPU stand-in for GpuEngine's resident protocol (open / layers / set_ghosts / step / positions)."""

    resident = True

    def open(self, xyz):
        self.x = xyz.clone()
        self.g = torch.zeros((0, 3), dtype=xyz.dtype)

    def layers(self, axis, lo_in, hi_in, lo_out, hi_out):
        v = self.x[:, axis]
        rows = torch.cat([self.x.contiguous().view(torch.int32),
                          torch.arange(len(v), dtype=torch.int32)[:, None]], 1)
        return rows[v < lo_in], rows[v >= hi_in], int(((v < lo_out) | (v >= hi_out)).sum())

    def set_ghosts(self, rows4):
        self.g = rows4[:, :3].contiguous().view(torch.float32)

    def step(self):
        self.x, st = self.sweep(torch.cat([self.g, self.x]).contiguous(), len(self.g))
        return st

    def step_and_layers(self, axis, lo_in, hi_in, lo_out, hi_out):
        st = self.step()
        return st, self.layers(axis, lo_in, hi_in, lo_out, hi_out)

    def positions(self):
        return self.x

    def close(self):
        pass


def _wall(wtp_amd, n_wall):
    b = wtp_amd.synth.uniform(n_wall, 3, np.float32, 19)
    f = np.arange(n_wall) % 6
    b[np.arange(n_wall), f % 3] = (f // 3).astype(np.float32)    # on the faces of the unit cube
    return b


def _worker(rank, world, port, n_total, iters, q, margin=None, resident=False, n_wall=0, use_run=False):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wtp_amd
    from whatsthepoint_jl_amd import sharded

    k = 21
    s = float(n_total) ** (-1.0 / 3.0)
    gen = lambda first, n: torch.from_numpy(wtp_amd.synth.uniform(n, 3, np.float32, 7, first))
    xyz, gid, cuts = sharded.uniform_shard(gen, rank, world, n_total, 7, "cpu", chunk=5000)
    eng = (ResidentOracleEngine if resident else OracleEngine)(s, k, s / 2000, s / 20)
    wall = torch.from_numpy(_wall(wtp_amd, n_wall)) if n_wall else None
    drv = sharded.ShardedRelax(eng, dist, xyz, gid, cuts, sharded.ghost_width(n_total, k), margin=margin, wall_xyz=wall)
    if n_wall:   # the stop rules of _relax! over the reduced scalars: tol = 0 -> exactly `iters` sweeps
        assert len(drv.relax(max_iters=50, tol=1e30)) == 1          # |F| s < tol after the first sweep
        conv = [drv.history[0]["max_force"]] + drv.relax(max_iters=iters - 1, tol=0.0)
    elif use_run:   # the benchmark loop: one collective per iteration, scalars reduced one phase later
        last = drv.run(iters)
        conv = [h["max_force"] for h in drv.history]
        assert len(conv) == iters and last["max_force"] == conv[-1]
    else:
        conv = [drv.step()["max_force"] for _ in range(iters)]
    allp = drv.gather_global(n_total)
    if rank == 0:
        q.put((conv, allp.numpy(), [h["n_ghost"] for h in drv.history], [h["n_move"] for h in drv.history],
               drv.migrations))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("resident,use_run", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("world,margin", [(2, None), (3, None), (2, 0.0), (3, 0.0)])
def test_sharded_matches_single_domain(O, wtp, world, margin, resident, use_run):
    # margin=None: lazy migration (points may stray a quarter ghost width past a cut);
    # margin=0: every crossing is handed over at once, so the migration path runs every iteration
    n_total, iters = 6000, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, iters, q, margin, resident, 0, use_run))
             for r in range(world)]
    for p in procs:
        p.start()
    try:
        conv, allp, n_ghost, n_move, migrations = q.get(timeout=240)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    finally:
        for p in procs:  # a rank that died leaves its peers waiting in a collective
            if p.is_alive():
                p.terminate()
    x = wtp.synth.uniform(n_total, 3, np.float32, 7)
    s = float(n_total) ** (-1.0 / 3.0)
    ref = O.relax_loop(x, 0, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s / 20, max_iters=iters, tol=0.0, rebuild_every=1,
                       stall_after=0)
    assert np.array_equal(allp, ref["p"])                      # decomposition-independent, bit for bit
    assert np.allclose(conv, ref["conv"], rtol=0, atol=0)
    assert all(g > 0 for g in n_ghost) and all(m == n_total for m in n_move)
    if margin == 0.0:
        assert migrations >= 1


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("world", [1, 3])
def test_sharded_with_a_global_wall_and_stop_rules(O, wtp, world, resident):
    """Volume-only repel: the boundary wall is the fixed head of the snapshot on every rank
    (src/repel.jl:80-84); the driver's relax() applies the reference's stop rules."""
    n_total, n_wall, iters = 6000, 900, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, iters, q, None, resident, n_wall))
             for r in range(world)]
    for p in procs:
        p.start()
    try:
        conv, allp, n_ghost, n_move, migrations = q.get(timeout=240)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    x = wtp.synth.uniform(n_total, 3, np.float32, 7)
    s = float(n_total) ** (-1.0 / 3.0)
    ref = O.relax_loop(np.concatenate([_wall(wtp, n_wall), x]), n_wall, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s / 20,
                       max_iters=iters, tol=0.0, rebuild_every=1, stall_after=0)
    assert np.array_equal(allp, ref["p"])
    assert np.allclose(conv, ref["conv"], rtol=0, atol=0)
    assert all(m == n_total for m in n_move) and all(g >= (n_wall if world == 1 else 1) for g in n_ghost)


# ---- the same logic with the PRODUCT engine: 2 ranks sharing the one GPU of the test box, payloads
# staged through host memory because gloo carries CPU tensors (RCCL needs one GPU per rank) ----------
def _gpu_worker(rank, world, port, n_total, iters, q, ghost_w_over_s=None, n_wall=0, margin=None, use_run=False):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wtp_amd
    from whatsthepoint_jl_amd import sharded

    torch.cuda.set_device(0)
    ctx = wtp_amd.Context(0)
    k = 21
    s = float(n_total) ** (-1.0 / 3.0)

    def gen(first, n):
        t = torch.empty((n, 3), dtype=torch.float32, device="cuda")
        ctx.gen_uniform_dev(7, first, n, 3, np.float32, t.data_ptr())
        return t

    xyz, gid, cuts = sharded.uniform_shard(gen, rank, world, n_total, 7, "cuda", chunk=50000)
    eng = sharded.GpuEngine(ctx, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), k, s / 2000, s / 20)
    w = sharded.ghost_width(n_total, k) if ghost_w_over_s is None else ghost_w_over_s * s
    wall = torch.from_numpy(_wall(wtp_amd, n_wall)).cuda() if n_wall else None
    drv = sharded.ShardedRelax(eng, dist, xyz, gid, cuts, w, comm_device="cpu", wall_xyz=wall, margin=margin)
    if use_run:
        drv.run(iters)
        conv = [h["max_force"] for h in drv.history]
    else:
        conv = drv.relax(max_iters=iters, tol=0.0) if n_wall else [drv.step()["max_force"] for _ in range(iters)]
    allp = drv.gather_global(n_total)
    if rank == 0:
        q.put((conv, allp.numpy(), drv.widened, drv.w / s, drv.migrations))
    dist.barrier()
    eng.close()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,ghost_w_over_s,n_wall,margin,use_run",
                         [(2, None, 0, None, False), (2, 0.6, 0, None, False), (2, None, 4000, None, False),
                          (2, None, 0, 0.0, False), (3, None, 0, None, False),
                          (2, None, 0, None, True), (2, 0.6, 0, None, True), (3, None, 0, 0.0, True)])
def test_sharded_gpu_engine_two_ranks_one_gpu(O, wtp, world, ghost_w_over_s, n_wall, margin, use_run):
    # ghost_w_over_s = 0.6: a ghost layer thinner than the force law's support — the sweep must
    # notice (n_uncovered), and the driver must undo, widen and repeat until the answer is global
    n_total, iters = 120000, 3   # world = 3: an interior rank with two neighbours
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker,
                         args=(r, world, port, n_total, iters, q, ghost_w_over_s, n_wall, margin, use_run))
             for r in range(world)]
    for p in procs:
        p.start()
    try:
        conv, allp, widened, w_over_s, migrations = q.get(timeout=240)
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    assert (widened == 0) if ghost_w_over_s is None else (widened >= 2 and w_over_s > 1.0)
    if margin == 0.0:   # every crossing is handed over at once: the engine's session restarts on the new owned set
        assert migrations >= 1
    x = wtp.synth.uniform(n_total, 3, np.float32, 7)
    s = float(n_total) ** (-1.0 / 3.0)
    snap = np.concatenate([_wall(wtp, n_wall), x]) if n_wall else x
    ref = O.relax_loop(snap, n_wall, s, 2, 0.2, 1.0, 3.0, 21, s / 2000, s / 20, max_iters=iters, tol=0.0, rebuild_every=1,
                       stall_after=0)
    assert len(conv) == iters and np.allclose(conv, ref["conv"], rtol=1e-3)
    err = np.abs(allp - ref["p"]).max(axis=1) / s
    assert np.quantile(err, 0.999) < 1e-4 and err.max() < 1e-2
