"""The context's RCCL communicator behind the C ABI (include/wtp.h: wtp_comm_*).  A one-GPU box can only form a
communicator of one rank, so the point-to-point round is exercised with the rank as its own low and high neighbour
(what a periodic axis of extent 1 would be): the posting order of the entry point pairs the low send with the low
receive and the high send with the high receive.  The multi-rank use of the same calls is the block driver's
WTP_COMM=abi transport (whatsthepoint.jl_amd/blocks.py), unmeasured on more than one GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def comm_ctx(wtp):
    with wtp.Context(0) as c:
        uid = c.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        c.comm_init(uid, 0, 1)
        import torch

        c.set_stream(torch.cuda.current_stream().cuda_stream)  # rows are stream-ordered: share torch's stream
        yield c
        c.comm_finalize()


def test_exchange_rows_with_itself_and_no_peer(wtp, comm_ctx):
    import torch

    c = comm_ctx
    rng = np.random.default_rng(3)
    a = torch.from_numpy(rng.random((1000, 4), dtype=np.float32)).cuda()
    b = torch.from_numpy(rng.random((700, 4), dtype=np.float32)).cuda()
    ra = torch.zeros((2048, 4), dtype=torch.float32, device="cuda")
    rb = torch.zeros((2048, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    n_lo, n_hi = c.comm_exchange_rows(0, 0, a.data_ptr(), 1000, b.data_ptr(), 700, ra.data_ptr(), rb.data_ptr(), 2048)
    torch.cuda.synchronize()
    assert (n_lo, n_hi) == (1000, 700)
    assert torch.equal(ra[:1000], a) and torch.equal(rb[:700], b)
    assert not ra[1000:].any() and not rb[700:].any()
    # one neighbour only, and none at all (the faces of the box)
    ra.zero_()
    torch.cuda.synchronize()
    assert c.comm_exchange_rows(0, -1, a.data_ptr(), 5, 0, 0, ra.data_ptr(), 0, 2048) == (5, 0)
    torch.cuda.synchronize()
    assert torch.equal(ra[:5], a[:5])
    assert c.comm_exchange_rows(-1, -1, 0, 0, 0, 0, 0, 0, 0) == (0, 0)
    # empty payloads travel as counts only
    assert c.comm_exchange_rows(0, 0, 0, 0, 0, 0, ra.data_ptr(), rb.data_ptr(), 16) == (0, 0)


def test_receive_buffer_too_small_fails_without_hanging(wtp, comm_ctx):
    import torch

    c = comm_ctx
    a = torch.ones((100, 4), dtype=torch.float32, device="cuda")
    r = torch.zeros((10, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(wtp.WtpArgumentError):
        c.comm_exchange_rows(0, -1, a.data_ptr(), 100, 0, 0, r.data_ptr(), 0, 10)
    torch.cuda.synchronize()
    assert not r.any()                                   # dropped, not written past the buffer
    assert c.comm_exchange_rows(0, -1, a.data_ptr(), 10, 0, 0, r.data_ptr(), 0, 10) == (10, 0)   # the communicator is still usable


def test_allreduce_stats_of_one_rank_is_the_identity(wtp, comm_ctx):
    st = dict(max_force=0.25, sum_u=12.5, sum_u2=40.0, n_move=17, argmin_i=3, argmin_j=9, argmin_r=0.125, n_fallback=2,
              n_uncovered=1, n_escaped=0)
    out = comm_ctx.comm_allreduce_stats(st)
    assert out == pytest.approx(st)
    assert out["argmin_i"] == 3 and out["argmin_j"] == 9


def test_comm_calls_before_init_are_state_errors(wtp):
    with wtp.Context(0) as c:
        with pytest.raises(wtp.WtpError):
            c.comm_exchange_rows(-1, -1, 0, 0, 0, 0, 0, 0, 0)
        with pytest.raises(wtp.WtpError):
            c.comm_allreduce_stats(dict(max_force=1.0))


def _one_rank_driver(wtp, n_total):
    import os
    import socket
    import sys

    import torch
    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from whatsthepoint_jl_amd import blocks, sharded

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    ctx = wtp.Context(0)
    s = float(n_total) ** (-1.0 / 3.0)

    def gen(first, n):
        t = torch.empty((n, 3), dtype=torch.float32, device="cuda")
        ctx.gen_uniform_dev(wtp.synth.SEED, first, n, 3, np.float32, t.data_ptr())
        return t

    xyz, gid, cuts = blocks.uniform_block_shard(gen, 0, (1, 1, 1), n_total, "cuda")
    eng = sharded.GpuEngine(ctx, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20)
    drv = blocks.BlockShardedRelax(eng, dist, xyz, gid, (1, 1, 1), cuts, sharded.ghost_width(n_total, 21), comm_device="cpu")
    return dist, ctx, eng, drv, s


def test_block_driver_over_the_abi_transport_one_rank(wtp, monkeypatch):
    """WTP_COMM=abi: the driver's statistics go through wtp_comm_allreduce_stats and its rounds through
    wtp_comm_exchange_rows.  One rank: the run must equal the single-domain session; the round itself is called
    directly with the rank as both neighbours (header row, split, payload come back as sent)."""
    import torch

    monkeypatch.setenv("WTP_COMM", "abi")
    n_total, iters = 200_000, 4
    dist, ctx, eng, drv, s = _one_rank_driver(wtp, n_total)
    try:
        assert drv.abi
        conv = [drv.step()["max_force"] for _ in range(iters)]
        mine = drv.gather_global(n_total).cpu().numpy()
        to_lo = torch.arange(40, dtype=torch.int32, device="cuda").reshape(10, 4)
        to_hi = (100 + torch.arange(24, dtype=torch.int32, device="cuda")).reshape(6, 4)
        from_lo, m_lo, from_hi, m_hi = drv._round_abi(0, 0, to_lo, 4, to_hi, 2)
        assert (m_lo, m_hi) == (4, 2) and torch.equal(from_lo, to_lo) and torch.equal(from_hi, to_hi)
        from_lo, m_lo, from_hi, m_hi = drv._round_abi(None, 0, to_lo, 0, to_hi[:0], 0)
        assert from_lo.shape[0] == 0 and m_lo == 0 and from_hi.shape[0] == 0 and m_hi == 0
    finally:
        eng.close()
        ctx.close()
        dist.destroy_process_group()
    x = wtp.synth.uniform(n_total, 3, np.float32)
    with wtp.Context(0) as c, c.relax(x, 0, s, dict(kind=2, beta=0.2, u0=1.0, gamma=3.0), 21, s / 2000, s / 20) as t:
        ref_conv, _ = t.run(iters, 1)
        ref = t.positions()
    # (the driver numbers its points in shard order, so sums inside a cell run in another order: rounding)
    assert np.abs(mine - ref).max() <= 2e-5 * s
    assert np.allclose(conv, ref_conv, rtol=1e-4)
