"""The C block driver (include/wtp.h: wtp_block_*; csrc/wtp_block.hip): the whole sharded iteration behind the ABI.

What one GPU can show: (a) a one-rank block session IS the plain session (bit for bit) and costs one host
synchronisation per iteration; (b) 2 x 2 x 2 ranks — eight contexts on the one GPU, as threads, rows carried by a
loopback transport — reproduce the single-domain run with the PRODUCT engine, with margin = 0 so that every crossing
migrates at once (direct routing to face, edge and corner owners); (c) a graded cloud with a device-evaluated
spacing law shards over count-median boxes; (d) the RCCL primitives of the exchange on a one-rank communicator
(messages to self).  More than one physical GPU is the driver's to run (bench.py --gpus N)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORCE = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
K = 21


def _single_domain(wtp, x, spacing, iters, alpha):
    with wtp.Context(0) as c:
        with c.relax(x, 0, spacing, FORCE, K, alpha / 100, alpha) as sess:
            conv = [sess.step(True)["max_force"] for _ in range(iters)]
            return np.array(conv), sess.positions()


def _run_blocks(wtp, x, boxes, spacing, iters, alpha, w, margin):
    import torch
    from whatsthepoint_jl_amd import blockc

    n = len(x)
    own = blockc.owner_of(x, boxes)
    nranks = len(boxes)

    def worker(rank, hub):
        torch.cuda.set_device(0)
        ctx = wtp.Context(0)
        try:
            idx = np.nonzero(own == rank)[0]
            drv = blockc.BlockRelax(ctx, rank, nranks, boxes, x[idx], idx.astype(np.int64), w, spacing, FORCE, K, alpha / 100,
                                    alpha, margin=margin, transport=blockc.loopback_transport(hub, rank))
            hist = [drv.step() for _ in range(iters)]
            xyz, gid = drv.owned()
            out = (xyz.cpu().numpy(), gid.cpu().numpy(), hist)
            drv.close()
            return out
        finally:
            ctx.close()

    res = blockc.run_threads(nranks, worker)
    p = np.full((n, 3), np.nan, dtype=np.float32)
    seen = np.zeros(n, dtype=np.int64)
    for xyz, gid, _ in res:
        p[gid] = xyz
        seen[gid] += 1
    assert (seen == 1).all(), "every point is owned by exactly one rank"
    return p, [r[2] for r in res]


def test_one_rank_block_session_is_the_plain_session(wtp):
    from whatsthepoint_jl_amd import blockc

    n, iters = 300_000, 5
    x = wtp.synth.uniform(n, 3, np.float32)
    s = float(n) ** (-1.0 / 3.0)
    conv0, p0 = _single_domain(wtp, x, s, iters, s / 20)
    boxes = blockc.orthtree_boxes(None, 1, equal_count=False)
    with wtp.Context(0) as ctx:
        drv = blockc.BlockRelax(ctx, 0, 1, boxes, x, None, 2.0 * s, s, FORCE, K, s / 2000, s / 20)
        hist = [drv.step() for _ in range(iters)]
        xyz, gid = drv.owned()
        drv.close()
    assert np.array_equal(gid.cpu().numpy(), np.arange(n))
    assert np.array_equal(xyz.cpu().numpy(), p0), "a one-rank block run equals the plain session bit for bit"
    assert np.array_equal(np.array([h["max_force"] for h in hist]), conv0)
    # the iteration's only host synchronisation is the read-back of {statistics, next row counts}
    assert hist[0]["host_syncs"] >= 2 and all(h["host_syncs"] == 1 for h in hist[1:]), [h["host_syncs"] for h in hist]
    assert all(h["n_move"] == n and h["n_ghost"] == 0 for h in hist)


def test_octants_with_the_product_engine_match_single_domain(wtp):
    """2 x 2 x 2: faces, edges and the corner exchange directly; margin 0 => a point migrates the moment it crosses."""
    from whatsthepoint_jl_amd import blockc

    n, iters = 240_000, 6
    x = wtp.synth.uniform(n, 3, np.float32)
    s = float(n) ** (-1.0 / 3.0)
    _, p0 = _single_domain(wtp, x, s, iters, s / 20)
    boxes = blockc.orthtree_boxes(None, 8, equal_count=False)
    p, hists = _run_blocks(wtp, x, boxes, s, iters, s / 20, w=2.2 * s, margin=0.0)
    err = np.abs(p - p0).max() / s
    assert err <= 2e-5, f"block run differs from the single-domain run by {err} spacings"
    assert all(h[-1]["n_peers"] == 7 for h in hists), "an octant exchanges with all seven others"
    assert sum(h["n_emigrated"] for hh in hists for h in hh) > 0, "margin 0: somebody crossed a face"
    assert sum(h["n_emigrated"] for hh in hists for h in hh) == sum(h["n_immigrated"] for hh in hists for h in hh)
    assert all(h["n_uncovered"] == 0 for hh in hists for h in hh)
    # host synchronisations per iteration: the statistics gather (the one a RCCL run has) + the two around the
    # loopback exchange's staging through host memory — migration, ghosts and classification add none
    assert all(h["host_syncs"] == 3 for hh in hists for h in hh[1:]), [[h["host_syncs"] for h in hh] for hh in hists]
    # every rank saw the same global statistics
    for i in range(iters):
        assert len({hh[i]["max_force"] for hh in hists}) == 1 and len({hh[i]["sum_u"] for hh in hists}) == 1
        assert hists[0][i]["n_move"] == n


def test_ranking_the_owned_points_under_the_exchange_changes_nothing(wtp, monkeypatch):
    """SURVEY 8e "interior work overlaps the exchange": in iterations where nobody migrates the first pass of the rebuild
    (owned points ranked into the cells) is issued before the ghost rows are in; the rebuild then ranks the appended
    rows only.  Same cells, same canonical order inside them: bit-identical to the serial chain (WTP_BLOCK_OVERLAP=0)."""
    from whatsthepoint_jl_amd import blockc

    n, iters = 200_000, 8
    x = wtp.synth.uniform(n, 3, np.float32)
    s = float(n) ** (-1.0 / 3.0)
    boxes = blockc.orthtree_boxes(None, 4, equal_count=False)
    p1, h1 = _run_blocks(wtp, x, boxes, s, iters, s / 20, w=2.2 * s, margin=0.5 * s)
    monkeypatch.setenv("WTP_BLOCK_OVERLAP", "0")
    p0, h0 = _run_blocks(wtp, x, boxes, s, iters, s / 20, w=2.2 * s, margin=0.5 * s)
    assert np.array_equal(p1, p0)
    assert [[h["max_force"] for h in hh] for hh in h1] == [[h["max_force"] for h in hh] for hh in h0]
    assert all(h["overlapped"] == 0 for hh in h0 for h in hh)
    quiet = [h for hh in h1 for h in hh[2:] if h["n_emigrated"] == 0 and h["n_immigrated"] == 0]
    assert quiet and sum(h["overlapped"] for h in quiet) >= len(quiet) // 2, [[h["overlapped"] for h in hh] for hh in h1]
    assert all(h["overlapped"] == 0 for hh in h1 for h in hh if h["n_emigrated"] or h["n_immigrated"])


def test_thin_ghost_layer_is_widened_and_the_step_repeated(wtp):
    from whatsthepoint_jl_amd import blockc

    n, iters = 120_000, 3
    x = wtp.synth.uniform(n, 3, np.float32)
    s = float(n) ** (-1.0 / 3.0)
    _, p0 = _single_domain(wtp, x, s, iters, s / 20)
    boxes = blockc.orthtree_boxes(None, 4, equal_count=False)
    p, hists = _run_blocks(wtp, x, boxes, s, iters, s / 20, w=0.5 * s, margin=0.05 * s)  # thinner than the law's support
    assert all(h[0]["redone"] == 1 and h[-1]["widened"] >= 1 for h in hists)
    assert np.abs(p - p0).max() / s <= 2e-5


def test_graded_cloud_with_a_device_law_over_count_median_boxes(wtp):
    """BASELINE.json configs[4] sharded: BoundaryLayerSpacing evaluated on the device (src/discretization/
    spacings.jl:121-133), boxes of equal point count; ghosts carry no spacing (only queries evaluate the law)."""
    from whatsthepoint_jl_amd import blockc

    n, iters = 150_000, 4
    x = wtp.synth.graded(n, 4.0, 0.2, np.float32)
    shell = (np.minimum(x, 1 - x).min(axis=1) < 0.02).sum()  # wall spacing from the density in the outer 2 % shell
    hw = float(((1 - 0.96 ** 3) / shell) ** (1 / 3))
    g = np.linspace(0.0, 1.0, 24, dtype=np.float32)
    u, v = np.meshgrid(g, g, indexing="ij")
    faces = []
    for a in range(3):
        for c in (0.0, 1.0):
            f = np.empty((u.size, 3), dtype=np.float32)
            f[:, a] = c
            f[:, (a + 1) % 3] = u.ravel()
            f[:, (a + 2) % 3] = v.ravel()
            faces.append(f)
    law = dict(kind=3, p0=hw, p1=4.0 * hw, p2=0.2, boundary=np.concatenate(faces))
    alpha = hw / 20
    _, p0 = _single_domain(wtp, x, law, iters, alpha)
    boxes = blockc.orthtree_boxes(x, 4, equal_count=True)
    counts = np.bincount(blockc.owner_of(x, boxes), minlength=4)
    assert counts.max() - counts.min() <= 4, counts
    p, hists = _run_blocks(wtp, x, boxes, law, iters, alpha, w=1.05 * 4.0 * hw, margin=0.1 * hw)
    err = np.abs(p - p0).max() / hw
    assert err <= 5e-5, f"graded block run differs from the single-domain run by {err} wall spacings"


def test_rccl_primitives_on_a_one_rank_communicator(wtp):
    """wtp_comm_exchange_peers / wtp_comm_allgather_dev through librccl: two messages to self in one group."""
    import ctypes as C

    import torch
    from whatsthepoint_jl_amd import _lib as L

    with wtp.Context(0) as ctx:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
        dev = torch.device("cuda", 0)
        a = torch.arange(4 * 1000, dtype=torch.int32, device=dev).reshape(1000, 4)
        b = (torch.arange(4 * 300, dtype=torch.int32, device=dev) * 7).reshape(300, 4)
        ra, rb = torch.zeros_like(a), torch.zeros_like(b)
        torch.cuda.synchronize()
        peers = (C.c_int * 2)(0, 0)
        sp = (C.c_void_p * 2)(a.data_ptr(), b.data_ptr())
        rp = (C.c_void_p * 2)(ra.data_ptr(), rb.data_ptr())
        ns = (C.c_int64 * 2)(1000, 300)
        L.check(ctx._h, ctx._lib.wtp_comm_exchange_peers(ctx._h, 2, peers, sp, ns, rp, ns))
        g_in = torch.arange(16, dtype=torch.int64, device=dev)
        g_out = torch.zeros(16, dtype=torch.int64, device=dev)
        L.check(ctx._h, ctx._lib.wtp_comm_allgather_dev(ctx._h, C.c_void_p(g_in.data_ptr()), C.c_void_p(g_out.data_ptr()), 128))
        ctx.set_stream(None)  # (synchronises the context's stream)
        assert torch.equal(ra, a) and torch.equal(rb, b) and torch.equal(g_in, g_out)
        # a one-rank block session over the RCCL transport path (no peers: the gather degenerates to a read-back)
        from whatsthepoint_jl_amd import blockc

        n = 100_000
        x = wtp.synth.uniform(n, 3, np.float32)
        s = float(n) ** (-1.0 / 3.0)
        drv = blockc.BlockRelax(ctx, 0, 1, blockc.orthtree_boxes(None, 1, False), x, None, 2 * s, s, FORCE, K, s / 2000, s / 20)
        out = drv.run(4)
        drv.close()
        assert out["host_syncs"] >= 5 and out["n_move"] == n
        ctx.comm_finalize()


def test_octants_two_million_points(wtp):
    """The rehearsal VERDICT r2 asked for: 2 x 2 x 2 ranks with the product engine, >= 2 M points, margin 0
    (all the migration routing runs); checked through properties that hold at any size."""
    from whatsthepoint_jl_amd import blockc

    n, iters = 2_000_000, 5
    x = wtp.synth.uniform(n, 3, np.float32)
    s = float(n) ** (-1.0 / 3.0)
    conv0, p0 = _single_domain(wtp, x, s, iters, s / 20)
    boxes = blockc.orthtree_boxes(None, 8, equal_count=False)
    p, hists = _run_blocks(wtp, x, boxes, s, iters, s / 20, w=2.2 * s, margin=0.0)
    assert np.abs(p - p0).max() / s <= 2e-5
    conv = np.array([h["max_force"] for h in hists[0]])
    assert np.allclose(conv, conv0, rtol=1e-5)
    assert sum(h["n_emigrated"] for hh in hists for h in hh) > 100


def _bench_line(args, env_extra=None):
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WTP_BENCH_REHEARSAL="1", **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, env=env,
                         timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_rehearsal_two_processes_over_gloo():
    """bench.py --gpus 2 on the one-GPU box: two rank processes, the C block driver over the host-callback transport
    (torch.distributed gloo), fixed total => "scaling": "strong"."""
    out = _bench_line(["--gpus", "2", "--steps", "4", "--warmup", "2", "--total-points", "600000", "--no-cpu"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert "C block driver" in out["config"]["sharding"]


def test_bench_rehearsal_eight_ranks_as_threads():
    """bench.py --gpus 8 in rehearsal mode: eight ranks as threads sharing the GPU."""
    out = _bench_line(["--gpus", "8", "--steps", "4", "--warmup", "2", "--total-points", "1600000", "--no-cpu"])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["value"] > 0
    assert out["block_info"]["n_peers"] == 7
