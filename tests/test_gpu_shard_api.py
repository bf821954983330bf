"""C-ABI entry points of the sharded session (include/wtp.h: wtp_relax_layers_dev,
wtp_relax_set_fixed_dev, wtp_set_stream) against numpy selections and against one-shot sessions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORCE = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)


def _torch():
    import torch

    return torch


def _dev(a):
    torch = _torch()
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rows4(x):
    r = np.zeros((len(x), 4), dtype=x.dtype)
    r[:, :3] = x
    return r


@pytest.mark.parametrize("stale", [False, True])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_layers_match_numpy_selection(wtp, ctx, dtype, stale):
    torch = _torch()
    n, n_fixed, k = 60000, 1500, 21
    s = n ** (-1.0 / 3.0)
    x = wtp.synth.uniform(n, 3, dtype, 11)
    sess = ctx.relax(x, n_fixed, s, FORCE, k, s / 2000, s / 20)
    try:
        sess.step(True)
        sess.step(not stale)                        # stale: a second sweep on the same grid (rebuild_every = 2)
        if stale:
            sess.step(False)
        pos = sess.positions()                      # movable points, movable-index order
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        idt = torch.int32 if dtype == np.float32 else torch.int64
        cap = n
        lo = torch.zeros((cap, 4), dtype=tdt, device="cuda")
        hi = torch.zeros((cap, 4), dtype=tdt, device="cuda")
        args = (2, 0.2, 0.75, 0.01, 0.995)
        cnt = sess.layers_dev(*args, lo.data_ptr(), hi.data_ptr(), cap)
        torch.cuda.synchronize()
        z = pos[:, 2]
        want_lo, want_hi = np.nonzero(z < dtype(0.2))[0], np.nonzero(z >= dtype(0.75))[0]
        assert cnt == (len(want_lo), len(want_hi), int((z < dtype(0.01)).sum()), int((z >= dtype(0.995)).sum()))
        for buf, want, m in ((lo, want_lo, cnt[0]), (hi, want_hi, cnt[1])):
            rows = buf[:m].cpu()
            idx = rows[:, 3].contiguous().view(idt).numpy().astype(np.int64)
            assert np.array_equal(np.sort(idx), want)                 # the same set, each point once
            assert np.array_equal(rows[:, :3].numpy(), pos[idx])      # carrying its current position
        # deterministic order, and a short buffer only truncates
        lo2 = torch.zeros_like(lo)
        hi2 = torch.zeros_like(hi)
        assert sess.layers_dev(*args, lo2.data_ptr(), hi2.data_ptr(), cap) == cnt
        torch.cuda.synchronize()
        assert torch.equal(lo2[: cnt[0]], lo[: cnt[0]]) and torch.equal(hi2[: cnt[1]], hi[: cnt[1]])
        small = 1000
        lo3 = torch.full((small + 8, 4), -7.0, dtype=tdt, device="cuda")
        hi3 = torch.full((small + 8, 4), -7.0, dtype=tdt, device="cuda")
        assert sess.layers_dev(*args, lo3.data_ptr(), hi3.data_ptr(), small) == cnt
        torch.cuda.synchronize()
        assert torch.equal(lo3[:small], lo[:small]) and bool((lo3[small:] == -7.0).all())
        assert torch.equal(hi3[:small], hi[:small]) and bool((hi3[small:] == -7.0).all())
    finally:
        sess.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_set_fixed_equals_fresh_session(wtp, ctx, dtype):
    """Swapping the fixed head of a resident session == starting over on [new head ; movable]."""
    n_own, k = 40000, 21
    s = (n_own * 1.3) ** (-1.0 / 3.0)
    own = wtp.synth.uniform(n_own, 3, dtype, 5)
    g1 = wtp.synth.uniform(9000, 3, dtype, 6)
    g2 = wtp.synth.uniform(12000, 3, dtype, 8)
    g2[:, 2] *= 0.3

    def one_shot(ghosts, movable):
        with ctx.relax(np.concatenate([ghosts, movable]), len(ghosts), s, FORCE, k, s / 2000, s / 20) as t:
            st = t.step(True)
            return t.positions(), st

    sess = ctx.relax(own, 0, s, FORCE, k, s / 2000, s / 20)
    try:
        d1 = _dev(_rows4(g1))
        sess.set_fixed_dev(d1.data_ptr(), len(g1))
        st_a = sess.step(True)
        pa = sess.positions()
        d2 = _dev(_rows4(g2))
        sess.set_fixed_dev(d2.data_ptr(), len(g2))
        st_b = sess.step(True)
        pb = sess.positions()
        sess.set_fixed_dev(0, 0)                      # and back to no fixed points at all
        st_c = sess.step(True)
        pc = sess.positions()
    finally:
        sess.close()
    ra, sa = one_shot(g1, own)
    rb, sb = one_shot(g2, ra)
    rc, sc = one_shot(g2[:0], rb)
    for got, ref, st, sr in ((pa, ra, st_a, sa), (pb, rb, st_b, sb), (pc, rc, st_c, sc)):
        assert np.array_equal(got, ref)               # same ids, same canonical state: bit for bit
        assert st["max_force"] == sr["max_force"] and st["n_move"] == sr["n_move"] == n_own


def test_lent_stream_gives_the_same_result(wtp):
    torch = _torch()
    n, k = 50000, 21
    s = n ** (-1.0 / 3.0)
    x = wtp.synth.uniform(n, 3, np.float32, 3)
    outs = []
    for lend in (False, True):
        c = wtp.Context(0)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            if lend:
                c.set_stream(side.cuda_stream)
            d = torch.from_numpy(x).cuda()
            side.synchronize()
            t = c.relax(None, 0, s, FORCE, k, s / 2000, s / 20, device_ptr=(d.data_ptr(), n, 3, np.float32))
            t.run(3, 1)
            outs.append(t.positions())
            t.close()
            if lend:
                c.set_stream(None)
        c.close()
    assert np.array_equal(outs[0], outs[1])


def test_set_fixed_rejects_per_point_spacing(wtp, ctx):
    x = wtp.synth.uniform(2000, 3, np.float32, 1)
    sp = np.full(2000, 0.08, np.float32)
    with ctx.relax(x, 0, sp, FORCE, 21, 1e-5, 1e-3) as t:
        with pytest.raises(wtp.WtpError):
            t.set_fixed_dev(0, 0)


def test_reading_the_state_between_set_fixed_and_step(wtp, ctx):
    """wtp_relax_set_fixed_dev only appends the new head and leaves the clean-up to the next hash
    build; any entry point that reads the state in between has to materialise it first."""
    torch = _torch()
    n_own, k = 30000, 21
    s = n_own ** (-1.0 / 3.0)
    own = wtp.synth.uniform(n_own, 3, np.float32, 5)
    g1 = wtp.synth.uniform(4000, 3, np.float32, 6)
    g2 = wtp.synth.uniform(6000, 3, np.float32, 8)
    with ctx.relax(own, 0, s, FORCE, k, s / 2000, s / 20) as sess:
        d1 = _dev(_rows4(g1))
        sess.set_fixed_dev(d1.data_ptr(), len(g1))           # first call: rewrites once (buffers had no room)
        sess.step(True)
        p1 = sess.positions()
        d2 = _dev(_rows4(g2))
        sess.set_fixed_dev(d2.data_ptr(), len(g2))           # pending view
        assert np.array_equal(sess.positions(), p1)          # read-back in between: same owned positions
        lo = torch.zeros((n_own, 4), dtype=torch.float32, device="cuda")
        hi = torch.zeros((n_own, 4), dtype=torch.float32, device="cuda")
        cnt = sess.layers_dev(2, 0.3, 0.6, -1.0, 2.0, lo.data_ptr(), hi.data_ptr(), n_own)
        assert cnt[0] == int((p1[:, 2] < np.float32(0.3)).sum()) and cnt[1] == int((p1[:, 2] >= np.float32(0.6)).sum())
        d3 = _dev(_rows4(g1))
        sess.set_fixed_dev(d3.data_ptr(), len(g1))           # and a second replacement before any step
        st = sess.step(True)
        p2 = sess.positions()
    with ctx.relax(np.concatenate([g1, p1]), len(g1), s, FORCE, k, s / 2000, s / 20) as ref:
        st_ref = ref.step(True)
        assert np.array_equal(p2, ref.positions()) and st["max_force"] == st_ref["max_force"]
