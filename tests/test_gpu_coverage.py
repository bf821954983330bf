"""Coverage radius of sharded sessions (include/wtp.h: wtp_relax_set_coverage): the sweep must
count every movable point whose answer could depend on points farther than the radius."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORCE = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)


def _rk_and_nn(ctx, x, k):
    _, d = ctx.knn(x, k, include_self=True, return_dist=True)
    return d[:, k - 1].astype(np.float64), d[:, 1].astype(np.float64)


@pytest.mark.parametrize("full_select", [False, True])
def test_uncovered_count_bounds(wtp, monkeypatch, full_select):
    if full_select:
        monkeypatch.setenv("WTP_FULL_SELECT", "1")
    n, k = 80000, 21
    s = n ** (-1.0 / 3.0)
    x = wtp.synth.uniform(n, 3, np.float32, 21)
    c = wtp.Context(0)
    try:
        rk, nn = _rk_and_nn(c, x, k)
        z = x[:, 2].astype(np.float64)
        for lo, hi in ((0.0, 1.0), (-0.5 * s, 1.0 + 0.5 * s), (-2 * s, 1 + 2 * s), (-np.inf, 0.7), (0.3, np.inf),
                       (-6 * s, 1 + 6 * s)):
            with c.relax(x, 0, s, FORCE, k, s / 2000, s / 20) as t:
                t.set_coverage(2, lo, hi)
                st = t.step(True)
            d = np.minimum(z - lo, hi - z)                       # distance to the nearer end of the cover
            # what an answer rests on: the k-th neighbour (explicit selection), or — on every path that
            # certifies by counting, the exact wave path included — the law's support and the nearest neighbour
            most = int((np.maximum(rk, s) > d * (1 - 1e-5)).sum())
            least = int((np.maximum(nn, s) > d * (1 + 1e-5)).sum())
            assert least <= st["n_uncovered"] <= most, ((lo, hi), least, st["n_uncovered"], most)
            if np.isfinite(lo) and lo < -3 * s:
                assert st["n_uncovered"] == 0
            elif least > 0:
                assert st["n_uncovered"] > 0
        with c.relax(x, 0, s, FORCE, k, s / 2000, s / 20) as t:
            assert t.step(True)["n_uncovered"] == 0          # unlimited by default
            t.set_coverage(2, 0.0, 1.0)
            assert t.step(True)["n_uncovered"] > 0
            t.set_coverage(-1)
            assert t.step(True)["n_uncovered"] == 0
            with pytest.raises(ValueError):
                t.set_coverage(3, 0.0, 1.0)
            with pytest.raises(ValueError):
                t.set_coverage(1, 1.0, 0.0)
    finally:
        c.close()


def test_uncovered_fp64_exact_path(wtp, ctx):
    n, k = 20000, 21
    s = n ** (-1.0 / 3.0)
    x = wtp.synth.uniform(n, 3, np.float64, 4)
    rk, nn = _rk_and_nn(ctx, x, k)
    d = np.minimum(x[:, 0] + s, 1.0 + s - x[:, 0])
    for kind in (2, 1):     # clipped law: support-ball shortcut of the wave kernel; equilibrium law: the k-th neighbour
        with ctx.relax(x, 0, s, dict(kind=kind, beta=0.2, u0=1.0, gamma=3.0), k, s / 2000, s / 20) as t:
            t.set_coverage(0, -s, 1.0 + s)
            st = t.step(True)
        if kind == 1:
            least, most = int((rk > d * (1 + 1e-9)).sum()), int((rk > d * (1 - 1e-9)).sum())
        else:
            least, most = int((np.maximum(nn, s) > d * (1 + 1e-9)).sum()), int((np.maximum(rk, s) > d * (1 - 1e-9)).sum())
        assert 0 < least <= st["n_uncovered"] <= most, (kind, least, st["n_uncovered"], most)
