"""RadiusTopology where rows are long: the brick-staged wave-per-query kernel (csrc/wtp_radb.hip) and what it hands back.
fp32 — bricks whose cells hold more than ~7.4 points; fp64 — every brick.  Rows bit-exact against the CPU oracle (canonical
(d2, index) order), and identical with the kernel switched off (WTP_RADIUS_DENSE=0: the wave-per-query path alone)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,pairs", [(60_000, 12), (60_000, 60), (40_000, 110), (30_000, 200)])
def test_dense_rows_match_oracle(ctx, O, wtp, dtype, n, pairs):
    # uniform cloud, r chosen for `pairs` neighbours per point: 12 (short rows: the one-sequence scan), 60 (one network of
    # two entries per lane at the wall of a graded cloud), 110 (near the wave's list of 128: some rows are handed back),
    # 200 (all rows beyond the list: the wave kernel's)
    x = wtp.synth.uniform(n, 3, dtype, 20261004 + pairs)
    r = float((pairs / n / 4.18879) ** (1.0 / 3.0))
    off, idx = ctx.radius(x, r)
    ooff, oidx = O.radius(x, r, "kdtree")
    assert np.array_equal(off, ooff)
    assert np.array_equal(idx, oidx)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_graded_cluster_and_lattice(ctx, O, wtp, dtype):
    # a dense cluster inside a sparse cloud (bricks beyond the kernel's LDS area next to empty ones), coincident points
    # (d2 = 0 ties ranked by index) and a lattice (every shell a mass tie in d2)
    x = wtp.synth.uniform(30_000, 3, dtype, 77)
    x[5_000:11_000] = 0.5 + 0.02 * (x[5_000:11_000] - 0.5)
    x[100:130] = x[100]
    r = 0.012
    off, idx = ctx.radius(x, r)
    ooff, oidx = O.radius(x, r, "kdtree")
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)
    g = np.stack(np.meshgrid(*[np.arange(24, dtype=dtype)] * 3, indexing="ij"), -1).reshape(-1, 3)
    off, idx = ctx.radius(g, 2.5)  # 80 neighbours in the interior, all on five distance shells
    ooff, oidx = O.radius(g, 2.5, "kdtree")
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)


def test_two_dimensional_cloud_fp64(ctx, O, wtp):
    x = wtp.synth.uniform(50_000, 2, np.float64, 5)
    r = float((40 / 50_000 / np.pi) ** 0.5)
    off, idx = ctx.radius(x, r)
    ooff, oidx = O.radius(x, r, "kdtree")
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_same_rows_with_the_kernel_switched_off(wtp, dtype, monkeypatch):
    x = wtp.synth.graded(150_000, dtype=dtype, seed=11)
    shell = int((np.minimum(x, 1 - x).min(axis=1) < 0.02).sum())
    hw = float(((1 - 0.96 ** 3) / shell) ** (1.0 / 3.0))
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("WTP_RADIUS_DENSE", flag)
        with wtp.Context(0) as c:
            res[flag] = c.radius(x, 2.5 * hw)
    assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1])
    assert res["1"][0][-1] > 20 * len(x), "rows long enough to reach the dense kernel"
