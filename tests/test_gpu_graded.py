"""Graded clouds (BASELINE config 5; SURVEY §8d "G-graded"): a 64x density contrast between wall
and bulk.  RadiusTopology in fp32 and fp64 with the tolerance cross-check the config names, and
the k-NN / sweep paths on the same cloud (they exercise the measured cell edge and the exact
hand-back path far more than a uniform cloud does)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(offsets, idx, i):
    return idx[offsets[i]:offsets[i + 1]]


def test_radius_fp32_vs_fp64_cross_check(O, wtp, ctx):
    n = 150_000
    x32 = wtp.synth.graded(n, 4.0, 0.2, np.float32)
    x64 = x32.astype(np.float64)                       # the same points, exactly
    shell = (np.minimum(x32, 1 - x32).min(axis=1) < 0.02).sum()
    hw = ((1 - 0.96 ** 3) / shell) ** (1 / 3)
    r = 2.5 * hw                                       # ~21 neighbours in the fine region, ~0.3 in the bulk
    o32, i32 = ctx.radius(x32, r)
    o64, i64 = ctx.radius(x64, r)
    c32, c64 = np.diff(o32), np.diff(o64)
    assert c32.max() > 30 and c32.min() == 0 and 8 < c32.mean() < 40   # list length varies by the grading
    # identical neighbour SETS except pairs with |d - r| <= 4 ulp_fp32(r)  
    tol = 4 * np.spacing(np.float32(r))
    differ = np.nonzero(c32 != c64)[0]
    same = np.nonzero(c32 == c64)[0]
    pick = same[:: max(len(same) // 3000, 1)]
    for i in pick:
        assert np.array_equal(np.sort(_rows(o32, i32, i)), np.sort(_rows(o64, i64, i)))
    assert len(differ) < 1e-3 * n
    for i in differ:
        a, b = set(_rows(o32, i32, i).tolist()), set(_rows(o64, i64, i).tolist())
        for j in a ^ b:
            d = np.sqrt(((x64[i] - x64[j]) ** 2).sum())
            assert abs(d - r) <= tol, (i, j, d, r)
    # and each precision against the oracle on the same cloud (exact, rows in canonical order)
    sub = np.arange(0, n, 7)
    wo, wi = O.radius(x32, r)
    assert np.array_equal(o32, wo) and np.array_equal(i32, wi)
    wo, wi = O.radius(x64, r)
    assert np.array_equal(o64, wo) and np.array_equal(i64, wi)
    assert len(sub) > 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_knn_on_graded_cloud_matches_oracle(O, wtp, ctx, dtype):
    x = wtp.synth.graded(80_000, 4.0, 0.2, dtype)
    idx, dist = ctx.knn(x, 21, return_dist=True)
    widx, wdist = O.knn(x, 21)
    assert np.array_equal(idx, widx) and np.array_equal(dist, wdist)


@pytest.mark.parametrize("n", [60_000, 300_000])   # the larger one: dead bricks passed over, runs taken in chunks, ball kernel
def test_sweep_on_graded_cloud_with_its_spacing_law(O, wtp, ctx, n):
    x = wtp.synth.graded(n, 4.0, 0.2, np.float32)
    shell = (np.minimum(x, 1 - x).min(axis=1) < 0.02).sum()
    hw = float(((1 - 0.96 ** 3) / shell) ** (1 / 3))
    m = int(1 / hw)
    g = (np.arange(m, dtype=np.float32) + 0.5) / m
    u, v = np.meshgrid(g, g, indexing="ij")
    faces = []
    for axis in range(3):
        for side in (0.0, 1.0):
            c = np.zeros((m * m, 3), np.float32)
            c[:, axis] = side
            c[:, (axis + 1) % 3] = u.ravel()
            c[:, (axis + 2) % 3] = v.ravel()
            faces.append(c)
    b = np.concatenate(faces)
    law = wtp.BoundaryLayerSpacing(b, at_wall=hw, bulk=4 * hw, layer_thickness=0.2)
    snap = np.concatenate([b, x])
    force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
    with ctx.relax(snap, len(b), law.desc(), force, 21, hw / 2000, hw / 20) as t:
        st = t.step(True)
        got = t.positions()
        sp = t.spacings()
    want_sp = O.spacing_boundary_layer(snap, b, hw, 4 * hw, 0.2)
    assert np.max(np.abs(sp - want_sp)) <= 4 * np.finfo(np.float32).eps * 4 * hw
    r = O.relax_sweep(snap, len(b), sp, 2, 0.2, 1.0, 3.0, 21, hw / 2000, hw / 20)
    err = np.abs(got - r["p"]).max(axis=1) / sp[len(b):]
    assert err.max() < 1e-4 and np.quantile(err, 0.999) < 1e-5
    assert st["n_move"] == n and 0 < st["n_fallback"] < 0.5 * n      # the coarse region takes the exact path
    assert abs(st["max_force"] - float(r["forces"].max())) <= 1e-4 * float(r["forces"].max())


def test_float64_sweep_on_graded_cloud_bit_exact_with_and_without_the_ball_kernel(O, wtp, monkeypatch):
    """Float64, ClippedSpacingForce, BoundaryLayerSpacing on the device: the queries whose support is wider than their cell
    take csrc/wtp_ball64.hip (eight lanes per query, the ball's points ranked and added in (d2, index) order) instead of the
    wave-per-query path — same positions to the last bit, and equal to the oracle's sequential evaluation."""
    n = 120_000
    x = wtp.synth.graded(n, 4.0, 0.2, np.float64)
    shell = (np.minimum(x, 1 - x).min(axis=1) < 0.02).sum()
    hw = float(((1 - 0.96 ** 3) / shell) ** (1 / 3))
    m = int(1 / hw)
    g = (np.arange(m, dtype=np.float64) + 0.5) / m
    u, v = np.meshgrid(g, g, indexing="ij")
    faces = []
    for axis in range(3):
        for side in (0.0, 1.0):
            c = np.zeros((m * m, 3), np.float64)
            c[:, axis] = side
            c[:, (axis + 1) % 3] = u.ravel()
            c[:, (axis + 2) % 3] = v.ravel()
            faces.append(c)
    b = np.concatenate(faces)
    law = wtp.BoundaryLayerSpacing(b, at_wall=hw, bulk=4 * hw, layer_thickness=0.2)
    snap = np.concatenate([b, x])
    force = dict(kind=2, beta=0.2, u0=1.0, gamma=3.0)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("WTP_BALL64", flag)
        with wtp.Context(0) as c:
            with c.relax(snap, len(b), law.desc(), force, 21, hw / 2000, hw / 20) as t:
                st = [t.step(True) for _ in range(3)]
                res[flag] = (t.positions(), t.point_data(), st, t.spacings())
    assert np.array_equal(res["1"][0], res["0"][0])
    assert np.array_equal(res["1"][1]["forces"], res["0"][1]["forces"]) and np.array_equal(res["1"][1]["nn_id"], res["0"][1]["nn_id"])
    assert res["1"][2][0]["n_fallback"] < res["0"][2][0]["n_fallback"] // 4, "the ball kernel finishes most of the hand-backs"
    assert res["1"][2][-1]["max_force"] == res["0"][2][-1]["max_force"]
    # one sweep against the oracle (its spacings are the device's, checked elsewhere)
    with wtp.Context(0) as c:
        with c.relax(snap, len(b), law.desc(), force, 21, hw / 2000, hw / 20) as t:
            t.step(True)
            got, sp = t.positions(), t.spacings()
    r = O.relax_sweep(snap, len(b), sp, 2, 0.2, 1.0, 3.0, 21, hw / 2000, hw / 20)
    assert np.array_equal(got, r["p"])
