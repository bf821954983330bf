"""isinside — the post-filter of the volume-only repel (src/repel.jl:90; src/isinside.jl).
CPU part: the oracle against the reference's own known answers (test/isinside.jl:1-75), with the
boundary elements of its test surface box.stl as a committed fixture (tests/golden/box_surface.npz,
made by tools/make_golden.py).  GPU part: libwtp against the oracle."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _box():
    z = np.load(os.path.join(GOLD, "box_surface.npz"))
    return z["centroid"], z["normal"], z["area"]


# test/isinside.jl:59-73 (3-D PointBoundary / PointCloud on box.stl)
BOX_KAT = [((0.5, 0.5, 0.5), True), ((0.5, 0.5, -0.5), False), ((0.5, 0.5, -0.001), False),
           ((12.5, 12.5, 12.5), True), ((12.5, 12.5, -10.0), False), ((30.0, 12.5, 12.5), False)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_box_known_answers(O, dtype):
    c, nrm, a = _box()
    assert c.shape == (46786, 3) and abs(float(a.sum()) - 3750.0) < 1e-2
    t = np.array([p for p, _ in BOX_KAT], dtype=dtype)
    inside, g = O.isinside_greens(t, c, nrm, a)
    assert inside.tolist() == [w for _, w in BOX_KAT]
    assert abs(g[3] + 4 * np.pi) < 1e-2          # deep inside: -4 pi (src/isinside.jl:92-93)


def test_oracle_unit_square_known_answers(O):
    # test/isinside.jl:1-14
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float64)
    t = np.array([[0.5, 0.5], [0.5, 1.5], [0.5, 1 + np.finfo(np.float64).eps], [1.5, 0.5], [0.5, -0.5]])
    inside, s = O.isinside_winding(t, sq)
    assert inside.tolist() == [True, False, False, False, False]
    assert abs(abs(s[0]) - 2 * np.pi) < 1e-12
    # :16-37 (2x2 and 3x3 squares), and a coincident point counts as inside (:22-24 of src)
    for L, tin, tout in ((2.0, (1.0, 1.0), [(3.0, 1.0), (1.0, -1.0)]), (3.0, (1.5, 1.5), [(4.0, 1.5), (1.5, -1.0)])):
        poly = np.array([[0, 0], [L, 0], [L, L], [0, L]], dtype=np.float64)
        ins, _ = O.isinside_winding(np.array([tin] + tout), poly)
        assert ins.tolist() == [True, False, False]
    ins, _ = O.isinside_winding(np.array([[1.0, 0.0]]), sq)
    assert ins.tolist() == [True]


def test_host_polygon_validation(wtp):
    # test/isinside.jl:51-59: unordered points / fewer than 3 points throw ArgumentError
    from whatsthepoint_jl_amd import inside as I

    with pytest.raises(ValueError):
        I.validate_polygon_ordering(np.array([[0, 0], [1, 1], [1, 0], [0, 1]], dtype=np.float64))
    with pytest.raises(ValueError):
        I.validate_polygon_ordering(np.array([[0, 0], [1, 0]], dtype=np.float64))
    I.validate_polygon_ordering(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float64))


# ---- GPU ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gpu_greens_matches_oracle(O, wtp, ctx, dtype):
    c, nrm, a = _box()
    rng = np.random.default_rng(5)
    t = np.concatenate([
        np.array([p for p, _ in BOX_KAT], dtype=np.float64),
        rng.uniform(-8, 33, size=(6000, 3)),                      # in and around the 25-cube
        c[:50].astype(np.float64) + rng.normal(0, 0.05, (50, 3)),  # hugging the surface
        c[100:103].astype(np.float64),                            # coincident with elements: NaN -> outside
    ]).astype(dtype)
    inside, g = ctx.isinside_greens(t, c, nrm, a, return_g=True)
    want, g0 = O.isinside_greens(t, c, nrm, a)
    assert inside[:6].tolist() == [w for _, w in BOX_KAT]
    assert not inside[-3:].any() and np.isnan(g[-3:]).all() and np.isnan(g0[-3:]).all()
    ok = ~np.isnan(g0)
    tol = (2e-4 if dtype == np.float32 else 1e-9) * np.maximum(1.0, np.abs(g0[ok]))
    assert np.all(np.abs(g[ok].astype(np.float64) - g0[ok]) <= tol)
    near = np.abs(g0.astype(np.float64) + 2 * np.pi) < 1e-3       # only threshold-grazing points may differ
    assert np.array_equal(inside[~near], want[~near])
    assert 0.15 < inside.mean() < 0.35                            # (25/41)^3 of the sample is inside


@pytest.mark.gpu
def test_gpu_greens_small_n_large_m_chunks(O, ctx):
    c, nrm, a = _box()                                            # few test points: the element range is split
    t = np.array([[12.5, 12.5, 12.5], [40.0, 1.0, 1.0], [1.0, 24.0, 3.0]], dtype=np.float32)
    inside, g = ctx.isinside_greens(t, c, nrm, a, return_g=True)
    want, g0 = O.isinside_greens(t, c, nrm, a)
    assert inside.tolist() == want.tolist() == [True, False, True]
    assert np.allclose(g, g0, rtol=0, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gpu_winding_matches_oracle(O, ctx, dtype):
    th = np.linspace(0, 2 * np.pi, 700, endpoint=False)
    poly = np.stack([(2 + 0.5 * np.cos(5 * th)) * np.cos(th), (2 + 0.5 * np.cos(5 * th)) * np.sin(th)], 1).astype(dtype)
    rng = np.random.default_rng(2)
    t = np.concatenate([rng.uniform(-3, 3, size=(5000, 2)).astype(dtype), poly[10:12]])
    inside, s = ctx.isinside_winding(t, poly, return_sum=True)
    want, s0 = O.isinside_winding(t, poly)
    assert inside[-2:].all() and want[-2:].all()                  # coincident with polygon points
    assert np.allclose(s[:-2], s0[:-2], rtol=0, atol=2e-3 if dtype == np.float32 else 1e-10)
    clear = np.abs(np.abs(s0.astype(np.float64)) - np.pi) > 1.0   # |sum| is ~0 or ~2 pi away from the polygon line
    assert np.array_equal(inside[clear], want[clear]) and clear.mean() > 0.98
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=dtype)
    tt = np.array([[0.5, 0.5], [0.5, 1.5], [1.5, 0.5], [0.5, -0.5]], dtype=dtype)
    assert ctx.isinside_winding(tt, sq).tolist() == [True, False, False, False]
    with pytest.raises(ValueError):
        ctx.isinside_winding(tt, sq[:2])


@pytest.mark.gpu
def test_repel_filters_points_outside_the_box(O, wtp):
    """repel's tail (src/repel.jl:88-94): survivors = the relaxed volume points that are inside."""
    c, nrm, a = _box()
    bnd = wtp.PointBoundary(c, nrm, a)
    rng = np.random.default_rng(9)
    vol_in = rng.uniform(1.0, 24.0, size=(4000, 3)).astype(np.float32)
    vol_out = np.array([[30.0, 12.0, 12.0], [12.0, -4.0, 12.0], [-3.0, -3.0, 28.0]], dtype=np.float32)
    cloud = wtp.PointCloud(bnd, wtp.PointVolume(np.concatenate([vol_in, vol_out])))
    assert wtp.isinside(np.array([12.5, 12.5, 12.5]), cloud) is True
    assert wtp.isinside(np.array([30.0, 12.5, 12.5]), bnd) is False
    with pytest.raises(TypeError):
        wtp.isinside(np.array([12.5, 12.5, 12.5]), bnd["surface1"])      # test/isinside.jl:75-80
    kw = dict(max_iters=3, stall_after=0, tol=0.0)
    everyone = wtp.repel(cloud, wtp.ConstantSpacing(1.5), inside=lambda q: np.ones(len(q), bool), **kw).volume.points()
    assert len(everyone) == len(vol_in) + 3
    want, _ = O.isinside_greens(everyone, c, nrm, a)       # the wall is 4x denser than the spacing asked for:
    assert 3 <= (~want).sum() < 50                         # a few points get thrown across it, as in the reference
    new = wtp.repel(cloud, wtp.ConstantSpacing(1.5), **kw)
    p = new.volume.points()
    assert np.array_equal(p, everyone[want])               # survivors, in order (filter keeps order)
    assert p.min() > 0.0 and p.max() < 25.0
    assert isinstance(new.topology, wtp.NoTopology) and new.boundary is cloud.boundary
