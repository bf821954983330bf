"""GPU parity of the consumers of the k-NN rows: PCA normals (src/normals.jl), the
orientation walk and split_surface! that sit on top of them, and the min-plus gradient limiter
(src/discretization/algorithms/octree.jl:677-717).

Normals: device vs oracle within a stated tolerance (|n_gpu . n_oracle| >= 1 - 1e-5 in fp32, 1e-12 in
fp64, on points whose two smallest covariance eigenvalues are separated; the sign convention is the
library's own — the reference leaves the sign to orient_normals!, i.e. "parity unpinned" for it).
Gradient limiter: bit for bit, sweep count included."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fib_sphere(N=100):
    ids = np.arange(0.0, N + 0.5 + 1e-9, 1.0)
    phi = np.arccos(1 - 2 * ids / N)
    th = np.pi * (1 + np.sqrt(5)) * ids
    return np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], axis=1)


@pytest.mark.parametrize("dtype,dim,n,k", [(np.float32, 3, 200000, 9), (np.float64, 3, 50000, 5), (np.float64, 2, 30000, 7),
                                           (np.float32, 2, 30000, 3)])
def test_pca_normals_match_oracle(ctx, O, dtype, dim, n, k):
    rng = np.random.default_rng(4)
    p = rng.random((n, dim))
    if dim == 3:
        p[:, 2] = 0.3 * np.sin(3 * p[:, 0]) * np.cos(2 * p[:, 1]) + 0.002 * rng.standard_normal(n)
    else:
        p[:, 1] = 0.3 * np.sin(3 * p[:, 0]) + 0.002 * rng.standard_normal(n)
    p = np.ascontiguousarray(p.astype(dtype))
    got = ctx.pca_normals(p, k)
    ref = O.pca_normals(p, k)
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5 if dtype == np.float32 else 1e-12)
    dots = (got.astype(np.float64) * ref.astype(np.float64)).sum(axis=1)
    tol = 1e-5 if dtype == np.float32 else 1e-12
    # identical arithmetic, so nearly every point agrees to rounding INCLUDING the sign; where the two
    # smallest eigenvalues nearly tie the direction is ill-conditioned and only |dot| of the pair is bounded
    assert (dots > 1 - tol).mean() > 0.999


def test_normals_reference_known_answers_on_device(ctx, wtp):
    """test/normals.jl:1-50 (2-D circle) and :54-110 (Fibonacci sphere): compute, orient, update round trip."""
    th = np.arange(8) * np.pi / 4
    circle = np.stack([np.cos(th), np.sin(th)], axis=1)
    nrm = wtp.compute_normals(circle, k=3, ctx=ctx)
    assert np.allclose(np.abs((nrm * circle).sum(axis=1)), 1.0, atol=1e-12)
    wtp.orient_normals(nrm, circle, k=3, ctx=ctx)
    assert np.allclose(nrm, circle, atol=1e-12)            # outward: the highest point faces up, the rest follow
    surf = wtp.PointSurface(circle, np.zeros_like(circle))
    cloud = wtp.PointCloud(wtp.PointBoundary(surf))
    wtp.update_normals(cloud.boundary["surface1"], k=3, ctx=ctx)
    wtp.orient_normals(cloud, k=3, ctx=ctx)
    original = cloud.boundary["surface1"].normals.copy()
    assert np.allclose(original, circle, atol=1e-12)
    rng = np.random.default_rng(0)
    junk = rng.standard_normal(circle.shape)
    cloud.boundary["surface1"].normals[:] = junk / np.linalg.norm(junk, axis=1)[:, None]
    assert not np.allclose(cloud.boundary["surface1"].normals, original)
    wtp.update_normals(cloud.boundary["surface1"], k=3, ctx=ctx)
    wtp.orient_normals(cloud, k=3, ctx=ctx)
    assert np.allclose(cloud.boundary["surface1"].normals, original, atol=1e-12)

    sph = _fib_sphere()
    ns = wtp.compute_normals(sph, k=5, ctx=ctx)
    ang = np.degrees(np.arccos(np.clip(np.abs((ns * sph).sum(axis=1)), 0, 1)))
    assert ang.max() < 10.0
    wtp.orient_normals(ns, sph, k=5, ctx=ctx)
    ang = np.degrees(np.arccos(np.clip((ns * sph).sum(axis=1), -1, 1)))
    assert ang.max() < 10.0                                 # every normal now points outward
    assert len(wtp.compute_normals(wtp.PointSurface(sph, np.zeros_like(sph)), k=500, ctx=ctx)) == len(sph)  # k > n is clamped


def test_split_surface_cube(ctx, wtp):
    """split_surface! (src/surface_operations.jl:58-94; test/surface_operations.jl:69-145): the face centres of
    a cube with their outward normals split into the 6 faces at 80 degrees; a finer angle never merges."""
    m = 12
    g = (np.arange(m) + 0.5) / m
    u, v = np.meshgrid(g, g, indexing="ij")
    pts, nrm = [], []
    for axis in range(3):
        for side in (0.0, 1.0):
            c = np.zeros((m * m, 3))
            c[:, axis], c[:, (axis + 1) % 3], c[:, (axis + 2) % 3] = side, u.ravel(), v.ravel()
            nn = np.zeros_like(c)
            nn[:, axis] = 1.0 if side else -1.0
            pts.append(c)
            nrm.append(nn)
    pts, nrm = np.concatenate(pts), np.concatenate(nrm)
    perm = np.random.default_rng(2).permutation(len(pts))
    pts, nrm = pts[perm], nrm[perm]
    bnd = wtp.PointBoundary(pts, nrm, np.full(len(pts), 1.0 / (m * m)))
    wtp.split_surface(bnd, np.radians(80.0), ctx=ctx)
    assert len(bnd.surfaces) == 6 and len(bnd) == 6 * m * m
    assert sorted(bnd.surfaces) == [f"surface{i}" for i in range(1, 7)]
    for s in bnd.surfaces.values():
        assert len(s) == m * m and np.allclose(s.normals, s.normals[0]) and s.areas is not None
    # surfaces are numbered by their first point in the original order
    firsts = [int(np.nonzero((pts == s.points()[0]).all(axis=1))[0][0]) for s in bnd.surfaces.values()]
    assert firsts == sorted(firsts)
    with pytest.raises(AssertionError):
        wtp.split_surface(bnd, np.radians(80.0), ctx=ctx)            # more than one surface, no target
    with pytest.raises(AssertionError):
        wtp.split_surface(bnd, np.radians(80.0), target="nonexistent", ctx=ctx)
    cloud = wtp.PointCloud(wtp.PointBoundary(pts, nrm, np.full(len(pts), 1.0 / (m * m))))
    wtp.split_surface(cloud, np.radians(80.0), target="surface1", ctx=ctx)
    assert len(cloud.boundary.surfaces) == 6


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gradient_limit_bit_exact(ctx, O, wtp, dtype):
    rng = np.random.default_rng(6)
    n = 40000
    c = np.ascontiguousarray(rng.random((n, 3)).astype(dtype))
    h0 = (0.05 + 0.5 * rng.random(n)).astype(dtype)
    h0[rng.integers(0, n, 25)] = dtype(0.004)                 # a few fine sources
    for g, tol, cap in ((0.2, 1e-3, 2000), (0.05, 1e-6, 2000), (0.3, 1e-3, 7)):
        got, sweeps = wtp.gradient_limit_field(c, h0, g, k=12, tol=tol, max_sweeps=cap, return_sweeps=True, ctx=ctx)
        ref, rs = O.gradient_limit(c, h0, dtype(g), k=12, tol=tol, max_sweeps=cap)
        assert sweeps == rs
        assert np.array_equal(got, ref)
        assert (got <= h0).all() and (got < h0).mean() > 0.5
    # closed form on a chain (see tests/test_oracle_kat.py)
    m = 40
    chain = np.stack([np.arange(m, dtype=np.float64), np.zeros(m), np.zeros(m)], axis=1).astype(dtype)
    hc = np.full(m, 5.0, dtype=dtype)
    hc[7] = 1.0
    got, sweeps = wtp.gradient_limit_field(chain, hc, 0.25, k=3, tol=1e-12, max_sweeps=100, return_sweeps=True, ctx=ctx)
    assert sweeps == 16 and np.allclose(got, np.minimum(5.0, 1.0 + 0.25 * np.abs(np.arange(m) - 7)), atol=1e-6)
    with pytest.raises(wtp.WtpArgumentError):
        ctx.gradient_limit(c, h0[:-1], 0.2)
    with pytest.raises(wtp.WtpArgumentError):
        ctx.pca_normals(c, 1)
