"""No-GPU checks: the C-ABI library loads and exports every symbol include/wtp.h declares;
the product path fails loudly without a device; host-side mirror logic (force laws, containers,
argument errors) behaves like the reference's."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "wtp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wtp_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(wtp):
    syms = _declared_symbols()
    assert len(syms) >= 20
    lib = ctypes.CDLL(wtp.SO_PATH)
    for s in syms:
        assert hasattr(lib, s), f"libwtp.so lacks {s} declared in include/wtp.h"
    from whatsthepoint_jl_amd import _lib

    assert set(syms) == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    assert wtp.load_library().wtp_version().startswith(b"wtp-mi355x")


def test_no_cpu_fallback_in_product_path():
    # the package never imports the oracle, and has no CPU implementation of the path
    pkg = os.path.join(ROOT, "whatsthepoint.jl_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f


def test_context_fails_loudly_without_gpu(wtp):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(wtp.WtpError) as e:
        wtp.Context(0)
    assert e.value.code == 5 and "no CPU path" in str(e.value)


def test_force_models_mirror_reference(wtp):
    # test/repel.jl:117-183 on the host-side compute_force
    cf = wtp.compute_force
    m1, m2, m3 = wtp.InverseDistanceForce(0.2), wtp.SpacingEquilibriumForce(0.2), wtp.ClippedSpacingForce(0.2)
    for u in (0.0, 0.5, 1.0, 2.0):
        assert cf(m1, u) == pytest.approx(1 / (u * u + 0.2) ** 2)
    assert cf(m2, 1.0) == 0.0 and cf(m2, 0.5) > 0 > cf(m2, 2.0)
    assert m3.u0 == 1.0
    for u in (0.0, 0.3, 0.7, 0.99):
        assert cf(m3, u) == pytest.approx(cf(m2, u)) and cf(m3, u) > 0
    assert cf(m3, 1.0) == cf(m3, 1.5) == cf(m3, 10.0) == 0.0
    m4 = wtp.ClippedSpacingForce(0.2, 0.8)
    assert cf(m4, 0.79) > 0 and cf(m4, 0.8) == 0.0
    assert wtp.InverseDistanceForce().beta == wtp.SpacingEquilibriumForce().beta == wtp.ClippedSpacingForce().beta == 0.2
    m5 = wtp.StrongSpacingForce(0.2, 3.0)
    for u in (0.0, 0.5, 1.0, 2.0):
        assert cf(m5, u) == pytest.approx((1 - u * u) / (u * u + 0.2) ** 3)
        assert cf(wtp.StrongSpacingForce(0.2, 2.0), u) == pytest.approx(cf(m2, u))
    assert wtp.StrongSpacingForce().gamma == 3.0 and wtp.StrongSpacingForce(0.5).gamma == 3.0


def test_containers_and_index_space(wtp):
    b = np.random.default_rng(0).random((7, 3))
    v = np.random.default_rng(1).random((5, 3))
    cloud = wtp.PointCloud(wtp.PointBoundary(b), wtp.PointVolume(v))
    p = wtp.points(cloud)
    assert len(cloud) == 12 and np.array_equal(p[:7], b) and np.array_equal(p[7:], v)  # boundary first
    assert isinstance(cloud.topology, wtp.NoTopology) and not wtp.hastopology(cloud) and wtp.isvalid(cloud.topology)
    with pytest.raises(wtp.WtpArgumentError):
        wtp.neighbors(cloud)                                       # "NoTopology has no neighbors"
    assert wtp.rebuild_topology(cloud) is None                     # no-op (test/topology.jl:97-104)
    assert repr(wtp.KNNTopology(np.zeros((3, 2), np.int32), 2)) == "KNNTopology(k=2)"
    assert repr(wtp.NoTopology()) == "NoTopology()"
    assert "├─k: 2" in wtp.KNNTopology(np.zeros((3, 2), np.int32), 2).show()


def test_repel_argument_errors_before_touching_the_gpu(wtp):
    b = np.random.default_rng(0).random((7, 3))
    cloud = wtp.PointCloud(wtp.PointBoundary(b), wtp.PointVolume(b + 0.1))
    with pytest.raises(wtp.WtpArgumentError):
        wtp.repel(cloud, 0.1, rebuild_every=0)                     # test/repel.jl:466


def test_stl_reader_matches_reference_fixture(wtp):
    path = "/root/reference/test/data/box.stl"
    if not os.path.exists(path):
        pytest.skip("reference checkout not mounted here")
    c = wtp.stl.face_centroids(path)
    assert c.shape == (46786, 3) and c.dtype == np.float32          # SURVEY.md §4 fixtures
    assert c.min() >= 0 and c.max() <= 25


def test_graded_cloud_generator(wtp):
    g = wtp.synth.graded(20000)
    assert g.shape == (20000, 3) and g.min() >= 0 and g.max() < 1
    d = np.minimum(g, 1 - g).min(1)
    near, far = (d < 0.03).sum() / (1 - 0.94 ** 3), (d > 0.25).sum() / 0.5 ** 3
    assert near > 5 * far                                           # denser at the wall (h ratio 4 -> ~64x)


def test_triangle_octree_host_guards(wtp):
    """TriangleOctree's constructor checks need no device (src/octree/triangle_octree.jl:338-385,
    427-456; test/octree_isinside.jl:138-165): orientation consistency by exact coordinates, the
    signed-volume guard against inside-out meshes, pure-triangle input."""
    import os

    v = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], dtype=np.float64)
    t = np.array([(1, 3, 2), (1, 4, 3), (5, 6, 7), (5, 7, 8), (1, 2, 6), (1, 6, 5), (3, 4, 8), (3, 8, 7), (1, 5, 8),
                  (1, 8, 4), (2, 3, 7), (2, 7, 6)], dtype=np.int32) - 1
    assert wtp.has_consistent_normals(v, t)
    assert abs(wtp.signed_volume(v, t) - 1.0) < 1e-12
    assert abs(wtp.signed_volume(v, t[:, ::-1]) + 1.0) < 1e-12
    oc = wtp.TriangleOctree(v, t)
    assert len(oc) == 12 and oc.num_triangles == 12 and oc.dtype == np.float64
    with pytest.raises(wtp.WtpArgumentError):
        wtp.TriangleOctree(v, t[:, ::-1].copy())                      # inside-out
    wtp.TriangleOctree(v, t[:, ::-1].copy(), verify_orientation=False)
    wtp.TriangleOctree(v, t[:, ::-1].copy(), classify_leaves=False)  # distance-only use: volume not checked
    flipped = t.copy()
    flipped[3] = flipped[3, ::-1]
    assert not wtp.has_consistent_normals(v, flipped)
    with pytest.raises(wtp.WtpArgumentError):
        wtp.TriangleOctree(v, flipped)
    with pytest.raises(wtp.WtpArgumentError):
        wtp.TriangleOctree(v, np.zeros((4, 4), dtype=np.int32))      # not triangles
    # triangle soup (every corner duplicated, as a binary STL stores it) is judged by coordinates
    soup = v[t].reshape(-1, 3)
    soup_t = np.arange(len(soup), dtype=np.int32).reshape(-1, 3)
    assert wtp.has_consistent_normals(soup, soup_t)
    wv, wt = wtp.octree._weld(v[t])
    assert len(wv) == 8 and np.array_equal(wv[wt], v[t])
    # the reference's own test surfaces, as committed fixtures
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for stem, vol in (("box", 15625.0), ("cavity", None)):
        z = np.load(os.path.join(gold, f"{stem}_mesh.npz"))
        assert wtp.has_consistent_normals(z["vertices"], z["triangles"])
        sv = wtp.signed_volume(z["vertices"], z["triangles"])
        assert sv > 0 and (vol is None or abs(sv - vol) < 1e-3 * vol)


def test_combine_surfaces(wtp):
    """combine_surfaces! (src/surface_operations.jl:7-31; test/surface_operations.jl:1-67): merged in the
    boundary's order under the first given name; unknown names are an assertion error."""
    rng = np.random.default_rng(0)
    a, b, c = (rng.random((n, 3)) for n in (4, 6, 3))
    bnd = wtp.PointBoundary(surfaces={"surface1": wtp.PointSurface(a, a * 0 + 1, np.ones(4)),
                                      "surface2": wtp.PointSurface(b, b * 0 + 2, np.ones(6)),
                                      "surface3": wtp.PointSurface(c, c * 0 + 3, np.ones(3))})
    wtp.combine_surfaces(bnd, "surface3", "surface1")
    assert list(bnd.surfaces) == ["surface2", "surface3"] and len(bnd) == 13
    s = bnd["surface3"]
    assert np.array_equal(s.points(), np.concatenate([a, c])) and np.array_equal(s.normals[:, 0], [1] * 4 + [3] * 3)
    with pytest.raises(AssertionError):
        wtp.combine_surfaces(bnd, "surface2", "nonexistent")
