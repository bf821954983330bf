"""Committed golden vectors (tests/golden/, made by tools/make_golden.py from the oracle's
brute-force method).  CPU: the oracle's kd-tree path reproduces them.  GPU: libwtp does."""
import glob
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return dict(np.load(os.path.join(G, name)))


def _cases(prefix):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(G, prefix + "*.npz")))


def _cloud(wtp, g):
    return wtp.synth.uniform(int(g["n"]), int(g["dim"]), np.dtype(str(g["dtype"])), int(g["seed"]))


def test_fixture_inventory():
    assert len(_cases("knn_")) >= 4 and len(_cases("radius_")) >= 2 and len(_cases("sweep_")) >= 3


@pytest.mark.parametrize("name", _cases("knn_"))
def test_oracle_knn_golden(O, wtp, name):
    g = _load(name)
    idx, dist = O.knn(_cloud(wtp, g), int(g["k"]), bool(g["include_self"]), "kdtree")
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])


@pytest.mark.parametrize("name", _cases("radius_"))
def test_oracle_radius_golden(O, wtp, name):
    g = _load(name)
    off, idx = O.radius(_cloud(wtp, g), float(g["r"]), "kdtree")
    assert np.array_equal(off, g["offsets"]) and np.array_equal(idx, g["idx"])


@pytest.mark.parametrize("name", _cases("sweep_"))
def test_oracle_sweep_golden(O, wtp, name):
    g = _load(name)
    kind, beta, u0, gamma = g["force"]
    s = float(g["s"])
    r = O.relax_sweep(_cloud(wtp, g), int(g["n_fixed"]), s, int(kind), beta, u0, gamma, int(g["k"]), s / 2000, s / 20)
    assert np.array_equal(r["p"], g["p"]) and np.array_equal(r["nn_id"], g["nn_id"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", _cases("knn_"))
def test_gpu_knn_golden(ctx, wtp, name):
    g = _load(name)
    idx, dist = ctx.knn(_cloud(wtp, g), int(g["k"]), include_self=bool(g["include_self"]), return_dist=True)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", _cases("radius_"))
def test_gpu_radius_golden(ctx, wtp, name):
    g = _load(name)
    off, idx = ctx.radius(_cloud(wtp, g), float(g["r"]))
    assert np.array_equal(off, g["offsets"]) and np.array_equal(idx, g["idx"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", _cases("sweep_"))
def test_gpu_sweep_golden(ctx, wtp, name):
    g = _load(name)
    kind, beta, u0, gamma = g["force"]
    s = float(g["s"])
    with ctx.relax(_cloud(wtp, g), int(g["n_fixed"]), s, dict(kind=int(kind), beta=beta, u0=u0, gamma=gamma),
                   int(g["k"]), s / 2000, s / 20) as sess:
        sess.step(True)
        p = sess.positions()
        pd = sess.point_data()
    assert np.array_equal(pd["nn_id"], g["nn_id"]) and np.array_equal(pd["nn_dist"], g["nn_dist"])
    tol = 1e-5 if p.dtype == np.float32 else 1e-12
    assert np.abs(p - g["p"]).max() <= tol * s
