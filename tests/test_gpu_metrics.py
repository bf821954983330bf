"""metrics / spacing_metrics / spacing_fidelity_metrics (src/metrics.jl) over the device k-NN."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_metrics_collinear_known_answer(ctx, wtp):
    # test/metrics.jl:115-143: 25 collinear points, spacing 1; k=10 and k=20 (self included in k)
    pts = np.array([(i * 1.0, 0.0, 0.0) for i in range(1, 26)])
    m = wtp.metrics(pts, k=10, ctx=ctx, verbose=False)
    assert set(m) >= {"avg", "std", "max", "min", "k", "separation", "fill", "mesh_ratio"}
    assert m["k"] == 10 and m["min"] == 1.0 and m["separation"] == 1.0 and m["fill"] == 1.0 and m["mesh_ratio"] == 1.0
    assert wtp.metrics(pts, k=20, ctx=ctx, verbose=False)["k"] == 20


def test_metrics_match_oracle_distances(ctx, O, wtp, capsys):
    x = wtp.synth.uniform(20000, 3, np.float64, 3)
    m = wtp.metrics(x, k=12, ctx=ctx)
    assert "Cloud Metrics" in capsys.readouterr().out
    _, d = O.knn(x, 12, True, "kdtree")
    r = d[:, 1:]
    assert m["avg"] == pytest.approx(r.mean(axis=1).mean(), rel=1e-12)
    assert m["std"] == pytest.approx(r.std(axis=1, ddof=1).mean(), rel=1e-12)
    assert m["separation"] == r[:, 0].min() and m["fill"] == r[:, 0].max()
    s = 20000 ** (-1 / 3)
    sm = wtp.spacing_metrics(x, wtp.ConstantSpacing(s), k=12, ctx=ctx)
    err = np.abs(r.mean(axis=1) - s) / s
    assert sm["mean_error"] == pytest.approx(err.mean(), rel=1e-12) and sm["max_error"] == pytest.approx(err.max())
    fm = wtp.spacing_fidelity_metrics(x, wtp.ConstantSpacing(s), k=30, ctx=ctx)
    assert 0.4 < fm["mean_dnn_h"] < 0.7 and fm["p05"] < fm["p50"] < fm["p95"] and fm["coordination"] > 5
