"""Quality metrics of the reference (src/metrics.jl:19-129) on top of the device k-NN with distance
output.  The neighbour search runs on the GPU; the per-point statistics are
numpy reductions over the returned (n, k) distance matrix."""
from __future__ import annotations

import numpy as np

from .engine import default_context


def _dists(cloud, k, ctx):
    pts = cloud.points() if hasattr(cloud, "points") else np.asarray(cloud)
    k = min(len(pts), int(k))
    _, d = (ctx or default_context()).knn(pts, k, include_self=True, return_dist=True)
    return pts, d[:, 1:], k  # [2:end] skips self (src/metrics.jl:22)


def metrics(cloud, k: int = 20, ctx=None, verbose: bool = True):
    """metrics(cloud; k): avg/std/max/min distance to the k nearest neighbours, separation, fill,
    mesh_ratio (src/metrics.jl:19-41).  std is the sample standard deviation (Julia `std`)."""
    _, r, k = _dists(cloud, k, ctx)
    nn = r[:, 0]
    out = dict(avg=float(r.mean(axis=1).mean()),
               std=float(r.std(axis=1, ddof=1).mean()) if r.shape[1] > 1 else float("nan"),
               max=float(r.max(axis=1).mean()), min=float(r.min(axis=1).mean()),
               separation=float(nn.min()), fill=float(nn.max()), k=k)
    out["mesh_ratio"] = out["fill"] / out["separation"] if out["separation"] > 0 else float("inf")
    if verbose:
        print("Cloud Metrics\n-------------")
        print(f"avg. distance to {k} nearest neighbors: {out['avg']}")
        print(f"std. distance to {k} nearest neighbors: {out['std']}")
        print(f"max. distance to {k} nearest neighbors: {out['max']}")
        print(f"min. distance to {k} nearest neighbors: {out['min']}")
        print(f"separation (min nearest-neighbor distance): {out['separation']}")
        print(f"fill (max nearest-neighbor distance):       {out['fill']}")
        print(f"mesh ratio (fill / separation, ≥1):         {out['mesh_ratio']}")
    return out


def spacing_metrics(cloud, spacing, k: int = 20, ctx=None):
    """Relative error of the local mean neighbour distance against the target spacing
    (src/metrics.jl:56-71)."""
    pts, r, k = _dists(cloud, k, ctx)
    target = np.asarray(spacing(pts) if callable(spacing) else spacing, dtype=np.float64)
    target = np.broadcast_to(target, (len(pts),))
    err = np.abs(r.mean(axis=1, dtype=np.float64) - target) / target
    return dict(max_error=float(err.max()), mean_error=float(err.mean()), std_error=float(err.std(ddof=1)), k=k)


def spacing_fidelity_metrics(cloud, spacing, k: int = 30, coord_radius: float = 1.4, ctx=None):
    """d_NN/h distribution and coordination number (src/metrics.jl:88-129)."""
    pts, r, k = _dists(cloud, k, ctx)
    h = np.asarray(spacing(pts) if callable(spacing) else spacing, dtype=np.float64)
    h = np.broadcast_to(h, (len(pts),))
    dnn_h = r[:, 0].astype(np.float64) / h
    coord = (r <= (coord_radius * h)[:, None]).sum(axis=1)
    mu = float(dnn_h.mean())
    q = np.quantile(dnn_h, [0.05, 0.5, 0.95])
    return dict(mean_dnn_h=mu, cv=float(dnn_h.std(ddof=1) / mu), p05=float(q[0]), p50=float(q[1]), p95=float(q[2]),
                coordination=float(coord.mean()), k=k, coord_radius=coord_radius)
