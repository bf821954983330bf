"""Gradient limiter of the octree discretization (src/discretization/algorithms/octree.jl:677-717):
the g-Lipschitz envelope of a per-leaf spacing field, by min-plus Jacobi sweeps over the k-NN graph of
the leaf centres — structurally the repel sweep with (min, +) in place of the force sum.  The k-NN
graph and every sweep run on the GPU (csrc/wtp_consumers.hip); the octree that produces the leaf
centres stays with the caller."""
from __future__ import annotations

from .engine import default_context


def gradient_limit_field(centers, h0, g: float, k: int = 12, tol: float = 1.0e-3, max_sweeps: int = 2000,
                         return_sweeps: bool = False, ctx=None):
    """h[i] <- min(h[i], min_j h[j] + g d_ij) to a fixpoint (largest relative change of a sweep < tol)."""
    h, sweeps = (ctx or default_context()).gradient_limit(centers, h0, g, k, tol, max_sweeps)
    return (h, sweeps) if return_sweeps else h
